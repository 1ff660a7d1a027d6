// Weight gradient of the regular 3x3 Conv2d layers (stride 1, dilation 1 / 2) on the split-bf16 matrix path: conv3d_split_wgrad.hip
// one dimension down (arithmetic: conv3d_split.hip / DESIGN.md 3j).
//
//     gW[o][c][kh][kw] = sum_{b,h,w} gy[b,o,h,w] * x[b,c, h + (kh-1) d, w + (kw-1) d]        D[i = o][j = c] per tap, GEMM-K = pixels
//
// Reference: the weight gradients cuDNN computes for the nn.Conv2d 3x3 layers of convbn (models/submodule.py:13-17).
//
// A workgroup owns a 32 x 32 (o, c) block and walks work units = (image, 32-pixel column strip, run of 4-row groups); wave v takes
// row v of a group for all nine taps (K-split across the waves: no tap imbalance; the four partial results are summed through LDS in
// wave order at the end).  x rows live in an LDS ring of 12 rows -- the 4 + 2d rows a group reads plus the 4 rows staged for the next
// -- and gy in two buffers, both pixel-fastest in bf16, three pieces each, split once when staged; the next group travels
// global -> registers -> LDS inside the MFMA stream of the current one (loads under the first K-step, split + stores under the
// second), one LDS-only barrier per group.  A tap's fragment starts kw * d elements into an aligned group of 8: ds_read_b128 +
// ds_read_b64 per (kh, piece) and one v_perm_b32 per dword for the shifted taps.  Split-K partials and their reduction are shared
// with the fp32 kernel (conv2d_wgrad.hip).
#include "common.h"

#include "bn_internal.h"
#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 256;
constexpr int RING = 12;                 // x rows in LDS
constexpr int XROWP = 40;                // bf16 per staged x row (32 + 2 d used)
constexpr int XPIECE = RING * XROWP;     // one piece of one channel
constexpr int XCS = 3 * XPIECE + 8;      // 1 448 elements = 724 dwords (= 4 * 181) per channel
constexpr int XALL = 32 * XCS;
constexpr int GPIECE = 4 * 32;
constexpr int GCS = 3 * GPIECE + 8;      // 392 elements = 196 dwords (= 4 * 49) per output channel
constexpr int GBUF = 32 * GCS;
constexpr size_t TILE_BYTES = (size_t)(XALL + 2 * GBUF) * 2;     // 142 848
constexpr size_t SUM_BYTES = (size_t)3 * 9 * 1024 * sizeof(float);  // partial sums of waves 1..3 at the end
constexpr size_t LDS_BYTES = TILE_BYTES > SUM_BYTES ? TILE_BYTES : SUM_BYTES;
constexpr int GIT = 32 * 4 * 16 / NT;    // 8 pixel pairs of a gy group per thread

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  // (the subtractions of a pair stay scalar: packed into v_pk_add_f32 each costs ~9 cycles of the MATRIX pipe -- packed fp32
  // instructions do not overlap with MFMAs on gfx950, plain ones do; tools/experiments/mfma_op_cost.hip)
  p1 = pack2(a, b);
  float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
  asm("" : "+v"(ra), "+v"(rb));
  p2 = pack2(ra, rb);
  float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = rb - __builtin_bit_cast(float, p2 & 0xffff0000u);
  asm("" : "+v"(sa), "+v"(sb));
  p3 = pack2(sa, sb);
}
__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ---- F16: the two-piece fp16 arithmetic of conv3d_split_wgrad.hip: both operands scaled by their tensor's power of two
// when staged, three v_mfma_f32_32x32x16_f16 per product, the block's sums unscaled when they are written
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float f16_scale_of(float m) {  // as in conv3d_split.hip: m * scale in [2^14, 2^15)
  const unsigned e = min(max((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu, 64u), 254u);
  return m == 0.f ? 1.f : __builtin_bit_cast(float, (268u - e) << 23);
}
__device__ __forceinline__ void split2_f16(float a, float b, uint32_t& p1, uint32_t& p2) {
  const f32x2 v = {a, b};
  const f16x2 h1 = __builtin_convertvector(v, f16x2);
  p1 = __builtin_bit_cast(uint32_t, h1);
  float ra = a - (float)h1[0], rb = b - (float)h1[1];
  asm("" : "+v"(ra), "+v"(rb));
  const f32x2 r = {ra, rb};
  p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}
__device__ __forceinline__ f32x16 mfma_f16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int DIL, bool F16>
__global__ __launch_bounds__(NT) void conv2d_bww_split_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                              float* __restrict__ part, mode::Wgrad2SplitDims d,
                                                              const float* __restrict__ amax_g, const float* __restrict__ amax_x) {
  constexpr int NPC = F16 ? 2 : 3;                     // pieces per value
  float sg = 1.f, sx = 1.f, unscale = 1.f;
  if (F16) {
    sg = f16_scale_of(mode::absmax_load(amax_g));
    sx = f16_scale_of(mode::absmax_load(amax_x));
    unscale = (1.f / sg) * (1.f / sx);
  }
  constexpr int NP = (32 + 2 * DIL) / 2;               // pixel pairs per staged x row
  constexpr int XIT = (32 * 4 * NP + NT - 1) / NT;     // 9 pairs of 4 x rows per thread
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  uint16_t* xl = lds;          // [32 c][3 pieces][RING rows][40]
  uint16_t* gl = lds + XALL;   // [2 buffers][32 o][3 pieces][4 rows][32]
  const int s = blockIdx.x, ob = blockIdx.y, cb = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;
  const int HWi = d.H * d.W;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16){0};

  // staging items (the same in every unit): x pair k = (channel, row of the 4 staged, pair of columns), gy pair k likewise
  int x_c[XIT], x_row[XIT], x_wp[XIT], g_o[GIT], g_row[GIT], g_wp[GIT];
#pragma unroll
  for (int k = 0; k < XIT; ++k) {
    const int item = min(tid + k * NT, 32 * 4 * NP - 1);
    x_wp[k] = item % NP;
    x_row[k] = (item / NP) % 4;
    x_c[k] = item / (NP * 4);
  }
#pragma unroll
  for (int k = 0; k < GIT; ++k) {
    const int item = tid + k * NT;
    g_wp[k] = item % 16;
    g_row[k] = (item / 16) % 4;
    g_o[k] = item / 64;
  }
  const float* xb = x;
  const float* gb = gy;
  float xr[XIT][2], gr[GIT][2];
  // Byte offsets of the pair inside row 0 of the (image, 32-channel block) -- or kBufOOB for a column outside the image / a channel
  // beyond the layer's.  The requests are buffer loads (round 6, as in conv3d_split_wgrad.hip): an offset at or beyond the descriptor's
  // size reads as ZERO and a row outside the image takes the empty descriptor, so the zero padding needs no validity masks -- and +
  // compare + select per loaded value were 170 of a group's ~520 vector instructions beside 54 MFMAs.
  // (markers: an invalid column / channel and an invalid row each contribute kHalfOOB = 2^30 -- above every valid offset, which the host
  // keeps below 2^30 -- so that their sum cannot wrap back into the block: a 32-bit kBufOOB + kBufOOB would be 0)
  constexpr unsigned kHalfOOB = 0x40000000u;
  unsigned xo0[XIT], xo1[XIT], go0[GIT], go1[GIT];
  int xdst[XIT], gdst[GIT];
#pragma unroll
  for (int k = 0; k < XIT; ++k) xdst[k] = x_c[k] * XCS + 2 * x_wp[k];
#pragma unroll
  for (int k = 0; k < GIT; ++k) gdst[k] = g_o[k] * GCS + g_row[k] * 32 + 2 * g_wp[k];
  auto unit_begin = [&](int w0) {  // (the item's own row inside the staged group of four is part of the offset)
#pragma unroll
    for (int k = 0; k < XIT; ++k) {
      const int gw = w0 - DIL + 2 * x_wp[k];
      const unsigned cok = (unsigned)(cb * 32 + x_c[k] < d.Ci);
      const unsigned base = 4u * (unsigned)(x_c[k] * HWi + x_row[k] * d.W + gw);
      xo0[k] = (cok & (unsigned)((unsigned)gw < (unsigned)d.W)) ? base : kHalfOOB;
      xo1[k] = (cok & (unsigned)((unsigned)(gw + 1) < (unsigned)d.W)) ? base + 4u : kHalfOOB;
    }
#pragma unroll
    for (int k = 0; k < GIT; ++k) {
      const int gw = w0 + 2 * g_wp[k];
      const unsigned ook = (unsigned)(ob * 32 + g_o[k] < d.Co);
      const unsigned base = 4u * (unsigned)(g_o[k] * HWi + g_row[k] * d.W + gw);
      go0[k] = (ook & (unsigned)(gw < d.W)) ? base : kHalfOOB;
      go1[k] = (ook & (unsigned)(gw + 1 < d.W)) ? base + 4u : kHalfOOB;
    }
  };
  const unsigned block_bytes = 128u * (unsigned)HWi;  // 32 channel planes of an image (the host guarantees < 2^30)
  // x rows [r0, r0 + 4) of the image into ring slots (slot0 + row) % RING; gy rows [r0, r0 + 4) into buffer `buf`
  auto load_x = [&](int k, int r0) {
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(xb, block_bytes);
    const unsigned ro = (unsigned)(r0 + x_row[k]) < (unsigned)d.H ? (unsigned)(4 * r0 * d.W) : kHalfOOB;  // (r0 may be negative: the sum is not)
    xr[k][0] = buf_load_f32(rs, xo0[k] + ro, 0);
    xr[k][1] = buf_load_f32(rs, xo1[k] + ro, 0);
  };
  auto commit_x = [&](int k, int r0, int slot0) {
    uint32_t p1, p2, p3;
    const float v0 = xr[k][0], v1 = xr[k][1];
    int sl = slot0 + x_row[k];
    sl = sl >= RING ? sl - RING : sl;
    uint32_t* dst = reinterpret_cast<uint32_t*>(xl + xdst[k] + sl * XROWP);
    if constexpr (F16) {
      split2_f16(v0 * sx, v1 * sx, p1, p2);
    } else {
      split2(v0, v1, p1, p2, p3);
      dst[XPIECE] = p3;
    }
    dst[0] = p1;  // (threads beyond the last item repeat it: same address, same value)
    dst[XPIECE / 2] = p2;
  };
  auto load_g = [&](int k, int r0) {
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(gb, block_bytes);
    const unsigned ro = r0 + g_row[k] < d.H ? (unsigned)(4 * r0 * d.W) : kHalfOOB;
    gr[k][0] = buf_load_f32(rs, go0[k] + ro, 0);
    gr[k][1] = buf_load_f32(rs, go1[k] + ro, 0);
  };
  auto commit_g = [&](int k, int r0, int buf) {
    uint32_t p1, p2, p3;
    const float v0 = gr[k][0], v1 = gr[k][1];
    uint32_t* dst = reinterpret_cast<uint32_t*>(gl + buf * GBUF + gdst[k]);
    if constexpr (F16) {
      split2_f16(v0 * sg, v1 * sg, p1, p2);
    } else {
      split2(v0, v1, p1, p2, p3);
      dst[GPIECE] = p3;
    }
    dst[0] = p1;
    dst[GPIECE / 2] = p2;
  };

  for (int u = xcd_remap(s, d.S); u < d.units; u += d.S) {
    int t = u;
    const int run = t % d.nRun;
    t /= d.nRun;
    const int wt = t % d.nWt;
    const int b = t / d.nWt;
    const int g_lo = run * d.run_groups, g_hi = min(d.nGroups, g_lo + d.run_groups);
    xb = x + ((long long)b * d.Ci + cb * 32) * HWi;
    gb = gy + ((long long)b * d.Co + ob * 32) * HWi;
    unit_begin(wt * 32);

    // prologue: x rows [h0 - d, h0 - d + 8) into slots 0..7, the gy rows of the first group into buffer 0
    const int h_first = g_lo * 4;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int k = 0; k < XIT; ++k) load_x(k, h_first - DIL + 4 * pass);
#pragma unroll
      for (int k = 0; k < XIT; ++k) commit_x(k, h_first - DIL + 4 * pass, 4 * pass);
    }
#pragma unroll
    for (int k = 0; k < GIT; ++k) load_g(k, h_first);
#pragma unroll
    for (int k = 0; k < GIT; ++k) commit_g(k, h_first, 0);
    lds_barrier();

    int sbase = 0;  // ring slot of x row h0 - d of the current group
#pragma unroll 1
    for (int g = g_lo; g < g_hi; ++g) {
      const int h0 = g * 4, gbuf = (g - g_lo) & 1;
      const int sfront = sbase + 8 >= RING ? sbase + 8 - RING : sbase + 8;  // slot of the first row staged under this group
      // this wave's output row: ring rows of the three kh taps
      int xrow[3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        int sl = sbase + wave + kh * DIL;
        sl = sl >= RING ? sl - RING : sl;
        xrow[kh] = (lane & 31) * XCS + sl * XROWP + 8 * half;
      }
      const uint16_t* ga = gl + gbuf * GBUF + (lane & 31) * GCS + wave * 32 + 8 * half;
      // Six (K-step, kh) stages of 18 MFMAs each (3 kw taps x 6 terms); K-step = 16 pixels of the row.  The raw fragment words of
      // stage i + 1 -- per piece dwords 0..5 of an aligned group of x elements (ds_read_b128 + ds_read_b64) and, at a new K-step, the
      // gy fragment -- are read under the MFMAs of stage i.
      uint4 ra[2][3], rlo[2][3];
      uint2 rhi[2][3];
      auto read_stage = [&](int i) {
        const int ks = i / 3, kh = i % 3;
#pragma unroll
        for (int p = 0; p < NPC; ++p) {
          if (kh == 0) ra[ks][p] = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(ga + p * GPIECE + 16 * ks, 16));
          const uint32_t* src = reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(xl + xrow[kh] + p * XPIECE + 16 * ks, 16));
          rlo[i & 1][p] = *reinterpret_cast<const uint4*>(src);
          rhi[i & 1][p] = *reinterpret_cast<const uint2*>(src + 4);
        }
      };
      read_stage(0);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int ks = i / 3, kh = i % 3;
        // the next group under this one (branch-free; beyond the unit's last group it stages rows nobody reads): loads under the
        // first K-step, split + stores under the second
        // (the next stage's fragment reads in front of the commits' LDS stores -- see conv3d_split_wgrad.hip)
        if (i + 1 < 6) read_stage(i + 1);
        if (i < 3) {
#pragma unroll
          for (int k = 3 * i; k < 3 * i + 3; ++k) load_x(k, h0 - DIL + 8);
#pragma unroll
          for (int k = (GIT * i) / 3; k < (GIT * (i + 1)) / 3; ++k) load_g(k, h0 + 4);
          if (!F16) __builtin_amdgcn_sched_barrier(0);  // in front of this stage's MFMAs (the bf16 group pattern below has no slot for
        } else {                                        // them and put them behind its 18 MFMAs; the fp16 pattern deals them over the gaps)
#pragma unroll
          for (int k = 3 * (i - 3); k < 3 * (i - 3) + 3; ++k) commit_x(k, h0 - DIL + 8, sfront);
#pragma unroll
          for (int k = (GIT * (i - 3)) / 3; k < (GIT * (i - 2)) / 3; ++k) commit_g(k, h0 + 4, gbuf ^ 1);
        }
        uint4 bq[3][3];
#pragma unroll
        for (int p = 0; p < NPC; ++p) {
          const uint4 lo = rlo[i & 1][p];
          const uint2 hi = rhi[i & 1][p];
          const uint32_t dw[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
          constexpr uint32_t SEL_MID = 0x05040302u, SEL_HI = 0x07060504u;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int sh = kw * DIL;  // elements the tap's fragment starts into the aligned group
            if (sh == 0) {
              bq[kw][p] = lo;
            } else {
              const int sd = sh == 4 ? 1 : 0;                    // whole dwords skipped before the permute
              const uint32_t sel = (sh & 1) ? SEL_MID : SEL_HI;  // 1 element = 16 bits; 2 elements = the next dword
              bq[kw][p] = make_uint4(__builtin_amdgcn_perm(dw[sd + 1], dw[sd], sel), __builtin_amdgcn_perm(dw[sd + 2], dw[sd + 1], sel),
                                     __builtin_amdgcn_perm(dw[sd + 3], dw[sd + 2], sel), __builtin_amdgcn_perm(dw[sd + 4], dw[sd + 3], sel));
            }
          }
        }
        if constexpr (F16) {
#define MODE_SPLIT_TERM(PA, PB) \
  _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) acc[3 * kh + kw] = mfma_f16(ra[ks][PA], bq[kw][PB], acc[3 * kh + kw]);
          MODE_SPLIT_TERM(1, 0)
          MODE_SPLIT_TERM(0, 1)
          MODE_SPLIT_TERM(0, 0)
#undef MODE_SPLIT_TERM
        } else {
#define MODE_SPLIT_TERM(PA, PB) \
  _Pragma("unroll") for (int kw = 0; kw < 3; ++kw) acc[3 * kh + kw] = mfma_bf16(ra[ks][PA], bq[kw][PB], acc[3 * kh + kw]);
          MODE_SPLIT_TERM(2, 0)
          MODE_SPLIT_TERM(0, 2)
          MODE_SPLIT_TERM(1, 1)
          MODE_SPLIT_TERM(1, 0)
          MODE_SPLIT_TERM(0, 1)
          MODE_SPLIT_TERM(0, 0)
#undef MODE_SPLIT_TERM
        }
        // (F16: half the MFMAs carry two thirds of the staging work of a stage -- ~50 vector instructions, ~12 loads (stages 0-2) or ~11
        // LDS stores (stages 3-5) and 5 fragment reads beside 9 MFMAs: the allowance per gap is the stage's AVERAGE, the scheduler fills
        // greedily -- 12 + 2 + 3 per gap put a stage's work into its first three gaps)
#pragma unroll
        for (int j = 0; j < (F16 ? 9 : 18); ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (F16) {
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            if (i < 3) {
              __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
            } else {
              __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);
            }
          } else {
            __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      sbase = sbase + 4 >= RING ? sbase + 4 - RING : sbase + 4;
      lds_barrier();
    }
  }

  // sum of the four waves' partial results in wave order (through LDS), written as this workgroup's split-K slice
  float* sums = reinterpret_cast<float*>(lds);  // [3 waves][9 taps][1024]
  if (wave > 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int q = 0; q < 16; ++q) sums[((wave - 1) * 9 + tap) * 1024 + q * 64 + lane] = acc[tap][q];
  }
  __syncthreads();
  if (wave == 0) {
    float* pb = part + (((long long)s * d.MTo + ob) * d.MTc + cb) * (9 * 1024);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = (q & 3) + 8 * (q >> 2) + 4 * half;
        const float v = ((acc[tap][q] + sums[tap * 1024 + q * 64 + lane]) + sums[(9 + tap) * 1024 + q * 64 + lane]) +
                        sums[(18 + tap) * 1024 + q * 64 + lane];
        pb[tap * 1024 + i * 32 + (lane & 31)] = F16 ? v * unscale : v;
      }
  }
}

}  // namespace

namespace mode {

int conv2d_bww_split_launch(const float* gy, const float* x, float* part, const Wgrad2SplitDims& d, int dilation, hipStream_t st,
                            const char* who, const float* amax_g, const float* amax_x) {
#define MODE_C2W_LAUNCH(DILV, F16V)                                                                                              \
  {                                                                                                                              \
    int rc = allow_lds(conv2d_bww_split_kernel<DILV, F16V>, LDS_BYTES, who);                                                     \
    if (rc != MODE_OK) return rc;                                                                                                \
    hipLaunchKernelGGL((conv2d_bww_split_kernel<DILV, F16V>), dim3(d.S, d.MTo, d.MTc), dim3(NT), LDS_BYTES, st, gy, x, part, d, amax_g, \
                       amax_x);                                                                                                  \
  }
  if (dilation == 1) {
    if (amax_g) MODE_C2W_LAUNCH(1, true) else MODE_C2W_LAUNCH(1, false)
  } else {
    if (amax_g) MODE_C2W_LAUNCH(2, true) else MODE_C2W_LAUNCH(2, false)
  }
#undef MODE_C2W_LAUNCH
  return check_launch(who);
}

}  // namespace mode
