// Weight gradient of the STRIDE-2 3x3x3 convolution (and, with the operands exchanged, of ConvTranspose3d k3 s2 p1 op1) on the bf16
// matrix pipe with exactly split fp32 operands (conv3d_split.hip has the arithmetic: three bf16 pieces per fp32 value, six MFMAs per
// product, fp32 accumulation).
//
//     gW[o][c][tap] = sum_{b, q} gy[b, o, q] * x[b, c, 2 q + tap - 1]        D[i = o][j = c] per tap, GEMM-K = output voxels q
//
// Reference: the weight gradients cuDNN computes for hourglass conv1 / conv3 (stride 2) and conv5 / conv6 (ConvTranspose3d),
// models/mode_disparity.py:17-25.
//
// One MFMA (v_mfma_f32_32x32x16_bf16) reduces 16 output voxels of one row: lanes 0..31 carry voxels w0 .. w0+7, lanes 32..63 the next
// eight.  The x operand of tap kw is then x[.., 2 (w0 + m) + kw - 1], a stride-2 walk -- so a staged x row is stored DE-INTERLEAVED, in
// bf16, three pieces:
//       E[m] = x[2 (w0 + m)]       m = 0..15   elements  0..15      kw = 1: one aligned 16-byte read
//       O[m] = x[2 (w0 + m) + 1]   m = -1..15  elements 23..39      kw = 2: one aligned 16-byte read; kw = 0: the same words and the
//                                                                   dword before them, shifted by one element (4 v_perm_b32)
// (a float4 of x is (E[m], O[m], E[m+1], O[m+1]): two packed dword stores per piece; the left halo column O[-1] is one extra item).
// A workgroup (4 waves, one per SIMD) owns a 64 x 32 (o, c) block -- both 32-row blocks of a 64-channel gy share the staged x and its
// B fragments -- and walks work units = (sample, output row, 16 output columns, run of output depths).  Output depth q needs the x
// planes 2q-1, 2q, 2q+1, the next one 2q+1, 2q+2, 2q+3: the planes live in a RING of five (three in use, two being staged), 3 rows
// each, so every x value is loaded and split once per unit; gy has two buffers.  The 27 taps are dealt to the waves as whole
// (kd, kh) groups (wave w: groups w and w + 4, the ninth group one tap each to waves 0..2), 14 accumulators of 16 registers per wave.
// A phase = one output depth = 84 MFMAs per wave; the loads of phase q + 3 are issued at the head of phase q and split + stored during
// phase q + 2 (three register sets), one LDS-only barrier per phase.  Split-K partials in the layout of conv3d.hip's weight-gradient
// kernels, reduced by its fixed-order kernel.
#include "common.h"

#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 256;
constexpr int XROW = 40;               // bf16 per staged x row
constexpr int XO = 24;                 // element of O[0] (O[-1] at 23)
constexpr int XSLOT = 3 * XROW;        // one plane of the ring
constexpr int XPIECE = 5 * XSLOT;      // one piece of one channel: 5 ring planes (1 200 B)
constexpr int XCS = 3 * XPIECE;        // 1 800 elements = 900 dwords (= 4 * 225) per channel: 16-byte reads of 32 channel lanes conflict-free
constexpr int XALL = 32 * XCS;
constexpr int GPIECE = 16;
constexpr int GCS = 3 * GPIECE + 8;    // 56 elements = 28 dwords (= 4 * 7) per output channel
constexpr int GBUF = 64 * GCS;
constexpr size_t LDS_BYTES = (size_t)(XALL + 2 * GBUF) * 2;  // 129 536

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
__device__ __forceinline__ int ring5(int z) { return (z + 10) % 5; }  // z >= -4

// one register set of a phase's staging: six float4 of x (2 planes x 3 rows), the halo element, a float4 of gy
struct Stage {
  float4 xr[6];
  float hr;
  float4 gr;
};

__global__ __launch_bounds__(NT) void conv3d_bww_s2_split_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                 float* __restrict__ part, mode::WgradS2SplitDims d) {
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  uint16_t* xl = lds;         // [32 c][3 pieces][5 planes][3 rows][40]
  uint16_t* gl = lds + XALL;  // [2 buffers][64 o][3 pieces][16]
  const int s = blockIdx.x, obp = blockIdx.y, cb = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;
  const int HWi = d.H * d.W, DHWi = d.D * HWi, oHW = d.Ho * d.Wo, oDHW = d.Do * oHW;  // (host: 32-bit byte offsets within a sample)

  // taps of this wave: slots 0..2 = group wave (kw = slot), 3..5 = group wave + 4, slot 6 = tap 24 + wave (wave 3: tap 26 again, dropped)
  const int g0 = wave, g1 = wave + 4;
  const int kw6 = min(wave, 2);
  const uint32_t sel6 = kw6 == 0 ? 0x05040302u : 0x07060504u;  // shift by one element / identity
  const int off6 = kw6 == 1 ? 0 : XO, poff6 = kw6 == 0 ? XO - 2 : off6;

  f32x16 acc[2][7];
#pragma unroll
  for (int o2 = 0; o2 < 2; ++o2)
#pragma unroll
    for (int t = 0; t < 7; ++t) acc[o2][t] = (f32x16){0};

  // staging items of this thread, the same in every unit
  const int f4 = tid & 7, xc = tid >> 3;                 // x: float4 f4 of a row, channel xc; item k = (plane k / 3, row k % 3)
  const int hc = tid & 31, hk = (tid >> 5) < 6 ? (tid >> 5) : (tid >> 5) - 2;  // halo column: channel hc, (plane, row) hk < 6 (threads
                                                         // 192..255 repeat the items of 128..191: no branch around the stores)
  const int hpz = hk / 3, hrw = hk % 3;
  const int go = tid >> 2, gf4 = tid & 3;                // gy: output channel go of the 64, float4 gf4 of the 16 voxels
  const int xdst = xc * XCS + 2 * f4;                    // (+ slot * XSLOT + row * XROW [+ XO] + piece * XPIECE)
  const int hdst = hc * XCS + hrw * XROW + XO - 1;
  const int gdst = go * GCS + 4 * gf4;

  const float* xb = x;
  const float* gb = gy;
  unsigned xo[3], ho = 0, gof = 0;  // byte offsets inside plane 0 / depth 0 of the sample (clamped into the volume)
  unsigned xm = 0, hm = 0, gm = 0;  // validity (row bits for x)
  auto unit_begin = [&](int qh, int w0) {
    const int gw = 2 * w0 + 4 * f4;
    const unsigned cok = (unsigned)(gw < d.W) & (unsigned)(cb * 32 + xc < d.Ci);
    xm = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int gh = 2 * qh - 1 + r;
      const unsigned ok = cok & (unsigned)((unsigned)gh < (unsigned)d.H);
      xo[r] = ok ? 4u * (unsigned)(xc * DHWi + gh * d.W + gw) : 0u;
      xm |= ok << r;
    }
    {
      const int gh = 2 * qh - 1 + hrw, gwh = 2 * w0 - 1;
      hm = (unsigned)(gwh >= 0) & (unsigned)((unsigned)gh < (unsigned)d.H) & (unsigned)(cb * 32 + hc < d.Ci);
      ho = hm ? 4u * (unsigned)(hc * DHWi + gh * d.W + gwh) : 0u;
    }
    {
      const int gw2 = w0 + 4 * gf4;
      gm = (unsigned)(gw2 < d.Wo) & (unsigned)(obp * 64 + go < d.Co);
      gof = gm ? 4u * (unsigned)(go * oDHW + qh * d.Wo + gw2) : 0u;
    }
  };
  // loads of the two x planes 2q, 2q + 1 and of the gy row of output depth q: unconditional, from clamped addresses
  auto load_set = [&](Stage& st, int q) {
    const unsigned z0 = 4u * (unsigned)(min(max(2 * q, 0), d.D - 1) * HWi), z1 = 4u * (unsigned)(min(max(2 * q + 1, 0), d.D - 1) * HWi);
#pragma unroll
    for (int k = 0; k < 6; ++k)
      st.xr[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xb) + (xo[k % 3] + (k < 3 ? z0 : z1)));
    st.hr = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xb) + (ho + (hpz ? z1 : z0)));
    st.gr = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(gb) + (gof + 4u * (unsigned)(min(max(q, 0), d.Do - 1) * oHW)));
  };
  auto commit_x = [&](const Stage& st, int q, int k) {
    const int z = 2 * q + k / 3;
    const bool ok = ((unsigned)z < (unsigned)d.D) && ((xm >> (k % 3)) & 1u);
    const float4 v = st.xr[k];
    uint32_t e1, e2, e3, o1, o2, o3;
    split2(ok ? v.x : 0.f, ok ? v.z : 0.f, e1, e2, e3);
    split2(ok ? v.y : 0.f, ok ? v.w : 0.f, o1, o2, o3);
    uint32_t* dst = reinterpret_cast<uint32_t*>(xl + xdst + ring5(z) * XSLOT + (k % 3) * XROW);
    dst[0] = e1;
    dst[XPIECE / 2] = e2;
    dst[XPIECE] = e3;
    dst[XO / 2] = o1;
    dst[XO / 2 + XPIECE / 2] = o2;
    dst[XO / 2 + XPIECE] = o3;
  };
  auto commit_hg = [&](const Stage& st, int q) {
    {
      const int z = 2 * q + hpz;
      const float v = (hm && (unsigned)z < (unsigned)d.D) ? st.hr : 0.f;
      uint32_t p1, p2, p3;
      split2(v, 0.f, p1, p2, p3);
      uint16_t* dst = xl + hdst + ring5(z) * XSLOT;
      dst[0] = (uint16_t)p1;
      dst[XPIECE] = (uint16_t)p2;
      dst[2 * XPIECE] = (uint16_t)p3;
    }
    {
      const bool ok = gm && (unsigned)q < (unsigned)d.Do;
      const float4 v = st.gr;
      uint32_t a1, a2, a3, b1, b2, b3;
      split2(ok ? v.x : 0.f, ok ? v.y : 0.f, a1, a2, a3);
      split2(ok ? v.z : 0.f, ok ? v.w : 0.f, b1, b2, b3);
      uint2* dst = reinterpret_cast<uint2*>(gl + (q & 1) * GBUF + gdst);
      dst[0] = make_uint2(a1, b1);
      dst[GPIECE / 4] = make_uint2(a2, b2);
      dst[GPIECE / 2] = make_uint2(a3, b3);
    }
  };

  // (Measured and dropped: the six items of a set in an order rotated by channel, so that one load instruction of the workgroup spreads
  // over six (plane, row) offsets instead of 32 channels at ONE offset -- the channels of the benchmark volume are 3 * 2^21 bytes apart
  // and the kernel is 15-20 % faster per voxel at 50 x 256 x 128 or 46 x 256 x 128; the rotation cost 26 registers and 8 % of the time
  // and bought nothing.)
  // one output depth: the MFMAs of depth dd on the ring; `cs` (the planes and gy row of depth dd + 1, loaded two phases ago) is split
  // and stored under them, the loads of depth dd + 3 go into `ls` (measured at 48 x 256 x 128: with the loads one phase ahead a phase
  // lasted 3.4 us against 1.9 us without memory traffic -- 32 channels 6.3 MB apart, every workgroup on the same beat)
  auto phase = [&](int dd, Stage& cs, Stage& ls) {
    load_set(ls, dd + 3);
    __builtin_amdgcn_sched_barrier(0);  // the loads leave at the head of the phase: they are split and stored two phases later
    const int lc = (lane & 31) * XCS + 8 * half;
    const int b0 = lc + ring5(2 * dd - 1 + g0 / 3) * XSLOT + (g0 % 3) * XROW;
    const int b1 = lc + ring5(2 * dd - 1 + g1 / 3) * XSLOT + (g1 % 3) * XROW;
    const int b2 = lc + ring5(2 * dd + 1) * XSLOT + 2 * XROW;
    const uint16_t* ga = gl + (dd & 1) * GBUF + (lane & 31) * GCS + 8 * half;
    uint4 a[2][3], bq[7][3];
    auto read_group = [&](int g, int base) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const uint16_t* src = xl + base + p * XPIECE;
        const uint4 e = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(src, 16));
        const uint4 o = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(src + XO, 16));
        const uint32_t pv = *reinterpret_cast<const uint32_t*>(src + XO - 2);
        bq[3 * g + 0][p] = make_uint4(__builtin_amdgcn_perm(o.x, pv, 0x05040302u), __builtin_amdgcn_perm(o.y, o.x, 0x05040302u),
                                      __builtin_amdgcn_perm(o.z, o.y, 0x05040302u), __builtin_amdgcn_perm(o.w, o.z, 0x05040302u));
        bq[3 * g + 1][p] = e;
        bq[3 * g + 2][p] = o;
      }
    };
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      a[0][p] = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(ga + p * GPIECE, 16));
      a[1][p] = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(ga + 32 * GCS + p * GPIECE, 16));
    }
    read_group(0, b0);
    __builtin_amdgcn_sched_barrier(0);
    // smallest terms first; consecutive MFMAs go to different accumulators
#define MODE_S2W_TERM(T0, T1, PA, PB)                          \
  _Pragma("unroll") for (int t7 = T0; t7 < T1; ++t7) {         \
    acc[0][t7] = mfma_bf16(a[0][PA], bq[t7][PB], acc[0][t7]);  \
    acc[1][t7] = mfma_bf16(a[1][PA], bq[t7][PB], acc[1][t7]);  \
  }
#define MODE_S2W_SET(T0, T1)  \
  MODE_S2W_TERM(T0, T1, 2, 0) \
  MODE_S2W_TERM(T0, T1, 0, 2) \
  MODE_S2W_TERM(T0, T1, 1, 1) \
  MODE_S2W_TERM(T0, T1, 1, 0) \
  MODE_S2W_TERM(T0, T1, 0, 1) \
  MODE_S2W_TERM(T0, T1, 0, 0)
    // first 36 MFMAs (group g0): under them the fragments of the other four taps are read and half of the staged set is split + stored
    read_group(1, b1);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const uint16_t* src = xl + b2 + p * XPIECE;
      const uint4 o = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(src + off6, 16));
      const uint32_t pv = *reinterpret_cast<const uint32_t*>(src + poff6);
      bq[6][p] = make_uint4(__builtin_amdgcn_perm(o.x, pv, sel6), __builtin_amdgcn_perm(o.y, o.x, sel6), __builtin_amdgcn_perm(o.z, o.y, sel6),
                            __builtin_amdgcn_perm(o.w, o.z, sel6));
    }
    MODE_S2W_SET(0, 3)
#pragma unroll
    for (int k = 0; k < 3; ++k) commit_x(cs, dd + 1, k);
#pragma unroll
    for (int i = 0; i < 15; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {  // (a store needs the ~40 instructions of its split first: no store slots yet)
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
    }
#pragma unroll
    for (int i = 0; i < 13; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    MODE_S2W_SET(3, 7)
#pragma unroll
    for (int k = 3; k < 6; ++k) commit_x(cs, dd + 1, k);
    commit_hg(cs, dd + 1);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#undef MODE_S2W_SET
#undef MODE_S2W_TERM
    lds_barrier();
  };

  Stage s0, s1, s2;
  for (int u = xcd_remap(s, d.S); u < d.units; u += d.S) {
    int t = u;
    const int dc = t % d.nDc;
    t /= d.nDc;
    const int wt = t % d.nWt;
    t /= d.nWt;
    const int qh = t % d.Ho;
    const int b = t / d.Ho;
    unit_begin(qh, wt * 16);
    const int dlo = dc * d.ring_dc, dhi = min(d.Do, dlo + d.ring_dc);
    xb = x + ((long long)b * d.Ci + cb * 32) * DHWi;
    gb = gy + ((long long)b * d.Co + obp * 64) * oDHW;

    // prologue of a unit (the last barrier of the previous unit has passed): planes 2 dlo - 2 .. 2 dlo + 1 and the gy row of depth dlo
    // (plane 2 dlo - 2 and gy row dlo - 1 ride along unused: one staging routine); the loads of depths dlo + 1 and dlo + 2 start
    load_set(s0, dlo - 1);
    load_set(s1, dlo);
#pragma unroll
    for (int k = 0; k < 6; ++k) commit_x(s0, dlo - 1, k);
    commit_hg(s0, dlo - 1);
#pragma unroll
    for (int k = 0; k < 6; ++k) commit_x(s1, dlo, k);
    commit_hg(s1, dlo);
    load_set(s0, dlo + 1);
    load_set(s1, dlo + 2);
    lds_barrier();

#pragma unroll 1
    for (int dd = dlo; dd < dhi; dd += 3) {
      phase(dd, s0, s2);
      if (dd + 1 < dhi) phase(dd + 1, s1, s0);
      if (dd + 2 < dhi) phase(dd + 2, s2, s1);
    }
  }

#pragma unroll
  for (int o2 = 0; o2 < 2; ++o2) {
    const int ob = obp * 2 + o2;
    if (ob < d.MTo) {
      float* pb = part + (((long long)s * d.MTo + ob) * d.MTc + cb) * (27 * 1024);
#pragma unroll
      for (int t7 = 0; t7 < 7; ++t7) {
        const int tap = t7 < 6 ? 3 * (wave + 4 * (t7 / 3)) + t7 % 3 : 24 + wave;
        if (tap < 27) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int i = (q & 3) + 8 * (q >> 2) + 4 * half;
            pb[tap * 1024 + i * 32 + (lane & 31)] = acc[o2][t7][q];
          }
        }
      }
    }
  }
}

}  // namespace

namespace mode {

int conv3d_bww_s2_split_launch(const float* gy, const float* x, float* part, const WgradS2SplitDims& d, hipStream_t st, const char* who) {
  int rc = allow_lds(conv3d_bww_s2_split_kernel, LDS_BYTES, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(conv3d_bww_s2_split_kernel, dim3(d.S, cdiv(d.MTo, 2), d.MTc), dim3(NT), LDS_BYTES, st, gy, x, part, d);
  return check_launch(who);
}

}  // namespace mode
