// Weight gradient of the stride-1 3x3x3 convolution on the bf16 matrix pipe with exactly split fp32 operands (see conv3d_split.hip
// for the arithmetic: three bf16 pieces per fp32 value, six MFMAs per product, fp32 accumulation -- fp32 accuracy).
//
//     gW[o][c][tap] = sum_{b, voxel} gy[b, o, voxel] * x[b, c, voxel + tap - 1]          D[i = o][j = c] per tap, GEMM-K = voxels
//
// Reference: the weight gradients cuDNN computes for the nn.Conv3d layers of convbn_3d (models/submodule.py:20-22).
//
// Same decomposition as conv3d_bwd_weight_ring_kernel (conv3d.hip): a workgroup owns a 32 x 32 (o, c) block and walks work units
// = (sample, 2-row x 32-voxel column, run of depths) with the x planes d-1, d, d+1 in an LDS ring; split-K partials go to the same
// workspace layout and are reduced by the same fixed-order kernel.  What differs:
//   * one MFMA (v_mfma_f32_32x32x16_bf16) reduces 16 voxels: lanes 0..31 carry 8 consecutive voxels of a row, lanes 32..63 the next
//     8, so both operands are stored voxel-fastest in bf16, three pieces each, split ONCE when a plane is staged:
//       x  [32 c][3 pieces][4 ring planes][4 rows][40 (34 used)]   gy [2 buffers][32 o][3 pieces][2 rows][32]
//     (channel strides padded to 4 * odd dwords: the 16-byte fragment reads of the 32 channel lanes are conflict-free);
//   * a tap's B fragment starts kw elements into an aligned group of 8: one ds_read_b128 + one ds_read_b32 fetch 10 elements and
//     kw = 1 is four v_alignbit, kw = 0 / 2 a choice of registers.  Taps are dealt to the 4 waves as whole (kd, kh) groups -- wave w
//     owns groups w and w + 4, i.e. kw = 0, 1, 2 at compile-time positions -- and the ninth group goes one tap each to waves 0..2
//     (27 taps on 28 slots);
//   * the ring has a fourth plane and gy two buffers, so the next depth is staged (loads under the first half of the 168 MFMAs of
//     a depth, v_cvt_pk_bf16_f32 split + LDS stores under the second) while the current one is multiplied: one LDS-only barrier per
//     depth, one wave per SIMD.
#include "common.h"

#include "bn_internal.h"
#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 256;
constexpr int WTH = 2, XR = WTH + 2;
constexpr int XROWP = 40;                           // bf16 per staged x row: w = w0 - 1 .. w0 + 32 in elements 0 .. 33
constexpr int XPLANE = XR * XROWP;                  // 160
constexpr int XPIECE = 4 * XPLANE;                  // one piece of one channel: 4 ring planes (1 280 B: pieces, rows and column groups of a
                                                    // fragment read are immediate offsets from ONE address register per tap)
constexpr int XCS = 3 * XPIECE + 8;                 // 1 928 elements = 964 dwords (= 4 * 241) per channel
constexpr int GPIECE = WTH * 32;                    // 64
constexpr int GCS = 3 * GPIECE + 8;                 // 200 elements = 100 dwords (= 4 * 25) per output channel
constexpr int GBUF = 32 * GCS;
constexpr int XALL = 32 * XCS;
constexpr size_t LDS_BYTES = (size_t)(XALL + 2 * GBUF) * 2;  // 148 992
constexpr int XIT = (32 * XR * 17 + NT - 1) / NT;   // 9 element pairs of an x plane per thread
constexpr int GIT = 32 * WTH * 16 / NT;             // 4 element pairs of the gy rows per thread

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// F16: two fp16 pieces (round to nearest even), three MFMAs per product -- the experimental arithmetic of conv3d_split.hip (its header);
// the operands arrive scaled by a power of two, the sums leave scaled back.
template <bool F16>
__device__ __forceinline__ void split2(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  if constexpr (F16) {
    const f32x2 v = {a, b};
    const f16x2 h1 = __builtin_convertvector(v, f16x2);
    p1 = __builtin_bit_cast(uint32_t, h1);
    const f32x2 r = {a - (float)h1[0], __builtin_fmaf(-1.f, (float)h1[1], b)};
    p2 = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
    p3 = 0;
  } else {
    // (the subtractions of a pair stay scalar: packed into v_pk_add_f32 each costs ~9 cycles of the MATRIX pipe -- packed fp32
    // instructions do not overlap with MFMAs on gfx950, plain ones do; tools/experiments/mfma_op_cost.hip)
    // One of the pair as a subtraction, the other as fma(-1, piece, value) (the same exact difference): two different operations are not
    // packed, and no empty asm statement is needed to keep them apart -- the scheduler's group pattern places plain VALU instructions
    // under the MFMAs, an inline-asm node in a chain it left (with everything behind it) for the end of the K-step.
    p1 = pack2(a, b);
    const float ra = a - __builtin_bit_cast(float, p1 << 16), rb = __builtin_fmaf(-1.f, __builtin_bit_cast(float, p1 & 0xffff0000u), b);
    p2 = pack2(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = __builtin_fmaf(-1.f, __builtin_bit_cast(float, p2 & 0xffff0000u), rb);
    p3 = pack2(sa, sb);
  }
}
__device__ __forceinline__ float f16_scale_of(float m) {  // as in conv3d_split.hip: m * scale in [2^14, 2^15)
  const unsigned e = min(max((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu, 64u), 254u);
  return m == 0.f ? 1.f : __builtin_bit_cast(float, (268u - e) << 23);
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma_split(uint4 a, uint4 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool F16>
__global__ __launch_bounds__(NT) void conv3d_bww_split_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                              float* __restrict__ part, mode::WgradSplitDims d,
                                                              const float* __restrict__ amax_g, const float* __restrict__ amax_x) {
  constexpr int NP = F16 ? 2 : 3;  // pieces (the LDS layout keeps room for three)
  // (F16) amax_g / amax_x = max |gy| / max |x|: both operands scaled when staged, the sums scaled back when written
  const float sg = F16 ? f16_scale_of(mode::absmax_load(amax_g)) : 1.f, sx = F16 ? f16_scale_of(mode::absmax_load(amax_x)) : 1.f;
  const float unscale = F16 ? (1.f / sg) * (1.f / sx) : 1.f;
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  uint16_t* xl = lds;          // [32 c][3 pieces][4 planes][4 rows][40]
  uint16_t* gl = lds + XALL;   // [2 buffers][32 o][3 pieces][2 rows][32]
  const int s = blockIdx.x, ob = blockIdx.y, cb = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;
  const int HWi = d.H * d.W, DHWi = d.D * HWi;  // (host guarantees 32-bit element offsets within a sample)

  // taps of this wave: slots 0..2 = group wave, 3..5 = group wave + 4 (kw = slot % 3), slot 6 = tap 24 + wave (wave 3: tap 26 again,
  // dropped)
  int tkd[7], tkh[7];
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int g = wave + 4 * (t / 3);
    tkd[t] = g / 3;
    tkh[t] = g % 3;
  }
  tkd[6] = 2;
  tkh[6] = 2;
  const int kw6 = min(wave, 2);
  const uint32_t sel6 = kw6 == 0 ? 0x03020100u : kw6 == 1 ? 0x05040302u : 0x07060504u;  // v_perm_b32 byte selector of a shift by kw6 elements

  // Two accumulators per tap: the leading products a1*b1 and the five correction terms (<= 2^-7 of them).  A sum over ~10^5 voxels
  // per workgroup is a long fp32 chain, and six additions per K-step into ONE accumulator round six times at the magnitude of the
  // running sum (measured: 2.6 x the error of the fp32 MFMA kernel); kept apart, the corrections round at 2^-7 of that magnitude and
  // the leading chain has one addition per 16 voxels -- an eighth of the fp32 kernel's.
  f32x16 acc[7], acs[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    acc[t] = (f32x16){0};
    acs[t] = (f32x16){0};
  }

  // staging items, the same in every unit: x pair k = (channel, row, pair of columns), gy pair k = (channel, row, pair of columns)
  int x_c[XIT], x_row[XIT], x_wp[XIT], g_o[GIT], g_row[GIT], g_wp[GIT];
#pragma unroll
  for (int k = 0; k < XIT; ++k) {
    const int item = min(tid + k * NT, 32 * XR * 17 - 1);
    x_wp[k] = item % 17;
    x_row[k] = (item / 17) % XR;
    x_c[k] = item / (17 * XR);
  }
#pragma unroll
  for (int k = 0; k < GIT; ++k) {
    const int item = tid + k * NT;
    g_wp[k] = item % 16;
    g_row[k] = (item / 16) % WTH;
    g_o[k] = item / (16 * WTH);
  }

  float xr[XIT][2], gr[GIT][2];
  // Per unit and item: the two BYTE offsets inside plane 0 of the (sample, 32-channel block) the unit reads (a sample is < 2^29
  // elements) -- or kBufOOB for an element outside the volume / beyond the layer's channels; per depth only the plane offset (a scalar)
  // is added.  The requests are buffer loads (round 6): the block's 32 channel planes are one descriptor, an offset at or beyond its
  // size reads as ZERO, and a depth outside the volume takes the empty descriptor -- so the zero padding costs nothing per value (the
  // validity masks were 108 of a depth's 518 vector / scalar instructions beside 84 MFMAs: and + compare + select per loaded value).
  unsigned xo0[XIT], xo1[XIT], go0[GIT], go1[GIT];
  int xdst[XIT], gdst[GIT];
#pragma unroll
  for (int k = 0; k < XIT; ++k) xdst[k] = x_c[k] * XCS + x_row[k] * XROWP + 2 * x_wp[k];
#pragma unroll
  for (int k = 0; k < GIT; ++k) gdst[k] = g_o[k] * GCS + g_row[k] * 32 + 2 * g_wp[k];  // (+ piece * GPIECE)
  auto unit_begin = [&](int h0, int w0) {
#pragma unroll
    for (int k = 0; k < XIT; ++k) {
      const int gh = h0 - 1 + x_row[k], gw = w0 - 1 + 2 * x_wp[k];
      const unsigned rowok = (unsigned)((unsigned)gh < (unsigned)d.H) & (unsigned)(cb * 32 + x_c[k] < d.Ci);
      const unsigned base = 4u * (unsigned)(x_c[k] * DHWi + gh * d.W + gw);
      xo0[k] = (rowok & (unsigned)((unsigned)gw < (unsigned)d.W)) ? base : kBufOOB;
      xo1[k] = (rowok & (unsigned)((unsigned)(gw + 1) < (unsigned)d.W)) ? base + 4u : kBufOOB;
    }
#pragma unroll
    for (int k = 0; k < GIT; ++k) {
      const int gh = h0 + g_row[k], gw = w0 + 2 * g_wp[k];
      const unsigned rowok = (unsigned)(gh < d.H) & (unsigned)(ob * 32 + g_o[k] < d.Co);
      const unsigned base = 4u * (unsigned)(g_o[k] * DHWi + gh * d.W + gw);
      go0[k] = (rowok & (unsigned)(gw < d.W)) ? base : kBufOOB;
      go1[k] = (rowok & (unsigned)(gw + 1 < d.W)) ? base + 4u : kBufOOB;
    }
  };
  const unsigned block_bytes = 128u * (unsigned)DHWi;  // 32 channels of a sample (the host guarantees < 2^31)
  // loads of x plane z and of the gy rows of depth z: unconditional; a plane outside the volume reads through an EMPTY descriptor
  const float* xb = x;
  const float* gb = gy;
  auto load_x = [&](int k, int z) {
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(xb, (unsigned)z < (unsigned)d.D ? block_bytes : 0u);
    const unsigned zo = 4u * (unsigned)(min(max(z, 0), d.D - 1) * HWi);
    xr[k][0] = buf_load_f32(rs, xo0[k], zo);
    xr[k][1] = buf_load_f32(rs, xo1[k], zo);
  };
  auto commit_x = [&](int k, int z) {
    uint32_t p1, p2, p3;
    float v0 = xr[k][0], v1 = xr[k][1];
    if (F16) {
      v0 *= sx;
      v1 *= sx;
    }
    split2<F16>(v0, v1, p1, p2, p3);
    uint32_t* dst = reinterpret_cast<uint32_t*>(xl + xdst[k] + ((z + 4) & 3) * XPLANE);
    dst[0] = p1;  // (threads beyond the last item repeat it: same address, same value -- no conditional store in the MFMA stream)
    dst[XPIECE / 2] = p2;
    if (!F16) dst[XPIECE] = p3;
  };
  auto load_g = [&](int k, int z) {
    const __amdgpu_buffer_rsrc_t rs = buf_rsrc(gb, z < d.D ? block_bytes : 0u);
    const unsigned zo = 4u * (unsigned)(min(z, d.D - 1) * HWi);
    gr[k][0] = buf_load_f32(rs, go0[k], zo);
    gr[k][1] = buf_load_f32(rs, go1[k], zo);
  };
  auto commit_g = [&](int k, int z) {
    uint32_t p1, p2, p3;
    float v0 = gr[k][0], v1 = gr[k][1];
    if (F16) {
      v0 *= sg;
      v1 *= sg;
    }
    split2<F16>(v0, v1, p1, p2, p3);
    uint32_t* dst = reinterpret_cast<uint32_t*>(gl + gdst[k] + (z & 1) * GBUF);
    dst[0] = p1;
    dst[GPIECE / 2] = p2;
    if (!F16) dst[GPIECE] = p3;
  };

  for (int u = xcd_remap(s, d.S); u < d.units; u += d.S) {
    int t = u;
    const int dc = t % d.nDc;
    t /= d.nDc;
    const int wt = t % d.nWt;
    t /= d.nWt;
    const int ht = t % d.nHt;
    const int b = t / d.nHt;
    unit_begin(ht * WTH, wt * 32);
    const int dlo = dc * d.ring_dc, dhi = min(d.D, dlo + d.ring_dc);
    xb = x + ((long long)b * d.Ci + cb * 32) * DHWi;
    gb = gy + ((long long)b * d.Co + ob * 32) * DHWi;

    // prologue of a unit: planes dlo-1, dlo, dlo+1 and the gy rows of depth dlo (the last barrier of the previous unit has passed)
#pragma unroll 1
    for (int z = dlo - 1; z <= dlo + 1; ++z) {
#pragma unroll
      for (int k = 0; k < XIT; ++k) load_x(k, z);
#pragma unroll
      for (int k = 0; k < XIT; ++k) commit_x(k, z);
    }
#pragma unroll
    for (int k = 0; k < GIT; ++k) load_g(k, dlo);
#pragma unroll
    for (int k = 0; k < GIT; ++k) commit_g(k, dlo);
    lds_barrier();

#pragma unroll 1
    for (int dd = dlo; dd < dhi; ++dd) {
      // fragments: element offsets of this depth's planes for the 7 tap slots
      int xoff[7];
#pragma unroll
      for (int t7 = 0; t7 < 7; ++t7) xoff[t7] = (lane & 31) * XCS + ((dd + tkd[t7] + 3) & 3) * XPLANE + tkh[t7] * XROWP + 8 * half;
      const uint16_t* ga = gl + (dd & 1) * GBUF + (lane & 31) * GCS + 8 * half;
      // Raw fragment words of a K-step (16 voxels: row ks / 2, columns 16 * (ks % 2) .. + 15), read one K-step ahead of their use:
      // per piece the gy fragment, and for each of the three (kd, kh) groups of this wave (slots 0-2, 3-5, 6) dwords 0..4 of an
      // aligned group of 10 x elements, fetched as ds_read_b128 + ds_read_b64 (dword-wide LDS reads of 32 channel lanes 964 dwords
      // apart would be 4-way bank conflicts).
      uint4 ra[2][NP], rlo[2][3][NP];
      uint2 rhi[2][3][NP];
      auto read_raw = [&](int ks, int set) {
        const int row = ks / 2, w16 = 16 * (ks % 2);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          ra[set][p] = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(ga + p * GPIECE + row * 32 + w16, 16));
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            const uint32_t* src =
                reinterpret_cast<const uint32_t*>(__builtin_assume_aligned(xl + xoff[3 * g] + p * XPIECE + row * XROWP + w16, 16));
            rlo[set][g][p] = *reinterpret_cast<const uint4*>(src);
            rhi[set][g][p] = *reinterpret_cast<const uint2*>(src + 4);
          }
        }
      };
      read_raw(0, 0);
#pragma unroll
      for (int ks = 0; ks < 2 * WTH; ++ks) {
        // the next depth under this one: loads in the first two K-steps, split + store in the last two (branch-free; beyond the
        // unit's last depth it stages a plane / rows nobody reads)
        if (ks == 0) {
#pragma unroll
          for (int k = 0; k < 5; ++k) load_x(k, dd + 2);
#pragma unroll
          for (int k = 0; k < GIT; ++k) load_g(k, dd + 1);
          __builtin_amdgcn_sched_barrier(0);  // (in front of this K-step's MFMAs: left to the group pattern below, which has no slot for
        }                                     // them, the requests were placed behind its 42 MFMAs -- half the lead to their commit)
        if (ks == 1) {
#pragma unroll
          for (int k = 5; k < XIT; ++k) load_x(k, dd + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
        // (the next K-step's fragment reads in front of the commits: behind them in program order they could not be placed ahead of the
        // commits' LDS stores -- the compiler does not know the buffers are different -- and ended up, with the stores, behind the MFMAs)
        if (ks + 1 < 2 * WTH) read_raw(ks + 1, (ks + 1) & 1);
        if (ks == 2) {
#pragma unroll
          for (int k = 0; k < 5; ++k) commit_x(k, dd + 2);
#pragma unroll
          for (int k = 0; k < GIT / 2; ++k) commit_g(k, dd + 1);
        }
        if (ks == 3) {
#pragma unroll
          for (int k = 5; k < XIT; ++k) commit_x(k, dd + 2);
#pragma unroll
          for (int k = GIT / 2; k < GIT; ++k) commit_g(k, dd + 1);
        }
        // the fragment of tap kw is elements kw .. kw + 7 of the 10: one v_perm_b32 per dword with the byte selector of the shift (an
        // MFMA operand is an even-aligned register quadruple: "dwords 1..4" is not addressable as such)
        uint4 a[NP], bq[7][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          a[p] = ra[ks & 1][p];
#pragma unroll
          for (int t7 = 0; t7 < 7; ++t7) {
            const uint4 lo = rlo[ks & 1][t7 / 3][p];
            const uint2 hi = rhi[ks & 1][t7 / 3][p];
            if (t7 < 6 && t7 % 3 == 0) {
              bq[t7][p] = lo;
            } else {
              const uint32_t sel = t7 < 6 ? (t7 % 3 == 1 ? 0x05040302u : 0x07060504u) : sel6;
              bq[t7][p] = make_uint4(__builtin_amdgcn_perm(lo.y, lo.x, sel), __builtin_amdgcn_perm(lo.z, lo.y, sel),
                                     __builtin_amdgcn_perm(lo.w, lo.z, sel), __builtin_amdgcn_perm(hi.x, lo.w, sel));
            }
          }
        }
        // smallest terms first; consecutive MFMAs go to different accumulators
#define MODE_SPLIT_TERM(ACC, PA, PB) \
  _Pragma("unroll") for (int t7 = 0; t7 < 7; ++t7) ACC[t7] = mfma_split<F16>(a[PA], bq[t7][PB], ACC[t7]);
        if constexpr (F16) {
          // tap-PAIR major (round 6): the six MFMAs of taps (a, b) need only those two taps' shifted fragments, so the 40 v_perm of a
          // K-step are consumed at a steady 16 per six MFMAs -- term major, all 40 had to precede the K-step's first fourteen MFMAs and
          // filled its first gaps.  Every accumulator still receives its terms in the same order (acs: lo x hi, then hi x lo): same bits.
#pragma unroll
          for (int ta = 0; ta < 7; ta += 2) {
            const int tb = ta + 1;
            acs[ta] = mfma_split<F16>(a[1], bq[ta][0], acs[ta]);
            if (tb < 7) acs[tb] = mfma_split<F16>(a[1], bq[tb][0], acs[tb]);
            acc[ta] = mfma_split<F16>(a[0], bq[ta][0], acc[ta]);
            if (tb < 7) acc[tb] = mfma_split<F16>(a[0], bq[tb][0], acc[tb]);
            acs[ta] = mfma_split<F16>(a[0], bq[ta][1], acs[ta]);
            if (tb < 7) acs[tb] = mfma_split<F16>(a[0], bq[tb][1], acs[tb]);
          }
        } else {
          MODE_SPLIT_TERM(acs, 2, 0)
          MODE_SPLIT_TERM(acs, 0, 2)
          MODE_SPLIT_TERM(acs, 1, 1)
          MODE_SPLIT_TERM(acs, 1, 0)
          MODE_SPLIT_TERM(acs, 0, 1)
          MODE_SPLIT_TERM(acc, 0, 0)
        }
#undef MODE_SPLIT_TERM
#pragma unroll
        for (int i = 0; i < (F16 ? 21 : 42); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (F16) {
            // a K-step's work dealt evenly over its 21 MFMA gaps -- the scheduler fills the groups greedily, so the allowance per gap
            // has to be the AVERAGE, not a ceiling (9 + 2 + 2 per gap put 11 instructions into each of a K-step's first six gaps and
            // none into the other fifteen): K-steps 0 / 1 carry 40 shifts, 14 fragment reads and 18 / 8 loads; K-steps 2 / 3 the 40
            // shifts, the split of 7 / 6 staged items (6 instructions each), the fragment reads and ~8 LDS stores
            if (ks < 2) {  // (the builtin takes literal constants: `ks` is a loop index the unroller resolves, not a constant expression)
              __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
              __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            } else {
              __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);
            }
          } else {
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      lds_barrier();
    }
  }

  float* pb = part + (((long long)s * d.MTo + ob) * d.MTc + cb) * (27 * 1024);
#pragma unroll
  for (int t7 = 0; t7 < 7; ++t7) {
    const int tap = t7 < 6 ? 3 * (wave + 4 * (t7 / 3)) + t7 % 3 : 24 + wave;
    if (tap < 27) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = (q & 3) + 8 * (q >> 2) + 4 * half;
        pb[tap * 1024 + i * 32 + (lane & 31)] = F16 ? (acc[t7][q] + acs[t7][q]) * unscale : acc[t7][q] + acs[t7][q];
      }
    }
  }
}

}  // namespace

namespace mode {

int conv3d_bww_split_launch(const float* gy, const float* x, float* part, const WgradSplitDims& d, hipStream_t st, const char* who,
                            const float* amax_g, const float* amax_x) {
  if (amax_g) {  // the two-piece fp16 arithmetic
    int rc = allow_lds(conv3d_bww_split_kernel<true>, LDS_BYTES, who);
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL(conv3d_bww_split_kernel<true>, dim3(d.S, d.MTo, d.MTc), dim3(NT), LDS_BYTES, st, gy, x, part, d, amax_g, amax_x);
    return check_launch(who);
  }
  int rc = allow_lds(conv3d_bww_split_kernel<false>, LDS_BYTES, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(conv3d_bww_split_kernel<false>, dim3(d.S, d.MTo, d.MTc), dim3(NT), LDS_BYTES, st, gy, x, part, d, nullptr, nullptr);
  return check_launch(who);
}

}  // namespace mode
