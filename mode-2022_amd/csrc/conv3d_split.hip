// 3x3x3 stride-1 convolution (pad 1) of the 3D regulariser on the bf16 matrix pipe with fp32 operands: every fp32 value is split
// EXACTLY into three bf16 pieces,  a = a1 + a2 + a3  (a1 = rne(a), a2 = rne(a - a1), a3 = rne(a - a1 - a2): 24 mantissa bits),
// and a product a*b is the sum of the six partial products of weight >= 2^-16 relative to |a||b|,
//     a3*b1 + a1*b3 + a2*b2 + a2*b1 + a1*b2 + a1*b1        (dropped: a2*b3, a3*b2, a3*b3 <= 3 * 2^-24 |a||b|),
// each exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16 and added smallest first.  The result carries the rounding of an
// fp32 convolution (tools/experiments/conv3d_bf16x6.hip measures it against an exact evaluation: max 4.5e-6 / rms 4.3e-7 at
// |y| <= 4.8, a sequential fp32 fma loop 4.7e-6 / 5.1e-7) at 6 / 16 of the bf16 MFMA rate = 2.7 x the fp32 MFMA rate.
//
// Reference: the same nn.Conv3d layers as conv3d.hip (models/submodule.py:20-22, models/mode_disparity.py:11-46, 66-80).
//
// Layout.  D[i = o][j = 32 consecutive w] as in conv3d.hip, NCDHW kept.  GEMM-K of one MFMA = 16 = 8 input channels x 2 taps
// (lanes 0..31 carry tap 2p, lanes 32..63 tap 2p+1 of tap pair p; 27 taps = 13 pairs + 1 half-empty pair), so an LDS chunk is 8
// channels deep: [3 pieces][4 x 10 haloed rows x 34 w, padded to 1 536] of uint4 (= 8 channels of bf16), 73 728 B for a 2 x 8-row
// output tile, double-buffered (147 KB: one workgroup of 4 waves per CU, one wave per SIMD, 256 VGPRs + 130 AGPRs).
// Every wave runs ONE instruction stream per chunk of 336 MFMAs (4 output rows x 14 pairs x 6 terms) that also carries, between
// the MFMAs (sched_group_barrier: 1 MFMA, <= 3 VALU, 1 LDS read), everything else:
//   * the fragment reads of the next pair (12 x ds_read_b128) and the weight fragments 6 pairs ahead (ring of 7, from L2);
//   * the loads of the NEXT chunk (8 channels x 6 positions per thread), position k under pair k, and their split
//     (v_cvt_pk_bf16_f32) + 3 x ds_write_b128 under pair k + 8;
// one LDS-only barrier per chunk.  What was measured on the way (32->32 at 48x256x128, B = 2, fp32 MFMA kernel 1.43 ms):
//   two workgroups of 4 waves per CU, stage -> barrier -> MFMA phases                                      1.00 ms
//   8 waves with fixed roles (4 matrix + 4 staging, one of each per SIMD)                                    0.89 ms: both roles
//       slow each other down on their shared SIMD (matrix wave 13.6k -> 21.2k cycles per chunk, staging 5.8k -> 23.2k)
//   one role, staging in the MFMA stream, 48 loads at the chunk start                                       0.92 ms: every CU is at
//       the same point of its chunk, the loads arrive as one burst and take a chunk time to drain (same loads from cache: 0.74)
//   loads spread over pairs 0..5, split under pairs 8..13 (this file)                                       0.83 ms = 209 TFLOP/s
// Workgroups are persistent (one per CU), walk an XCD-contiguous tile range, and the chunk stream runs across tile boundaries.
// Weights are split and packed once per launch ([m][chunk][pair][piece][lane] uint4, L2-resident).
#include "common.h"

#include "bn_internal.h"
#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef MODE_SPLIT_NT
#define MODE_SPLIT_NT 256  // (512: two waves per SIMD, two rows each -- measured in round 5)
#endif
constexpr int NT = MODE_SPLIT_NT;
// Two arithmetics (template parameter F16 of the kernels below):
//   false  three bf16 pieces per fp32 value, six MFMAs per product (the product arithmetic: 24 mantissa bits for every element);
//   true   two fp16 pieces (v_cvt_pk_f16_f32, round to nearest even; the remainder a - a1 is exact in fp32), three MFMAs per product
//          (a1b1, a1b2, a2b1: 2^-22 per product).  fp16's range is narrow: both operands are multiplied by a power of two that brings
//          their tensor's largest magnitude (a device scalar the caller provides: mode_abs_max) to [2^14, 2^15), and the accumulators by
//          the inverse -- all exact.  Elements more than ~2^17 below their tensor's maximum lose relative precision.
template <bool F16> struct Arith {
  static constexpr int NP = F16 ? 2 : 3;     // pieces per fp32 value
  static constexpr int NTERM = F16 ? 3 : 6;  // MFMAs per product
};
constexpr int TD = 2, ID = TD + 2, IW = 34;
// Tile geometry for TH output rows per depth plane: 8 (every instantiation of rounds 2-5) or 16 (round 6, the fp16 plain-store
// instantiation at volumes with enough tiles: twice the MFMAs per chunk for 1.8 x the staged positions and the same 28 weight
// fragments -- 2.3 instead of 3.1 other instructions beside an MFMA; two 80 KB buffers are exactly the CU's 160 KB of LDS)
template <int TH_>
struct Geo {
  static constexpr int TH = TH_, IH = TH_ + 2;
  static constexpr int ROWS = ID * IH;                 // 40 (72) haloed rows
  static constexpr int ITEMS = ROWS * IW;              // 1 360 (2 448) positions per chunk
  static constexpr int KIT = (ITEMS + NT - 1) / NT;    // 6 (10) positions per thread
  static constexpr int PIECE = KIT * NT;               // 1 536 (2 560): positions per piece incl. the unused tail, so that no staging store is conditional
  static constexpr int R = TD * TH_ / (NT / 64);       // 4 (8) output rows per matrix wave
  static constexpr int RB = R / 4;                     // blocks of four rows a tap pair is multiplied in
};
template <bool F16, int TH_ = 8> constexpr int buf_of() { return Arith<F16>::NP * Geo<TH_>::PIECE; }  // uint4 per buffer
constexpr int NPAIR = 14;
constexpr int WAHEAD = 6;  // weight fragments are loaded this many tap pairs ahead (3 measured the same, r03w)
template <bool F16, int TH_ = 8> constexpr size_t lds_bytes() { return 2 * (size_t)buf_of<F16, TH_>() * sizeof(uint4); }  // 147 456 B (98 304 B; 163 840 B)

struct SDims {
  int B, K, Co, D, H, W;  // K = reduction channels of this GEMM, Co = its output channels
  int nWt, nHt, nDt;
  int MT, NCHUNK;
  int ntiles;
  int TH;  // output rows per depth plane of a tile: 8, or 16 (Geo)
  int o0;  // first output channel of this launch (a layer with 33..64 output channels runs as two launches of 32)
};

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// (a, b) -> the packed pieces of the pair; the remainders are exact in fp32
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
    p1 = pack2(a, b);
    float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
    asm("" : "+v"(ra), "+v"(rb));
    p2 = pack2(ra, rb);
    float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = rb - __builtin_bit_cast(float, p2 & 0xffff0000u);
    asm("" : "+v"(sa), "+v"(sb));
    p3 = pack2(sa, sb);
  }
}

// 2^(14 - floor(log2 m)) for the largest magnitude m of a tensor (m * scale in [2^14, 2^15)); 1 for m = 0; magnitudes below 2^-63 are
// treated as 2^-63 (the tensor is zero for every purpose); Inf / NaN maxima give a finite scale and propagate through the products
__device__ __forceinline__ float f16_scale_of(float m) {
  const unsigned e = min(max((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu, 64u), 254u);
  return m == 0.f ? 1.f : __builtin_bit_cast(float, (268u - e) << 23);
}

// wp[(((m * NCHUNK + ch) * NPAIR + pair) * 3 + piece) * 64 + lane] = 8 bf16: piece of Wsrc(o = m*32 + (lane & 31), c = ch*8 + j,
// tap = 2 * pair + (lane >> 5)), j = 0..7; zero for tap 27, o >= rows, c >= K.  flip / fold as pack_w3d (conv3d.hip): flip 0 forward
// (w is (rows, K, 27)), flip 1 backward-data (w is (K, rows, 27), taps mirrored); fold: row o scaled by the folded BatchNorm scale
// and the shifts written to the floats at wp + total.
// F16: the values are multiplied by the weight tensor's power-of-two scale (amax_w[0] = its largest magnitude) before the split.
template <bool F16>
__global__ void pack_w3d_split(const float* __restrict__ w, uint4* __restrict__ wp, int rows, int K, int MT, int NCHUNK, int flip,
                               int fold, mode_bn_epilogue bn, const float* __restrict__ amax_w) {
  constexpr int NP = Arith<F16>::NP;
  const float sw = F16 ? f16_scale_of(mode::absmax_load(amax_w)) : 1.f;
  const long long total = (long long)MT * NCHUNK * NPAIR * 64;
  if (fold && blockIdx.x == 0) {
    float* shifts = reinterpret_cast<float*>(wp + total * NP);
    for (int o = threadIdx.x; o < rows; o += blockDim.x) shifts[o] = fold == 1 ? fold_shift(bn, o) : 0.f;  // (fold 2: zero shifts, no scale --
  }                                                                                                       // the accumulate form)
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int pair = (int)(r % NPAIR);
    r /= NPAIR;
    const int ch = (int)(r % NCHUNK);
    const int m = (int)(r / NCHUNK);
    const int o = m * 32 + (lane & 31);
    const int tap = 2 * pair + (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ch * 8 + j;
      v[j] = 0.f;
      if (o < rows && c < K && tap < 27)
        v[j] = flip == 0 ? w[((long long)o * K + c) * 27 + tap] : w[((long long)c * rows + o) * 27 + (26 - tap)];
      if (fold == 1 && o < rows) v[j] *= fold_scale(bn, o);
      if (F16) v[j] *= sw;
    }
    uint32_t q1[4], q2[4], q3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2<F16>(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
    uint4* dst = wp + (idx - lane) * NP + lane;
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
    if (NP == 3) dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  }
}

// Workgroup barrier that orders LDS accesses only: __syncthreads() also drains the vector-memory counter, i.e. waits for the weight
// fragments already requested for the next chunk and for the output stores of a finished tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool F16>
__device__ __forceinline__ f32x16 mfma_split(uint4 a, uint4 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__host__ __device__ constexpr int tap_off(int tap, int IH) {  // LDS position offset of a tap inside the haloed tile
  return (tap / 9) * (IH * IW) + ((tap / 3) % 3) * IW + tap % 3;
}

// EPI: 0 plain store; 1 folded-BatchNorm shift (+ ReLU); 2 shift + residual add (+ ReLU) -- the eval-mode epilogues
// EPI 3 (training): plain store + BatchNorm batch statistics of the layer's output taken from the accumulators -- per output channel
// sum(y - K) and sum((y - K)^2) with the pivot K[c] = stat_pivot[c] (anything near the data keeps the shifted sums from cancelling;
// first_voxel_kernel below puts the layer's own first output value there: a function of x and w only, so the result does not depend on
// BatchNorm's running state), accumulated per lane over all tiles of the workgroup, reduced over lanes and waves at the end and
// written as ONE partial pair per (channel, workgroup): stats[(c * gridDim.x + blockIdx.x) * 2 + {0, 1}]; the pivots sit behind them at
// stats[2 * Co * gridDim.x + c] -- the layout bn_apply_kernel reduces (mode_bn_train_fwd_prestats).  Saves the statistics pass over y.
__global__ __launch_bounds__(64) void first_voxel_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ pivot,
                                                         int K, int Co, int D, int H, int W) {
  // y[b = 0][o][0][0][0]: taps (kd, kh, kw) in {1, 2}^3 (the others fall into the zero padding), in plain fp32.  One wave per output
  // channel, the (channel, tap) terms spread over the lanes (a single thread walking them is 2 * 8 K dependent loads: 0.25 ms)
  const int o = blockIdx.x, lane = threadIdx.x;
  const long long HW = (long long)H * W, DHW = (long long)D * HW;
  float acc = 0.f;
  for (int item = lane; item < K * 8; item += 64) {
    const int c = item >> 3, t = item & 7;
    const int kd = 1 + (t >> 2), kh = 1 + ((t >> 1) & 1), kw = 1 + (t & 1);
    if (kd - 1 < D && kh - 1 < H && kw - 1 < W)
      acc = __builtin_fmaf(w[((long long)o * K + c) * 27 + kd * 9 + kh * 3 + kw], x[c * DHW + (kd - 1) * HW + (kh - 1) * W + (kw - 1)], acc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) pivot[o] = acc;
}

// AMAX (the eval epilogues of the fp16 arithmetic): the largest finite magnitude of what the kernel stores goes to amax_y (the buffer of
// bn_internal.h, zeroed by the host in front of the launch) -- the operand maximum of the NEXT layer, which has no BatchNorm pass to
// take it from in eval mode.  One v_and + v_cmp + v_cndmask + v_max per stored value, one request per workgroup at the end.
template <int MT, int EPI, bool F16 = false, int TH_ = 8, bool AMAX = false>
__global__ __launch_bounds__(NT) void conv3d_split_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                          float* __restrict__ y, SDims d, Epi epi, float* __restrict__ stats,
                                                          const float* __restrict__ stat_pivot, const float* __restrict__ amax_x,
                                                          const float* __restrict__ amax_w, unsigned* __restrict__ amax_y) {
  constexpr int NP = Arith<F16>::NP, NTERM = Arith<F16>::NTERM, BUF = buf_of<F16, TH_>();
  using Ge = Geo<TH_>;
  constexpr int TH = Ge::TH, IH = Ge::IH, ITEMS = Ge::ITEMS, KIT = Ge::KIT, PIECE = Ge::PIECE, R = Ge::R, RB = Ge::RB;
  static_assert(EPI == 0 || EPI == 1 || TH_ == 8, "the tall tile exists for the plain store and the shift epilogue (a residual's values do not fit)");
  // (F16) x is multiplied by sx when it is staged, the weights by sw when they are packed, the sums by 1 / (sx sw) when they are stored
  const float sx = F16 ? f16_scale_of(mode::absmax_load(amax_x)) : 1.f;
  const float unscale = F16 ? (1.f / sx) * (1.f / f16_scale_of(mode::absmax_load(amax_w))) : 1.f;
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];  // [2][3][ITEMS]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // grid.y = the 32 * MT-channel output blocks of the layer: every y-slice is the persistent grid described above on ITS block (a
  // 64-channel layer used to be two launches; as one, the tail of the first block's tiles overlaps with the second block, and at the
  // small volumes -- 96 tiles at 12 x 64 x 32 -- both blocks run side by side on CUs that sat idle)
  d.o0 += 32 * MT * (int)blockIdx.y;
  wp += (long long)blockIdx.y * MT * d.NCHUNK * NPAIR * (64 * NP);

  // this workgroup's tiles: XCD x = blockIdx % 8 owns a contiguous tile range, its workgroups take every nwx-th tile of it
  const int nwx = gridDim.x / kNumXCD;
  const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
  const int q = d.ntiles / kNumXCD, rr = d.ntiles % kNumXCD;
  const int t_begin = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
  const int t_count = xcd < rr ? q + 1 : q;
  const int mine = slot < t_count ? (t_count - slot + nwx - 1) / nwx : 0;  // tiles of this workgroup
  const int G = mine * d.NCHUNK;                                           // chunks in its stream

  const long long HW = (long long)d.H * d.W;
  const long long DHW = (long long)d.D * HW;

  auto tile_of = [&](int k, int& b, int& d0, int& h0, int& w0) {
    int t = t_begin + slot + k * nwx;
    w0 = (t % d.nWt) * 32;
    t /= d.nWt;
    h0 = (t % d.nHt) * TH;
    t /= d.nHt;
    d0 = (t % d.nDt) * TD;
    b = t / d.nDt;
  };

  // ---- staging: position k of this thread is (depth row, row, column) = (pdz, phy, pwi)[k] of the haloed tile, in every tile
  int pdz[KIT], phy[KIT], pwi[KIT], poff[KIT];
#pragma unroll
  for (int k = 0; k < KIT; ++k) {
    const int item = min(tid + k * NT, ITEMS - 1);
    const int row = item / IW;
    pwi[k] = item - row * IW;
    pdz[k] = row / IH;
    phy[k] = row - pdz[k] * IH;
    poff[k] = pdz[k] * (int)HW + phy[k] * d.W + pwi[k];
  }
  float raw[KIT][8];
  // Loads of the next chunk: unconditional, position k under tap pair k and split under pair k + 8 -- the workgroups run in lockstep,
  // and 48 loads per thread issued at once by every CU arrive as one 20 MB burst that takes a whole chunk time to drain (measured:
  // 0.92 ms per launch against 0.74 with the same loads served from cache).  No `&&` in here: the compiler turns a chain of
  // short-circuit tests into nested branches, and a branch ends the region in which these instructions can move under the MFMAs.
  // The requests are buffer loads (round 6): the chunk's 8 channel planes are one descriptor, a channel is a scalar offset, a
  // position a 32-bit lane offset, and a position in the zero padding is an offset beyond the descriptor -- it reads as zero.  That
  // took 48 64-bit vector adds and 54 selects out of a chunk's 168 MFMA gaps (fp16 arithmetic: 4.4 -> 3.3 other instructions per gap).
  // The staging walks its tiles incrementally -- (sb, sd, sh, sw) in tile units, advanced by the workgroup's stride of nwx tiles with
  // carries when the chunk index wraps: the division of a tile index by three runtime extents was ~110 scalar instructions per chunk
  // in ONE gap.
  int jw, jh, jd, jb, sw, sh, sd, sb, s_ch = 0;
  {
    int t = nwx;
    jw = t % d.nWt;
    t /= d.nWt;
    jh = t % d.nHt;
    t /= d.nHt;
    jd = t % d.nDt;
    jb = t / d.nDt;
    t = t_begin + slot;
    sw = t % d.nWt;
    t /= d.nWt;
    sh = t % d.nHt;
    t /= d.nHt;
    sd = t % d.nDt;
    sb = t / d.nDt;
  }
  unsigned soff[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) soff[c] = (unsigned)c * (unsigned)DHW * 4u;
  __amdgpu_buffer_rsrc_t st_rs = buf_rsrc(x, 0);
  int st_base = 0, st_d0 = 0, st_h0 = 0, st_w0 = 0;
  auto stage_advance = [&](int step) {  // step = 1: the next chunk of this workgroup's stream; 0: stay (after the last one)
    s_ch += step;
    const int wrap = s_ch >= d.NCHUNK ? 1 : 0;
    s_ch = wrap ? 0 : s_ch;
    sw += wrap ? jw : 0;
    int c = sw >= d.nWt ? 1 : 0;
    sw -= c ? d.nWt : 0;
    sh += (wrap ? jh : 0) + c;
    c = sh >= d.nHt ? 1 : 0;
    sh -= c ? d.nHt : 0;
    sd += (wrap ? jd : 0) + c;
    c = sd >= d.nDt ? 1 : 0;
    sd -= c ? d.nDt : 0;
    sb += (wrap ? jb : 0) + c;
  };
  auto stage_begin = [&]() {  // scalar part: which tile / channels the staged chunk reads
    st_d0 = sd * TD;
    st_h0 = sh * TH;
    st_w0 = sw * 32;
    st_rs = buf_rsrc(x + ((long long)sb * d.K + s_ch * 8) * DHW, (unsigned)DHW * 32u);
    st_base = (st_d0 - 1) * (int)HW + (st_h0 - 1) * d.W + (st_w0 - 1);
  };
  auto stage_load = [&](int k) {  // the 8 channel values of position k
    const unsigned ok = (unsigned)((unsigned)(st_d0 + pdz[k] - 1) < (unsigned)d.D) & (unsigned)((unsigned)(st_h0 + phy[k] - 1) < (unsigned)d.H) &
                        (unsigned)((unsigned)(st_w0 + pwi[k] - 1) < (unsigned)d.W);
    const unsigned off = ok ? (unsigned)(st_base + poff[k]) * 4u : kBufOOB;
#pragma unroll
    for (int c = 0; c < 8; ++c) raw[k][c] = buf_load_f32(st_rs, off, soff[c]);  // (K is a multiple of 8)
  };
  // split position k of the loaded chunk and write its three pieces into buffer `buf`, in two halves (one per tap pair):
  // half 0 splits channels 0..3, half 1 channels 4..7 and stores.  Branch-free: the code sits between the MFMAs of a pair.
  uint32_t sq[3][4];  // (NP pieces used)
  auto stage_commit = [&](int buf, int k, int h) {
#pragma unroll
    for (int j = 2 * h; j < 2 * h + 2; ++j) {
      float v0 = raw[k][2 * j], v1 = raw[k][2 * j + 1];
      if (F16) {
        v0 *= sx;
        v1 *= sx;
      }  // (this file is compiled with -fno-slp-vectorize, build.py: as <2 x float> the pair's remainders lose v_fma_mix_f32 -- two
         // v_cvt_f32_f16 more per pair -- and become packed fp32 instructions, which do not overlap with MFMAs)
      split2<F16>(v0, v1, sq[0][j], sq[1][j], sq[2][j]);
    }
    if (h == 1) {
      uint4* dst = sm + buf * BUF + tid + k * NT;
#pragma unroll
      for (int p = 0; p < NP; ++p) dst[p * PIECE] = make_uint4(sq[p][0], sq[p][1], sq[p][2], sq[p][3]);
    }
  };

  f32x16 acc[MT][R];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[m][r] = (f32x16){0};
  int rowpos[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = wave * R + r;
    rowpos[r] = (row / TH) * (IH * IW) + (row % TH) * IW + (lane & 31);
  }
  const int half = lane >> 5;
  // Eval epilogues.  Loads issued in the epilogue itself stall the in-order stream once per tile (measured: 0.42 -> 0.51 ms with the
  // shifts, 0.95 ms with the residual): the 16 shifts of a lane's output channels are loaded once per kernel, and the residual
  // values of a tile are fetched under the first 8 tap pairs of its last chunk (from a dummy cached address in the other chunks:
  // a branch would split the scheduling region).
  float shiftv[MT][16], addv[MT][R][16];
  float st0[MT][16], st1[MT][16], stK[MT][16];  // (EPI == 3) running sums and pivots of this lane's 16 output channels
  if (EPI == 3) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) {
        st0[m][qq] = st1[m][qq] = 0.f;
        stK[m][qq] = stat_pivot ? stat_pivot[min(d.o0 + m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half, d.Co - 1)] : 0.f;
      }
  }
  unsigned out_mag = 0;  // (AMAX)
  const float relu_floor = (EPI == 1 || EPI == 2) && epi.relu ? 0.f : -__builtin_inff();
  if (EPI == 1 || EPI == 2) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) shiftv[m][qq] = epi.shift[min(d.o0 + m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half, d.Co - 1)];
  }
  // (EPI == 2) per tile, not per chunk: the element offset of this lane's pixel of row r inside channel 0 of the tile's sample (a cached
  // dummy element for pixels outside the volume), and once per kernel the channel part of the 16 offsets -- a request is one 32-bit add
  // on a uniform base.  (Computed in every chunk, with the tile's coordinates and 64-bit offsets, this took ~100 vector instructions at
  // the top of each chunk and the registers that made the variant spill: 13 reloads, each behind an s_waitcnt vmcnt(0).)
  unsigned ep_off[R], ep_chan[MT][16];
  int ep_b = 0;
  auto ep_tile = [&](int k) {
    int b, d0, h0, w0;
    tile_of(k, b, d0, h0, w0);
    ep_b = __builtin_amdgcn_readfirstlane(b);  // uniform: the sample's base stays in scalar registers
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = wave * R + r;
      const int gd = d0 + row / TH, gh = h0 + row % TH, gw = w0 + (lane & 31);
      const unsigned ok = (unsigned)(gd < d.D) & (unsigned)(gh < d.H) & (unsigned)(gw < d.W);
      ep_off[r] = ok ? (unsigned)(gd * (int)HW + gh * d.W + gw) : (unsigned)(lane & 31);
    }
  };
  if (EPI == 2) {
    ep_tile(0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int qq = 0; qq < 16; ++qq)
        ep_chan[m][qq] = (unsigned)min(d.o0 + m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half, d.Co - 1) * (unsigned)DHW;
  }
  const long long mstride = (long long)d.NCHUNK * NPAIR * (64 * NP);

  // weight fragments: a ring of 7 tap pairs, fetched 6 pairs ahead (their loads queue behind the 48 staging loads of a chunk)
  uint4 aring[7][MT][NP];
  auto load_a = [&](int slot7, int ch, int pair) {
    const uint4* wq = wp + ((long long)ch * NPAIR + pair) * (64 * NP) + lane;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int p = 0; p < NP; ++p) aring[slot7][m][p] = wq[m * mstride + p * 64];
  };

  if (G > 0) {
    stage_begin();
#pragma unroll
    for (int k = 0; k < KIT; ++k) stage_load(k);
#pragma unroll
    for (int k = 0; k < KIT; ++k) {
      stage_commit(0, k, 0);
      stage_commit(0, k, 1);
    }
#pragma unroll
    for (int pair = 0; pair < WAHEAD; ++pair) load_a(pair, 0, pair);
  }
  __syncthreads();

  int ch = 0, k_tile = 0;
  for (int g = 0; g < G; ++g) {
    const uint4* src = sm + (g & 1) * BUF;
    const int ch_next = ch + 1 < d.NCHUNK ? ch + 1 : 0;
    // the chunk staged under this one; after the last chunk it is staged once more into the idle buffer, which keeps the loop body
    // free of branches (a branch would pin the staging code to one spot instead of letting it spread between the MFMAs)
    stage_advance(g + 1 < G ? 1 : 0);
    stage_begin();
    // (EPI == 2) where this chunk's residual requests go: the tile's pixels in its last chunk, a cached dummy row in the others
    unsigned ep_cur[R];
    const float* ep_base = epi.add;
    if (EPI == 2) {
      const bool last = ch == d.NCHUNK - 1;
      ep_base = epi.add + (long long)(last ? ep_b : 0) * d.Co * DHW;  // (a select of the index, not of the product: no branch here)
#pragma unroll
      for (int r = 0; r < R; ++r) ep_cur[r] = last ? ep_off[r] : (unsigned)(lane & 31);
    }
    // A tap pair is multiplied in RB blocks of four rows (RB = 1: the 8-row tile, the loop of rounds 2-5; RB = 2: the 16-row tile -- the
    // same weight fragment serves both blocks, the fragment registers stay those of four rows).  idx = pair * RB + block.
    uint4 bq[2][4][NP];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int p = 0; p < NP; ++p) bq[0][r][p] = src[p * PIECE + rowpos[r] + (half ? tap_off(1, IH) : tap_off(0, IH))];
#pragma unroll
    for (int idx = 0; idx < NPAIR * RB; ++idx) {
      const int pair = idx / RB, rb = idx % RB;
      // fragments of the next block (the empty second half of the last pair reads tap 26 again: finite data under zero weights)
      if (idx + 1 < NPAIR * RB) {
        const int npair = (idx + 1) / RB, nrb = (idx + 1) % RB;
        const int t0 = 2 * npair, t1 = t0 + 1 < 27 ? t0 + 1 : 26;
        const int toff = half ? tap_off(t1, IH) : tap_off(t0, IH);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int p = 0; p < NP; ++p) bq[(idx + 1) & 1][r][p] = src[p * PIECE + rowpos[4 * nrb + r] + toff];
      }
      if (rb == 0) {
        // weights 6 pairs ahead, into the slot the previous pair has left
        if (pair + WAHEAD < NPAIR)
          load_a((pair + WAHEAD) % 7, ch, pair + WAHEAD);
        else
          load_a((pair + WAHEAD) % 7, ch_next, pair + WAHEAD - NPAIR);
        // position k of the next chunk is requested under pair k and split under pair k + (NPAIR - KIT): 8 pairs (16-row tile: 4 pairs
        // of twice the MFMAs) for the loads to land
        if (pair < KIT) stage_load(pair);
      }
      if (EPI == 2 && pair >= NPAIR - 2 * R) {  // residual values of one row, 8 of its 16 output channels, under each of the last 8 pairs
        const int r = (pair - (NPAIR - 2 * R)) / 2;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int qq = 8 * (pair & 1); qq < 8 * (pair & 1) + 8; ++qq) addv[m][r][qq] = ep_base[ep_cur[r] + ep_chan[m][qq]];
      }
      if (pair >= NPAIR - KIT) {  // (RB = 2: one half of the position's channels under each row block)
        if (RB == 1 || rb == 0) stage_commit((g + 1) & 1, pair - (NPAIR - KIT), 0);
        if (RB == 1 || rb == 1) stage_commit((g + 1) & 1, pair - (NPAIR - KIT), 1);
      }
      // smallest terms first; consecutive MFMAs go to different accumulators
#define MODE_SPLIT_TERM(PA, PB)                                                      \
  _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int r = 0; r < 4; ++r) \
      acc[m][4 * rb + r] = mfma_split<F16>(aring[pair % 7][m][PA], bq[idx & 1][r][PB], acc[m][4 * rb + r]);
      if constexpr (F16) {
        MODE_SPLIT_TERM(1, 0)
        MODE_SPLIT_TERM(0, 1)
        MODE_SPLIT_TERM(0, 0)
      } else {
        MODE_SPLIT_TERM(2, 0)
        MODE_SPLIT_TERM(0, 2)
        MODE_SPLIT_TERM(1, 1)
        MODE_SPLIT_TERM(1, 0)
        MODE_SPLIT_TERM(0, 1)
        MODE_SPLIT_TERM(0, 0)
      }
#undef MODE_SPLIT_TERM
      // one MFMA, then up to 3 vector-ALU instructions and one fragment read, 24 times: spreads the staging arithmetic and the
      // next pair's reads over the matrix instructions (5 single-issue slots fit under one 32x32x16 MFMA)
#pragma unroll
      for (int i = 0; i < MT * 4 * NTERM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        if (F16) {  // (half the MFMAs for the same staging: every gap also takes an LDS access of either kind and a load.  6 + 2 per
          __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);  // gap let the scheduler fill the first gaps of a pair with 9-16
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // instructions and leave the last five empty)
        } else {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch == d.NCHUNK - 1) {  // tile finished: D[i = o][j = w]
      int b, d0, h0, w0;
      tile_of(k_tile, b, d0, h0, w0);
      float* yb = y + (long long)b * d.Co * DHW;
      const int gw = w0 + (lane & 31);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int row = wave * R + r;
        const int gd = d0 + row / TH, gh = h0 + row % TH;
        if (gd < d.D && gh < d.H && gw < d.W) {
          const long long sp = gd * HW + (long long)gh * d.W + gw;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
              const int o = d.o0 + m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half;
              if (o < d.Co) {
                const long long idx = o * DHW + sp;
                float v = acc[m][r][qq];
                if (F16) v *= unscale;
                if (EPI == 1 || EPI == 2) v += shiftv[m][qq];
                if (EPI == 2) v += addv[m][r][qq];
                if (EPI == 3) {
                  const float dv = v - stK[m][qq];
                  st0[m][qq] += dv;
                  st1[m][qq] = __builtin_fmaf(dv, dv, st1[m][qq]);
                }
                const float vo = ((EPI == 1 || EPI == 2) && v < relu_floor) ? relu_floor : v;  // (NaN passes, as in torch.relu and the fp32 kernels)
                if (AMAX) out_mag = max(out_mag, mode::absmax_mag(vo));
                yb[idx] = vo;
              }
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][r] = (f32x16){0};
      }
      ++k_tile;
      if (EPI == 2) ep_tile(min(k_tile, max(mine - 1, 0)));
    }
    ch = ch_next;
    lds_barrier();
  }
  if (AMAX) {
    __syncthreads();
    mode::absmax_block_commit(out_mag, amax_y, reinterpret_cast<unsigned*>(sm));
  }
  if (EPI == 3) {
    // lanes 0..31 / 32..63 hold the same 16 channels each (o = (qq & 3) + 8 (qq >> 2) + 4 half): butterfly over the 32 lanes of a half,
    // then the four waves through LDS in wave order -- a fixed association
    float* red = reinterpret_cast<float*>(sm);  // [4 waves][MT * 32 channels][2]
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) {
        float a = st0[m][qq], b = st1[m][qq];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
          a += __shfl_xor(a, off, 64);
          b += __shfl_xor(b, off, 64);
        }
        if ((lane & 31) == 0) {
          const int c = m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half;
          red[(wave * MT * 32 + c) * 2] = a;
          red[(wave * MT * 32 + c) * 2 + 1] = b;
        }
      }
    __syncthreads();
    if (tid < MT * 32) {
      const int o = d.o0 + tid;
      if (o < d.Co) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int wv = 0; wv < NT / 64; ++wv) {
          a += red[(wv * MT * 32 + tid) * 2];
          b += red[(wv * MT * 32 + tid) * 2 + 1];
        }
        stats[((long long)o * gridDim.x + blockIdx.x) * 2] = a;
        stats[((long long)o * gridDim.x + blockIdx.x) * 2 + 1] = b;
      }
    }
  }
}

template <int MT>
int launch_split(const float* x, const float* wpack, float* y, SDims d, int nblocks, hipStream_t st, const char* who, Epi epi,
                 float* stats = nullptr, const float* stat_pivot = nullptr, const float* amax_x = nullptr, const float* amax_w = nullptr,
                 float* amax_y = nullptr) {
  const uint4* wp = reinterpret_cast<const uint4*>(wpack);
  const int grid = kNumCU;  // persistent, one workgroup per CU (130 KB of LDS each)
#define MODE_SPLIT_LAUNCH_TH(EPIV, F16V, THV, AMAXV, STATS, PIVOT)                                                           \
  {                                                                                                                          \
    int rc = mode::allow_lds(conv3d_split_kernel<MT, EPIV, F16V, THV, AMAXV>, lds_bytes<F16V, THV>(), who);                  \
    if (rc != MODE_OK) return rc;                                                                                            \
    hipLaunchKernelGGL((conv3d_split_kernel<MT, EPIV, F16V, THV, AMAXV>), dim3(grid, nblocks), dim3(NT), (lds_bytes<F16V, THV>()), st, x, wp, \
                       y, d, epi, STATS, PIVOT, amax_x, amax_w, reinterpret_cast<unsigned*>(amax_y));                        \
    return mode::check_launch(who);                                                                                          \
  }
#define MODE_SPLIT_LAUNCH(EPIV, F16V, STATS, PIVOT) MODE_SPLIT_LAUNCH_TH(EPIV, F16V, 8, false, STATS, PIVOT)
  if (amax_x && amax_y) {  // eval mode on the two-piece fp16 arithmetic: folded BatchNorm (+ residual) (+ ReLU), the output's maximum
    MODE_REQUIRE(!stats && epi.shift, MODE_ERR_BAD_ARG, "%s: the output maximum belongs to the eval epilogues", who);
    if (epi.add) MODE_SPLIT_LAUNCH_TH(2, true, 8, true, nullptr, nullptr)
    if (d.TH == 16) MODE_SPLIT_LAUNCH_TH(1, true, 16, true, nullptr, nullptr)
    MODE_SPLIT_LAUNCH_TH(1, true, 8, true, nullptr, nullptr)
  }
  if (amax_x) {  // the two-piece fp16 arithmetic: plain store and the accumulate form (training)
    MODE_REQUIRE(!stats && (!epi.shift || epi.add), MODE_ERR_UNSUPPORTED, "%s: the fp16 arithmetic has no BatchNorm epilogues", who);
    if (epi.shift && epi.add) MODE_SPLIT_LAUNCH(2, true, nullptr, nullptr)
    if (d.TH == 16) MODE_SPLIT_LAUNCH_TH(0, true, 16, false, nullptr, nullptr)
    MODE_SPLIT_LAUNCH(0, true, nullptr, nullptr)
  }
  if (stats) MODE_SPLIT_LAUNCH(3, false, stats, stat_pivot)
  if (epi.shift && epi.add) MODE_SPLIT_LAUNCH(2, false, nullptr, nullptr)
  if (epi.shift) MODE_SPLIT_LAUNCH(1, false, nullptr, nullptr)
  MODE_SPLIT_LAUNCH(0, false, nullptr, nullptr)
#undef MODE_SPLIT_LAUNCH
#undef MODE_SPLIT_LAUNCH_TH
}

// out[0] = the largest FINITE |x[i]| as an unsigned maximum over the bit patterns of the magnitudes (order-independent: the same bits
// however the blocks are scheduled).  Inf and NaN elements do not take part: the scale has to fit the finite data, and a non-finite
// element then stays non-finite through scaling, split and MFMA -- inside its own receptive field only, as in the bf16 arithmetic.
__device__ __forceinline__ unsigned finite_mag(unsigned bits) {
  const unsigned v = bits & 0x7fffffffu;
  return v < 0x7f800000u ? v : 0u;
}
__global__ __launch_bounds__(256) void abs_max_kernel(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
  unsigned m = 0;
  const long long n4 = n >> 2, stride = (long long)gridDim.x * blockDim.x;
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {  // four 16-byte requests in flight per lane (one at a time: 3.4 TB/s)
    const uint4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    m = max(max(m, finite_mag(v0.x)), max(max(finite_mag(v0.y), finite_mag(v0.z)), finite_mag(v0.w)));
    m = max(max(m, finite_mag(v1.x)), max(max(finite_mag(v1.y), finite_mag(v1.z)), finite_mag(v1.w)));
    m = max(max(m, finite_mag(v2.x)), max(max(finite_mag(v2.y), finite_mag(v2.z)), finite_mag(v2.w)));
    m = max(max(m, finite_mag(v3.x)), max(max(finite_mag(v3.y), finite_mag(v3.z)), finite_mag(v3.w)));
  }
  for (; i < n4; i += stride) {
    const uint4 v = x4[i];
    m = max(max(m, finite_mag(v.x)), max(max(finite_mag(v.y), finite_mag(v.z)), finite_mag(v.w)));
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    m = max(m, finite_mag(__builtin_bit_cast(unsigned, x[i])));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  // (one request per block, into one of 128 words of different cache lines: bn_internal.h has the layout and the measurements)
  __shared__ unsigned sh[4];
  mode::absmax_block_commit(m, out, sh);
}

// One block per tensor: out[j] (MODE_BN_ABSMAX_FLOATS words) = the maximum buffer of tensor j -- word 0 the value, everything else zero
// (the layout's reading is "the maximum over word 0 and the 128 slots").  For the WEIGHTS of a model: tens of small tensors whose
// maxima a training step needs before its first convolution -- one launch instead of a fill + a pass per tensor.
__global__ __launch_bounds__(1024) void abs_max_batch_kernel(const float* const* __restrict__ ptrs, const long long* __restrict__ sizes,
                                                             unsigned* __restrict__ out) {
  const int j = blockIdx.x;
  const float* x = ptrs[j];
  const long long n = sizes[j];
  unsigned m = 0;
  long long i = threadIdx.x;
  for (; i + 3072 < n; i += 4096) {  // four requests in flight per lane (the largest weights are ~150 k floats: one block each)
    const float a = x[i], b = x[i + 1024], c = x[i + 2048], e = x[i + 3072];
    m = max(max(m, finite_mag(__builtin_bit_cast(unsigned, a))), finite_mag(__builtin_bit_cast(unsigned, b)));
    m = max(max(m, finite_mag(__builtin_bit_cast(unsigned, c))), finite_mag(__builtin_bit_cast(unsigned, e)));
  }
  for (; i < n; i += 1024) m = max(m, finite_mag(__builtin_bit_cast(unsigned, x[i])));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  __shared__ unsigned sh[16];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  unsigned* o = out + (long long)j * MODE_BN_ABSMAX_FLOATS;
  for (int k = threadIdx.x + 1; k < MODE_BN_ABSMAX_FLOATS; k += 1024) o[k] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned r = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) r = max(r, sh[k]);
    o[0] = r;
  }
}

// The maximum buffer of an eval layer's FOLDED weights w[o][..] * scale[o] (scale = gamma / sqrt(var + eps): what pack_w3d_split splits
// when fold == 1), one block: word 0 the value, the slots zero.  n = elements per output row.
__global__ __launch_bounds__(1024) void abs_max_folded_kernel(const float* __restrict__ w, mode_bn_epilogue bn, int rows, int n,
                                                              unsigned* __restrict__ out) {
  unsigned m = 0;
  for (int i = threadIdx.x; i < rows * n; i += 1024) m = max(m, finite_mag(__builtin_bit_cast(unsigned, w[i] * fold_scale(bn, i / n))));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off, 64));
  __shared__ unsigned sh[16];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  for (int k = threadIdx.x + 1; k < MODE_BN_ABSMAX_FLOATS; k += 1024) out[k] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned r = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) r = max(r, sh[k]);
    out[0] = r;
  }
}

}  // namespace

namespace mode {

int abs_max_batch(const float* const* ptrs, const long long* sizes, int n, float* out, hipStream_t st, const char* who) {
  MODE_REQUIRE(n >= 0, MODE_ERR_BAD_ARG, "%s: negative count", who);
  if (n == 0) return MODE_OK;
  MODE_REQUIRE(ptrs && sizes && out, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  hipLaunchKernelGGL(abs_max_batch_kernel, dim3(n), dim3(1024), 0, st, ptrs, sizes, reinterpret_cast<unsigned*>(out));
  return mode::check_launch(who);
}

int abs_max(const float* x, long long n, float* out, hipStream_t st, const char* who) {
  MODE_REQUIRE(n >= 0 && out && (n == 0 || x), MODE_ERR_BAD_ARG, "%s: bad argument", who);
  MODE_REQUIRE((reinterpret_cast<size_t>(x) & 15) == 0, MODE_ERR_BAD_ARG, "%s: the tensor must be 16-byte aligned", who);
  int rc = mode::fill_words(out, 0u, MODE_BN_ABSMAX_FLOATS, st, who);
  if (rc != MODE_OK || n == 0) return rc;
  const int blocks = (int)std::min<long long>(cdiv(n, 256 * 16), 16 * kNumCU);
  hipLaunchKernelGGL(abs_max_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, st, x, n, reinterpret_cast<unsigned*>(out));
  return mode::check_launch(who);
}

size_t conv3d_split_wpack_floats(int K, int rows) {
  return (size_t)cdiv(rows, 32) * cdiv(K, 8) * NPAIR * 3 * 64 * 4 + 32 * (size_t)cdiv(rows, 32);
}

bool conv3d_split_supported(int K, int rows) { return rows > 1 && rows <= 64 && K % 8 == 0; }

int conv3d_split_stat_partials() { return kNumCU; }  // partial pairs per channel that the statistics epilogue writes

int conv3d_s1_split(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int D, int H, int W, int flip,
                    hipStream_t st, const char* who, const mode_bn_epilogue* bn, float* stats, const float* acc_in, const float* amax_x,
                    const float* amax_w, float* amax_y) {
  const float* absmax = amax_x;  // (non-null: the fp16 arithmetic)
  // eval mode on the fp16 arithmetic (bn, amax_x, amax_y; no amax_w): the weights' maximum is that of the FOLDED weights, taken when they
  // are packed and kept in the workspace behind the shifts -- the third piece's room is free there
  const bool eval16 = absmax && bn;
  MODE_REQUIRE(eval16 ? (!amax_w && amax_y && !stats && !acc_in) : ((amax_x == nullptr) == (amax_w == nullptr) && !amax_y), MODE_ERR_BAD_ARG,
               "%s: both operand maxima, or neither (eval epilogue: the input's and the output's)", who);
  MODE_REQUIRE(!(acc_in && (bn || stats)), MODE_ERR_BAD_ARG, "%s: the accumulate form takes no BatchNorm epilogue and no statistics", who);
  MODE_REQUIRE(!(absmax && stats), MODE_ERR_UNSUPPORTED, "%s: the fp16 arithmetic has no statistics epilogue", who);
  SDims d;
  d.B = B; d.K = K; d.Co = rows; d.D = D; d.H = H; d.W = W;
  d.MT = cdiv(rows, 32);
  d.NCHUNK = cdiv(K, 8);
  MODE_REQUIRE(conv3d_split_supported(K, rows), MODE_ERR_UNSUPPORTED, "%s: %d output / %d reduction channels not supported by the split kernel", who,
               rows, K);
  MODE_REQUIRE((long long)rows * D * H * W < (1ll << 31), MODE_ERR_UNSUPPORTED, "%s: a sample of the output has 2^31 elements or more", who);
  // The 16-row tile: the plain-store fp16 instantiation (and the eval epilogue without a residual), where the rows divide and the 8-row tiling has at least four rounds of
  // tiles per output block (half as many tiles: fewer would leave CUs without one)
  static const char* tall_env = getenv("MODE_SPLIT_TALL");  // (tuning override: 0 keeps the 8-row tile everywhere)
  const bool tall_ok = !tall_env || tall_env[0] != '0';
  d.TH = 8;
  if (absmax && !acc_in && !(bn && bn->add) && H % 16 == 0 && (long long)B * cdiv(D, TD) * (H / 8) * cdiv(W, 32) >= 4ll * kNumCU && tall_ok)
    d.TH = 16;
  d.nWt = cdiv(W, 32);
  d.nHt = cdiv(H, d.TH);
  d.nDt = cdiv(D, TD);
  d.ntiles = B * d.nDt * d.nHt * d.nWt;
  const long long npack = (long long)d.MT * d.NCHUNK * NPAIR * 64;
  const int NP = absmax ? 2 : 3;
  if (eval16) {
    float* wmax = wpack + npack * NP * 4 + 32 * d.MT;  // (npack * 4 >= 3 584 floats of the third piece's room: MODE_BN_ABSMAX_FLOATS fit)
    static_assert(NPAIR * 64 * 4 >= MODE_BN_ABSMAX_FLOATS + 64, "the folded weights' maximum does not fit behind the shifts");
    if (mode::pack_needed()) {
      hipLaunchKernelGGL(abs_max_folded_kernel, dim3(1), dim3(1024), 0, st, w, *bn, rows, K * 27, reinterpret_cast<unsigned*>(wmax));
      hipLaunchKernelGGL(pack_w3d_split<true>, dim3(cdiv(npack, 256)), dim3(256), 0, st, w, reinterpret_cast<uint4*>(wpack), rows, K, d.MT, d.NCHUNK,
                         flip, 1, *bn, wmax);
    }
    int rc = mode::absmax_begin(amax_y, st, who);
    if (rc != MODE_OK) return rc;
    d.o0 = 0;
    return launch_split<1>(x, wpack, y, d, d.MT, st, who, make_epi(bn, wpack + npack * NP * 4), nullptr, nullptr, amax_x, wmax, amax_y);
  }
  if (mode::pack_needed()) {
    if (absmax)
      hipLaunchKernelGGL(pack_w3d_split<true>, dim3(cdiv(npack, 256)), dim3(256), 0, st, w, reinterpret_cast<uint4*>(wpack), rows, K, d.MT, d.NCHUNK,
                         flip, acc_in ? 2 : 0, mode_bn_epilogue(), amax_w);
    else
      hipLaunchKernelGGL(pack_w3d_split<false>, dim3(cdiv(npack, 256)), dim3(256), 0, st, w, reinterpret_cast<uint4*>(wpack), rows, K, d.MT, d.NCHUNK,
                         flip, bn ? 1 : acc_in ? 2 : 0, bn ? *bn : mode_bn_epilogue(), nullptr);
  }
  Epi epi = make_epi(bn, wpack + npack * NP * 4);
  if (acc_in) {  // y = conv(x) + acc_in: the residual epilogue with zero shifts ((v + 0) + a is a + v exactly)
    epi.shift = wpack + npack * NP * 4;
    epi.add = acc_in;
    epi.relu = 0;
  }
  // 32 output channels per workgroup: the second half of a 64-channel layer stages the input a second time, which at 6 / 16 of the
  // matrix rate still beats the fp32 kernel with two output tiles (64 -> 64 at 24 x 128 x 64: 0.41 ms against 0.70); the halves are
  // the y-slices of ONE launch
  d.o0 = 0;
  if (stats && !bn) {  // the pivots of the statistics epilogue: behind the partial pairs, where bn_apply_kernel looks for them
    float* pivot = stats + 2LL * rows * conv3d_split_stat_partials();
    hipLaunchKernelGGL(first_voxel_kernel, dim3(rows), dim3(64), 0, st, x, w, pivot, K, rows, D, H, W);
    return launch_split<1>(x, wpack, y, d, d.MT, st, who, epi, stats, pivot);
  }
  return launch_split<1>(x, wpack, y, d, d.MT, st, who, epi, nullptr, nullptr, amax_x, amax_w);
}

}  // namespace mode
