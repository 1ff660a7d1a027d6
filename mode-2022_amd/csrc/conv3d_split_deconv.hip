// ConvTranspose3d k3 s2 p1 op1 (hourglass conv5 / conv6, models/mode_disparity.py:23-25) -- and with it the input gradient of the
// stride-2 convolutions conv1 / conv3 (:17-19), which is the same operator on the same weight layout -- on the bf16 matrix pipe with
// fp32 operands split exactly into three bf16 pieces: the arithmetic and the structure of conv3d_split.hip (read that file first).
//
//   y[o][2 q + p] = sum_c sum_{taps of parity class p} w[c][o][k] * x[c][q + s(k)]        (q: low-resolution voxel, p in {0,1}^3)
// Per axis an even output index takes kernel index 1 from input q, an odd one kernel index 0 from input q + 1 and kernel index 2
// from input q: the 27 taps fall into 8 parity classes of 1, 2, 2, 2, 4, 4, 4, 8 taps, i.e. 14 tap PAIRS (one half-empty) -- so K of one
// v_mfma_f32_32x32x16_bf16 is again 8 input channels x 2 taps and a chunk is 14 pairs x 6 terms, only that a pair adds into the
// accumulator of ITS class: D_class[i = o][j = 32 consecutive low-resolution columns].
// The classes are split between two waves: set 0 = classes {0, 6, 7}, set 1 = classes {1, 2, 3, 4, 5}, seven pairs each.  A wave owns
// two low-resolution rows x one 32-channel output tile x its class set: at most 10 accumulators (160 registers), 84 MFMAs per chunk
// against 21 KB of weight fragments (all eight classes in one wave: 256 accumulator registers, 110 spilled, and slower than the fp32
// kernel), alternating between its two rows so that consecutive MFMAs never share an accumulator.  No reduction between waves:
// the classes are different output voxels.
// The LDS tile is tiny: 2 x (TH + 1) rows x 33 columns of the low-resolution input per 8-channel chunk (198 / 330 positions) against
// 336 MFMAs per workgroup and chunk: the staging sits inside the MFMA stream and the buffer is double.
// Output: the classes pw = 0 / 1 of (pd, ph) = (0, 1), (1, 0), (1, 1) leave as one float2 per lane (columns 2 j, 2 j + 1); classes 0 and
// 1 live in different waves and are stored one float each.
#include "common.h"

#include <type_traits>

#include "bn_internal.h"
#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 256;
constexpr int IW = 33;
constexpr int NPAIR = 14;
constexpr int KIT = 2;
constexpr int PIECE = KIT * NT;  // 512 >= 2 x 5 x 33 positions
constexpr int BUF = 3 * PIECE;
constexpr size_t LDS_BYTES = 2 * (size_t)BUF * sizeof(uint4);  // 49 152 B

// kernel taps (kd * 9 + kh * 3 + kw; -1 = empty) of the 14 tap pairs in packed order: pairs 0..6 = set 0 (class 0: one pair, class 6:
// two, class 7: four), pairs 7..13 = set 1 (classes 1, 2: one pair each, class 3: two, class 4: one, class 5: two)
__device__ __forceinline__ int pair_tap(int q, int h) {
  constexpr int a[NPAIR] = {13, 1, 19, 0, 6, 18, 24, 12, 10, 9, 15, 4, 3, 21};
  constexpr int b[NPAIR] = {-1, 7, 25, 2, 8, 20, 26, 14, 16, 11, 17, 22, 5, 23};
  int t = -1;
#pragma unroll
  for (int p = 0; p < NPAIR; ++p)
    if (p == q) t = h ? b[p] : a[p];
  return t;
}

struct DcDims {
  int B, K, Co, D, H, W;  // low-resolution input volume; K = its channels (reduction), Co = output channels
  int nWt, nHt, NCHUNK, ntiles;
};

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
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// wp[(((m * NCHUNK + ch) * NPAIR + pair) * 3 + piece) * 64 + lane] = 8 bf16: piece of W[c = ch*8 + j][o = m*32 + (lane & 31)]
// [tap = pair_tap(pair, lane >> 5)], j = 0..7, from the (K, Co, 27) weight of the transposed convolution; zero for the empty tap, o >= Co, c >= K
// fold: output channel o scaled by the folded BatchNorm scale, the shifts written to the floats at wp + 3 * total (eval mode)
__global__ void pack_w3d_deconv_split(const float* __restrict__ w, uint4* __restrict__ wp, int K, int Co, int MTr, int NCHUNK, int fold,
                                      mode_bn_epilogue bn) {
  const long long total = (long long)MTr * NCHUNK * NPAIR * 64;
  if (fold && blockIdx.x == 0) {
    float* shifts = reinterpret_cast<float*>(wp + total * 3);
    for (int o = threadIdx.x; o < Co; o += blockDim.x) shifts[o] = fold == 1 ? fold_shift(bn, o) : 0.f;  // (fold 2: the accumulate form)
  }
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int pair = (int)(r % NPAIR);
    r /= NPAIR;
    const int ch = (int)(r % NCHUNK);
    const int m = (int)(r / NCHUNK);
    const int o = m * 32 + (lane & 31);
    const int tap = pair_tap(pair, lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ch * 8 + j;
      v[j] = (o < Co && c < K && tap >= 0) ? w[((long long)c * Co + o) * 27 + tap] : 0.f;
      if (fold == 1 && o < Co) v[j] *= fold_scale(bn, o);
    }
    uint32_t q1[4], q2[4], q3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
    uint4* dst = wp + (idx - lane) * 3 + lane;
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
    dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  }
}

// MT = output-channel tiles per launch.  MT = 2 (33..64 output channels): waves = 2 tiles x 2 class sets on a tile of 2 rows;
// MT = 1: waves = 2 class sets x 2 row pairs on a tile of 4 rows.  One low-resolution depth plane per tile (+ its halo plane).
// EPI: the eval-mode epilogue (folded-BatchNorm shift, optional residual, optional ReLU) as its own instantiation; needs whole 32-channel
// output tiles (Co % 32 == 0: no test per channel between the batched requests and their stores).
template <int MT, bool EPI>
__global__ __launch_bounds__(NT, 1) void deconv3d_split_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                                float* __restrict__ y, DcDims d, Epi epi) {
  constexpr int TH = 4 / MT, IH = TH + 1;
  constexpr int ITEMS = 2 * IH * IW;
  static_assert(ITEMS <= PIECE, "tile does not fit the staging map");
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];  // [2][3][PIECE]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m = MT == 2 ? (wave & 1) : 0;
  const int set = MT == 2 ? (wave >> 1) : (wave & 1);  // class set of this wave
  const int rp = MT == 2 ? 0 : (wave >> 1);            // its row pair: tile rows 2 rp, 2 rp + 1
  const int half = lane >> 5;

  const int nwx = gridDim.x / kNumXCD;
  const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
  const int q = d.ntiles / kNumXCD, rr = d.ntiles % kNumXCD;
  const int t_begin = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
  const int t_count = xcd < rr ? q + 1 : q;
  const int mine = slot < t_count ? (t_count - slot + nwx - 1) / nwx : 0;
  const int G = mine * d.NCHUNK;

  const long long HW = (long long)d.H * d.W;
  const long long DHW = (long long)d.D * HW;
  const int oH = 2 * d.H, oW = 2 * d.W;
  const long long oHW = (long long)oH * oW, oDHW = 2 * d.D * oHW;

  auto tile_of = [&](int k, int& b, int& d0, int& h0, int& w0) {
    int t = t_begin + slot + k * nwx;
    w0 = (t % d.nWt) * 32;
    t /= d.nWt;
    h0 = (t % d.nHt) * TH;
    t /= d.nHt;
    d0 = t % d.D;
    b = t / d.D;
  };

  int pdz[KIT], phy[KIT], pwx[KIT], poff[KIT];
#pragma unroll
  for (int k = 0; k < KIT; ++k) {
    const int item = min(tid + k * NT, ITEMS - 1);
    const int r = item / IW;
    pwx[k] = item - r * IW;
    pdz[k] = r / IH;
    phy[k] = r - pdz[k] * IH;
    poff[k] = pdz[k] * (int)HW + phy[k] * d.W + pwx[k];
  }
  float raw[KIT][8];
  // Staging as in conv3d_split.hip since round 6: buffer loads -- the chunk's 8 channel planes one descriptor, a channel a
  // scalar offset, a position beyond the volume an out-of-range lane offset that reads as zero -- and the tiles walked incrementally
  // (sb, sd, sh, sw in tile units; the depth extent of a tile is one plane) instead of a division of the tile index per chunk.
  int jw, jh, jd, jb, sw, sh, sd, sb, s_ch = 0;
  {
    int t = nwx;
    jw = t % d.nWt;
    t /= d.nWt;
    jh = t % d.nHt;
    t /= d.nHt;
    jd = t % d.D;
    jb = t / d.D;
    t = t_begin + slot;
    sw = t % d.nWt;
    t /= d.nWt;
    sh = t % d.nHt;
    t /= d.nHt;
    sd = t % d.D;
    sb = t / d.D;
  }
  unsigned soff[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) soff[c] = (unsigned)c * (unsigned)DHW * 4u;
  __amdgpu_buffer_rsrc_t st_rs = buf_rsrc(x, 0);
  int st_base = 0, st_d0 = 0, st_h0 = 0, st_w0 = 0;
  auto stage_advance = [&](int step) {
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
    c = sd >= d.D ? 1 : 0;
    sd -= c ? d.D : 0;
    sb += (wrap ? jb : 0) + c;
  };
  auto stage_begin = [&]() {
    st_d0 = sd;
    st_h0 = sh * TH;
    st_w0 = sw * 32;
    st_rs = buf_rsrc(x + ((long long)sb * d.K + s_ch * 8) * DHW, (unsigned)DHW * 32u);
    st_base = st_d0 * (int)HW + st_h0 * d.W + st_w0;
  };
  auto stage_load = [&](int k) {
    const unsigned ok = (unsigned)(st_d0 + pdz[k] < d.D) & (unsigned)(st_h0 + phy[k] < d.H) & (unsigned)(st_w0 + pwx[k] < d.W);
    const unsigned off = ok ? (unsigned)(st_base + poff[k]) * 4u : kBufOOB;
#pragma unroll
    for (int c = 0; c < 8; ++c) raw[k][c] = buf_load_f32(st_rs, off, soff[c]);
  };
  uint32_t sq[3][4];
  auto stage_commit = [&](int buf, int k, int h) {
#pragma unroll
    for (int j = 2 * h; j < 2 * h + 2; ++j) split2(raw[k][2 * j], raw[k][2 * j + 1], sq[0][j], sq[1][j], sq[2][j]);
    if (h == 1) {
      uint4* dst = sm + buf * BUF + tid + k * NT;
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[p * PIECE] = make_uint4(sq[p][0], sq[p][1], sq[p][2], sq[p][3]);
    }
  };

  f32x16 acc[2][5];  // [row][class slot]: set 0: classes 0, 6, 7; set 1: classes 1, 2, 3, 4, 5
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) acc[r][c] = (f32x16){0};
  int rowpos[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) rowpos[r] = (2 * rp + r) * IW + (lane & 31);
  const long long mstride = (long long)d.NCHUNK * NPAIR * 192;
  const uint4* wpm = wp + m * mstride + set * (7 * 192);
  unsigned out_mag = 0;  // (EPI) the largest finite magnitude this thread stored: the next eval layer's operand maximum (epi.amax)
  float shv[16];  // (EPI) the folded shifts of this lane's 16 output channels, once per kernel
  if (EPI) {
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) shv[qq] = epi.shift[m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half];
  }
  // (Round 5 also measured the request order "weight fragments a whole chunk ahead, the staging loads of chunk g + 2 at the end of chunk
  // g" -- no wait of a chunk then reaches past the weights it needs, where the in-order vector-memory counter otherwise drags the HBM
  // staging loads into a wait for an L2 weight fragment: same-box A/B 0.341-0.353 against 0.343-0.350 ms at 64 -> 32, 0.080 against
  // 0.077 ms at 64 -> 64 (tools/experiments/deconv_ab.sh): not what bounds the kernel, not kept.)
  uint4 aring[7][3];  // slot = pair of the set; fetched three pairs ahead
  auto load_a = [&](int i, int ch) {
    const uint4* wq = wpm + ((long long)ch * NPAIR + i) * 192 + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) aring[i][p] = wq[p * 64];
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
    for (int i = 0; i < 3; ++i) load_a(i, 0);
  }
  __syncthreads();

  // The chunk loop exists TWICE, once per class set, and a wave enters the copy of its set: with `if (set == 0) ... else ...` INSIDE one loop
  // the two paths use different accumulators (3 against 5 classes), the register allocator gave them different homes, and the join behind
  // the branch paid for it with 117 v_accvgpr_write + 91 v_accvgpr_mov + 64 v_accvgpr_read per chunk and wave -- 272 of the loop's 446
  // vector instructions beside its 84 MFMAs (round 5, tools/isa_mix.py).  Both copies execute the same barriers in the same order.
  auto run = [&](auto set_tag) {
  constexpr int SET = decltype(set_tag)::value;
  int ch = 0, k_tile = 0;
  for (int g = 0; g < G; ++g) {
    const uint4* src = sm + (g & 1) * BUF;
    const int ch_next = ch + 1 < d.NCHUNK ? ch + 1 : 0;
    stage_advance(g + 1 < G ? 1 : 0);  // (after the last chunk the same one is staged once more into the idle buffer: no branch in the body)
    stage_begin();
    uint4 bq[2][2][3];
    // One tap pair of this wave's set (I, SLOT literal: every register array index is a compile-time constant): fragment reads of the
    // next pair, the weight fragments three pairs ahead, this pair's share of the staging, then 12 MFMAs -- smallest terms first, the
    // two rows alternating.
#define MODE_DC_TERM(I, SLOT, PA, PB)                                               \
  acc[0][SLOT] = mfma_bf16(aring[I][PA], bq[(I) & 1][0][PB], acc[0][SLOT]);         \
  acc[1][SLOT] = mfma_bf16(aring[I][PA], bq[(I) & 1][1][PB], acc[1][SLOT]);
#define MODE_DC_PAIR(I, SLOT, NOFFA, NOFFB)                                                                        \
  {                                                                                                                \
    if (I + 1 < 7) {                                                                                               \
      const int toff = half ? (NOFFB) : (NOFFA);                                                                   \
      _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int p = 0; p < 3; ++p)                  \
          bq[(I + 1) & 1][r][p] = src[p * PIECE + rowpos[r] + toff];                                               \
    }                                                                                                              \
    if (I + 3 < 7)                                                                                                 \
      load_a(I + 3, ch);                                                                                           \
    else                                                                                                           \
      load_a(I + 3 - 7, ch_next);                                                                                  \
    if (I < KIT) stage_load(I);                                                                                    \
    if (I >= 7 - KIT) {                                                                                            \
      stage_commit((g + 1) & 1, I - (7 - KIT), 0);                                                                 \
      stage_commit((g + 1) & 1, I - (7 - KIT), 1);                                                                 \
    }                                                                                                              \
    MODE_DC_TERM(I, SLOT, 2, 0) MODE_DC_TERM(I, SLOT, 0, 2) MODE_DC_TERM(I, SLOT, 1, 1) MODE_DC_TERM(I, SLOT, 1, 0) \
    MODE_DC_TERM(I, SLOT, 0, 1) MODE_DC_TERM(I, SLOT, 0, 0)                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                                            \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                           \
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                           \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                           \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
  }
    if constexpr (SET == 0)
      {  // classes 0, 6, 7
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[0][r][p] = src[p * PIECE + rowpos[r] + (half ? 0 : (0 * IH + 0) * IW + 0)];
      MODE_DC_PAIR(0, 0, (1 * IH + 1) * IW + 0, (1 * IH + 0) * IW + 0)
      MODE_DC_PAIR(1, 1, (0 * IH + 1) * IW + 0, (0 * IH + 0) * IW + 0)
      MODE_DC_PAIR(2, 1, (1 * IH + 1) * IW + 1, (1 * IH + 1) * IW + 0)
      MODE_DC_PAIR(3, 2, (1 * IH + 0) * IW + 1, (1 * IH + 0) * IW + 0)
      MODE_DC_PAIR(4, 2, (0 * IH + 1) * IW + 1, (0 * IH + 1) * IW + 0)
      MODE_DC_PAIR(5, 2, (0 * IH + 0) * IW + 1, (0 * IH + 0) * IW + 0)
      MODE_DC_PAIR(6, 2, 0, 0)
      }
    else
      {  // classes 1 .. 5
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[0][r][p] = src[p * PIECE + rowpos[r] + (half ? (0 * IH + 0) * IW + 0 : (0 * IH + 0) * IW + 1)];
      MODE_DC_PAIR(0, 0, (0 * IH + 1) * IW + 0, (0 * IH + 0) * IW + 0)
      MODE_DC_PAIR(1, 1, (0 * IH + 1) * IW + 1, (0 * IH + 1) * IW + 0)
      MODE_DC_PAIR(2, 2, (0 * IH + 0) * IW + 1, (0 * IH + 0) * IW + 0)
      MODE_DC_PAIR(3, 2, (1 * IH + 0) * IW + 0, (0 * IH + 0) * IW + 0)
      MODE_DC_PAIR(4, 3, (1 * IH + 0) * IW + 1, (1 * IH + 0) * IW + 0)
      MODE_DC_PAIR(5, 4, (0 * IH + 0) * IW + 1, (0 * IH + 0) * IW + 0)
      MODE_DC_PAIR(6, 4, 0, 0)
      }
#undef MODE_DC_PAIR
#undef MODE_DC_TERM
    if (ch == d.NCHUNK - 1) {  // tile finished: D_class[i = o][j = low-resolution column]
      int b, d0, h0, w0;
      tile_of(k_tile, b, d0, h0, w0);
      const int gw = w0 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int gh = h0 + 2 * rp + r;
        if (gh < d.H && gw < d.W) {
          // running pointers, advanced by whole channel planes and made opaque after every step (left to itself the compiler computes
          // all store addresses of the two rows up front: ~200 registers next to the accumulators)
          // (running OFFSETS from y, opaque after every step -- not running pointers: an opaque pointer loses its address space and the
          // stores become flat_store, which count on both memory counters and made every chunk's first wait a vmcnt(0))
          const long long yb = ((long long)b * d.Co + m * 32 + 4 * half) * oDHW + (long long)(2 * d0) * oHW + (long long)(2 * gh) * oW + 2 * gw;
          // set 0: class 0 -> (0,0,0) one float; classes 6, 7 -> (1,1,.) float2.  set 1: class 1 -> (0,0,1); 2, 3 -> (0,1,.); 4, 5 -> (1,0,.)
          long long y1 = yb + (SET == 0 ? 0 : 1);
          long long y2a = yb + (SET == 0 ? oHW + oW : oW);
          long long y2b = yb + oHW;  // (set 1 only)
          if constexpr (!EPI) {
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
              const int o = m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half;
              asm volatile("" : "+v"(y1), "+v"(y2a), "+v"(y2b));
              if (o < d.Co) {
                y[y1] = acc[r][0][qq];
                if (SET == 0) {
                  *reinterpret_cast<float2*>(y + y2a) = make_float2(acc[r][1][qq], acc[r][2][qq]);
                } else {
                  *reinterpret_cast<float2*>(y + y2a) = make_float2(acc[r][1][qq], acc[r][2][qq]);
                  *reinterpret_cast<float2*>(y + y2b) = make_float2(acc[r][3][qq], acc[r][4][qq]);
                }
              }
              const long long step = ((qq & 3) == 3 ? 5 : 1) * oDHW;
              y1 += step;
              y2a += step;
              y2b += step;
            }
          } else {
            // eval / accumulate: (sum + shift) + residual, ReLU as torch computes it.  The residual values of a ROW (16 channels, up to 80
            // floats: the fragment and staging registers of the tap loop are free here) are requested together ahead of its stores --
            // one round trip per row; `add` may alias y for all the compiler knows, and a load next to each store waits for the store in
            // front of it (in batches of four channels: four round trips per row, 192 -> NNN us at 64 -> 32 / 24 x 128 x 64)
            const bool has_add = epi.add != nullptr;
            float r1[16];
            float2 r2a[16], r2b[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              r1[i] = 0.f;
              r2a[i] = r2b[i] = make_float2(0.f, 0.f);
            }
            if (has_add) {
              long long a1 = y1, a2a = y2a, a2b = y2b;
#pragma unroll
              for (int i = 0; i < 16; ++i) {
                asm volatile("" : "+v"(a1), "+v"(a2a), "+v"(a2b));
                r1[i] = epi.add[a1];
                r2a[i] = *reinterpret_cast<const float2*>(epi.add + a2a);
                if (SET == 1) r2b[i] = *reinterpret_cast<const float2*>(epi.add + a2b);
                const long long step = ((i & 3) == 3 ? 5 : 1) * oDHW;
                a1 += step;
                a2a += step;
                a2b += step;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
              asm volatile("" : "+v"(y1), "+v"(y2a), "+v"(y2b));
              auto fin = [&](float v, float res) {
                v = (v + shv[qq]) + res;
                v = epi.relu ? relu_nan(v) : v;
                out_mag = max(out_mag, mode::absmax_mag(v));
                return v;
              };
              y[y1] = fin(acc[r][0][qq], r1[qq]);
              *reinterpret_cast<float2*>(y + y2a) = make_float2(fin(acc[r][1][qq], r2a[qq].x), fin(acc[r][2][qq], r2a[qq].y));
              if (SET == 1) *reinterpret_cast<float2*>(y + y2b) = make_float2(fin(acc[r][3][qq], r2b[qq].x), fin(acc[r][4][qq], r2b[qq].y));
              const long long step = ((qq & 3) == 3 ? 5 : 1) * oDHW;
              y1 += step;
              y2a += step;
              y2b += step;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[r][c] = (f32x16){0};
      }
      ++k_tile;
    }
    ch = ch_next;
    lds_barrier();
  }
  };
  if (set == 0)
    run(std::integral_constant<int, 0>{});
  else
    run(std::integral_constant<int, 1>{});
  if (EPI) {
    if (epi.amax) {  // (uniform)
      __syncthreads();
      mode::absmax_block_commit(out_mag, epi.amax, reinterpret_cast<unsigned*>(sm));
    }
  }
}

}  // namespace

namespace mode {

size_t deconv3d_split_wpack_floats(int K, int Co) {  // (+ the folded BatchNorm shifts behind the fragments)
  return (size_t)cdiv(Co, 32) * cdiv(K, 8) * NPAIR * 3 * 64 * 4 + 32 * (size_t)cdiv(Co, 32);
}

bool deconv3d_split_bn_supported(int K, int Co) { return deconv3d_split_supported(K, Co) && Co % 32 == 0; }

bool deconv3d_split_supported(int K, int Co) { return Co > 1 && Co <= 64 && K > 0 && K % 8 == 0; }

// x (B, K, D, H, W), w (K, Co, 27) -> y (B, Co, 2D, 2H, 2W)
int deconv3d_split(const float* x, const float* w, float* y, float* wpack, int B, int K, int Co, int D, int H, int W, hipStream_t st,
                   const char* who, const mode_bn_epilogue* bn, const float* acc_in, float* amax_y) {
  MODE_REQUIRE(!amax_y || bn, MODE_ERR_BAD_ARG, "%s: the output maximum belongs to the eval epilogue", who);
  MODE_REQUIRE(!(acc_in && bn), MODE_ERR_BAD_ARG, "%s: the accumulate form takes no BatchNorm epilogue", who);
  MODE_REQUIRE(deconv3d_split_supported(K, Co), MODE_ERR_UNSUPPORTED, "%s: %d output / %d input channels not supported by the split kernel", who, Co, K);
  MODE_REQUIRE(!(bn || acc_in) || deconv3d_split_bn_supported(K, Co), MODE_ERR_UNSUPPORTED,
               "%s: the epilogue forms need whole 32-channel output tiles, got %d", who, Co);
  MODE_REQUIRE((long long)D * H * W * 8 * std::max(Co, 8) < (1ll << 31) && (long long)D * H * W < (1ll << 27), MODE_ERR_UNSUPPORTED,
               "%s: volume beyond the 32-bit offsets of the split kernel", who);
  DcDims d;
  d.B = B; d.K = K; d.Co = Co; d.D = D; d.H = H; d.W = W;
  const int MTr = cdiv(Co, 32), TH = 4 / MTr;
  d.nWt = cdiv(W, 32); d.nHt = cdiv(H, TH);
  d.NCHUNK = cdiv(K, 8);
  d.ntiles = B * D * d.nHt * d.nWt;
  const long long npack = (long long)MTr * d.NCHUNK * NPAIR * 64;
  if (mode::pack_needed())
    hipLaunchKernelGGL(pack_w3d_deconv_split, dim3(cdiv(npack, 256)), dim3(256), 0, st, w, reinterpret_cast<uint4*>(wpack), K, Co, MTr, d.NCHUNK,
                       bn ? 1 : acc_in ? 2 : 0, bn ? *bn : mode_bn_epilogue());
  Epi epi = make_epi(bn, wpack + npack * 3 * 4);
  if (acc_in) {  // y = deconv(x) + acc_in: the residual epilogue with zero shifts
    epi.shift = wpack + npack * 3 * 4;
    epi.add = acc_in;
    epi.relu = 0;
  }
  const bool with_epi = bn || acc_in;
  const uint4* wq = reinterpret_cast<const uint4*>(wpack);
  int rc;
  if (amax_y) {
    rc = mode::absmax_begin(amax_y, st, who);
    if (rc != MODE_OK) return rc;
    epi.amax = reinterpret_cast<unsigned*>(amax_y);
  }
#define MODE_DC_LAUNCH(MTV, EPIV)                                                                                     \
  {                                                                                                                   \
    rc = mode::allow_lds(deconv3d_split_kernel<MTV, EPIV>, LDS_BYTES, who);                                           \
    if (rc != MODE_OK) return rc;                                                                                     \
    hipLaunchKernelGGL((deconv3d_split_kernel<MTV, EPIV>), dim3(kNumCU), dim3(NT), LDS_BYTES, st, x, wq, y, d, epi);  \
  }
  if (MTr == 2) {
    if (with_epi) MODE_DC_LAUNCH(2, true) else MODE_DC_LAUNCH(2, false)
  } else {
    if (with_epi) MODE_DC_LAUNCH(1, true) else MODE_DC_LAUNCH(1, false)
  }
#undef MODE_DC_LAUNCH
  return mode::check_launch(who);
}

}  // namespace mode
