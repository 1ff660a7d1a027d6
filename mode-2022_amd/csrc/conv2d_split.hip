// Regular 3x3 Conv2d layers (stride 1, dilation 1 / 2, pad = dilation) of the extractor as fp32 convolutions on the bf16 matrix pipe:
// conv3d_split.hip one dimension down.  Operands are split exactly into three bf16 pieces when a tile is staged, a product is six
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the arithmetic and its error analysis: conv3d_split.hip).
//
// Reference: the stock nn.Conv2d 3x3 layers of convbn (models/submodule.py:13-17) in firstconv / layer1-3 (submodule.py:155-172) and
// the tap products of the folded cost volume -- cuDNN NCHW fp32 there.
//
// What differs from the 3-D kernel: K of one MFMA = 16 input CHANNELS of one tap (lanes 0..31 channels 0..7, lanes 32..63 channels
// 8..15 of the chunk) -- with 9 taps there is no tap to pair up without wasting a tenth of the MFMAs -- so an LDS chunk is 16
// channels deep, [3 pieces][2 channel octets][(16 + 2d) x (32 + 2d) haloed positions, padded to 768] of uint4 = 73 728 B for a
// 16-row x 32-column output tile, double-buffered.  Per chunk a wave issues 9 taps x 4 rows x MT tiles x 6 terms = 216 / 432 MFMAs
// and, between them, stages the next chunk: 3 positions x 16 channels per thread, position k loaded under tap k and split + stored
// under tap 6 + k.  Weight fragments come from L2 two taps ahead (ring of 3).  Persistent workgroups, XCD-contiguous tile ranges,
// chunk stream across tile boundaries, LDS-only barrier: as in conv3d_split.hip.
#include "common.h"

#include "bn_internal.h"
#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 256;
// Tile geometry for TH output rows (16, or 8 when the layer has fewer than two 16-row tiles per CU) and dilation DIL
template <int TH, int DIL>
struct Geo2 {
  static constexpr int R = TH / 4;                                   // output rows per wave
  static constexpr int IH = TH + 2 * DIL, IW = 32 + 2 * DIL;
  static constexpr int ITEMS = IH * IW;
  static constexpr int KIT = (ITEMS + NT - 1) / NT;                  // haloed positions per thread and chunk (3 / 2)
  static constexpr int PIECE = KIT * NT;                             // positions per (piece, octet) incl. the unused tail
  static constexpr int BUF = 3 * 2 * PIECE;                          // uint4 per buffer
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF * sizeof(uint4);  // 147 456 / 98 304 B
};

struct S2Dims {
  int B, K, Co, H, W;  // K = reduction channels of this GEMM, Co = its output channels
  int nWt, nHt;
  int NCHUNK;
  int ntiles;
  int o0;  // first output channel of this launch
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
// ---- F16: the two-piece fp16 arithmetic of conv3d_split.hip for the training step's 3 x 3 layers: two fp16 pieces per value,
// three v_mfma_f32_32x32x16_f16 per product (lo x hi, hi x lo, hi x hi), a power-of-two scale per operand tensor from its maximum buffer
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

// wp[(((m * NCHUNK + ch) * 9 + tap) * 3 + piece) * 64 + lane] = 8 bf16: piece of Wsrc(o = m*32 + (lane & 31), c = ch*16 + 8 * (lane >> 5)
// + j, tap), j = 0..7; zero for o >= rows, c >= K.  flip 0: Wsrc = w[o][c][tap] (forward, w is (rows, K, 9)); flip 1: w[c][o][8 - tap]
// (input gradient, w is (K, rows, 9)); fold: row o scaled by the folded BatchNorm scale, shifts written behind the fragments.
template <bool F16>
__global__ void pack_w2d_split(const float* __restrict__ w, uint4* __restrict__ wp, int rows, int K, int MT, int NCHUNK, int flip, int fold,
                               mode_bn_epilogue bn, const float* __restrict__ amax_w) {
  const float sw = F16 ? f16_scale_of(mode::absmax_load(amax_w)) : 1.f;
  const long long total = (long long)MT * NCHUNK * 9 * 64;
  if (fold && blockIdx.x == 0) {
    float* shifts = reinterpret_cast<float*>(wp + total * 3);
    for (int o = threadIdx.x; o < rows; o += blockDim.x) shifts[o] = fold == 1 ? fold_shift(bn, o) : 0.f;  // (fold 2: the accumulate form)
  }
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int tap = (int)(r % 9);
    r /= 9;
    const int ch = (int)(r % NCHUNK);
    const int m = (int)(r / NCHUNK);
    const int o = m * 32 + (lane & 31);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ch * 16 + 8 * (lane >> 5) + j;
      v[j] = 0.f;
      if (o < rows && c < K) v[j] = flip ? w[((long long)c * rows + o) * 9 + 8 - tap] : w[((long long)o * K + c) * 9 + tap];
      if (fold == 1 && o < rows) v[j] *= fold_scale(bn, o);
      if (F16) v[j] *= sw;
    }
    uint32_t q1[4], q2[4], q3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (F16) {
        split2_f16(v[2 * j], v[2 * j + 1], q1[j], q2[j]);
        q3[j] = 0u;
      } else {
        split2(v[2 * j], v[2 * j + 1], q1[j], q2[j], q3[j]);
      }
    }
    uint4* dst = wp + (idx - lane) * 3 + lane;
    dst[0] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    dst[64] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
    dst[128] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
  }
}

// EPI: 0 plain store; 1 folded-BatchNorm shift (+ ReLU); 2 shift + residual add (+ ReLU)
// F16 (EPI 0 and 2 -- the training step; EPI 1 and 2 with epi.amax -- inference since round 6): two fp16 pieces, three MFMAs
// per product; x is scaled when a tile is staged, the accumulators unscaled in front of the epilogue's additions.  The LDS and
// weight-fragment layouts keep room for three pieces.  epi.amax (eval): the largest finite magnitude the kernel stored goes there.
template <int MT, int TH, int DIL, int EPI, bool F16 = false>
__global__ __launch_bounds__(NT) void conv2d_split_kernel(const float* __restrict__ x, const uint4* __restrict__ wp, float* __restrict__ y,
                                                          S2Dims d, Epi epi, const float* __restrict__ amax_x,
                                                          const float* __restrict__ amax_w) {
  constexpr int NP = F16 ? 2 : 3;  // pieces per value
  float sx = 1.f, unscale = 1.f;
  if (F16) {
    sx = f16_scale_of(mode::absmax_load(amax_x));
    unscale = (1.f / sx) * (1.f / f16_scale_of(mode::absmax_load(amax_w)));
  }
  using G2 = Geo2<TH, DIL>;
  constexpr int R = G2::R, IW = G2::IW, ITEMS = G2::ITEMS, KIT = G2::KIT, PIECE = G2::PIECE, BUF = G2::BUF;
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];  // [2][3 pieces][2 octets][PIECE]
  // grid.y = the 32 * MT-channel output blocks of the layer, every y-slice the persistent grid on ITS block (as in conv3d_split.hip): a
  // 256-channel layer at 128 x 64 has 32 tiles per block -- one launch per block left seven CUs in eight idle
  d.o0 += 32 * MT * (int)blockIdx.y;
  wp += (long long)blockIdx.y * MT * d.NCHUNK * 9 * 192;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5;

  const int nwx = gridDim.x / kNumXCD;
  const int xcd = blockIdx.x % kNumXCD, slot = blockIdx.x / kNumXCD;
  const int q = d.ntiles / kNumXCD, rr = d.ntiles % kNumXCD;
  const int t_begin = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
  const int t_count = xcd < rr ? q + 1 : q;
  const int mine = slot < t_count ? (t_count - slot + nwx - 1) / nwx : 0;
  const int G = mine * d.NCHUNK;
  const int HWi = d.H * d.W;  // (host guarantees max(K, Co) * H * W < 2^29)

  auto tile_of = [&](int k, int& b, int& h0, int& w0) {
    int t = t_begin + slot + k * nwx;
    w0 = (t % d.nWt) * 32;
    t /= d.nWt;
    h0 = (t % d.nHt) * TH;
    b = t / d.nHt;
  };

  // ---- staging: position k of this thread is (row, column) = (phy, pwi)[k] of the haloed tile, in every tile
  int phy[KIT], pwi[KIT], poff[KIT];
#pragma unroll
  for (int k = 0; k < KIT; ++k) {
    const int item = min(tid + k * NT, ITEMS - 1);
    phy[k] = item / IW;
    pwi[k] = item - phy[k] * IW;
    poff[k] = phy[k] * d.W + pwi[k];
  }
  float raw[KIT][16];
  // Staging as in conv3d_split.hip (round 6): buffer loads -- the chunk's 16 channel planes are one descriptor, a channel a scalar
  // offset, a position a 32-bit lane offset, a position in the zero padding an offset beyond the descriptor (reads as zero: no select
  // per loaded value) -- and the tiles walked incrementally (sb, sh, sw in tile units, advanced by the workgroup's stride with carries
  // when the chunk index wraps) instead of a division of the tile index per chunk (113 scalar instructions in one MFMA gap).
  int jw, jh, jb, sw, sh, sb, s_ch = 0;
  {
    int t = nwx;
    jw = t % d.nWt;
    t /= d.nWt;
    jh = t % d.nHt;
    jb = t / d.nHt;
    t = t_begin + slot;
    sw = t % d.nWt;
    t /= d.nWt;
    sh = t % d.nHt;
    sb = t / d.nHt;
  }
  unsigned soff[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) soff[c] = (unsigned)c * (unsigned)HWi * 4u;
  __amdgpu_buffer_rsrc_t st_rs = buf_rsrc(x, 0);
  int st_base = 0, st_h0 = 0, st_w0 = 0;
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
    sb += (wrap ? jb : 0) + c;
  };
  auto stage_begin = [&]() {
    st_h0 = sh * TH;
    st_w0 = sw * 32;
    st_rs = buf_rsrc(x + ((long long)sb * d.K + s_ch * 16) * HWi, (unsigned)HWi * 64u);  // (K is a multiple of 16)
    st_base = (st_h0 - DIL) * d.W + (st_w0 - DIL);
  };
  auto stage_load = [&](int k) {  // the 16 channel values of position k: unconditional (no && : no branches)
    const unsigned ok = (unsigned)((unsigned)(st_h0 + phy[k] - DIL) < (unsigned)d.H) & (unsigned)((unsigned)(st_w0 + pwi[k] - DIL) < (unsigned)d.W);
    const unsigned off = ok ? (unsigned)(st_base + poff[k]) * 4u : kBufOOB;
#pragma unroll
    for (int c = 0; c < 16; ++c) raw[k][c] = buf_load_f32(st_rs, off, soff[c]);
  };
  auto stage_commit = [&](int buf, int k) {
    uint4* dst = sm + buf * BUF + tid + k * NT;
#pragma unroll
    for (int oct = 0; oct < 2; ++oct) {
      uint32_t sq[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v0 = raw[k][8 * oct + 2 * j], v1 = raw[k][8 * oct + 2 * j + 1];
        if constexpr (F16)
          split2_f16(v0 * sx, v1 * sx, sq[0][j], sq[1][j]);
        else
          split2(v0, v1, sq[0][j], sq[1][j], sq[2][j]);
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) dst[(2 * p + oct) * PIECE] = make_uint4(sq[p][0], sq[p][1], sq[p][2], sq[p][3]);
    }
  };

  f32x16 acc[MT][R];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[m][r] = (f32x16){0};
  int rowpos[R];
#pragma unroll
  for (int r = 0; r < R; ++r) rowpos[r] = half * PIECE + (wave * R + r) * IW + (lane & 31);
  const long long mstride = (long long)d.NCHUNK * 9 * 192;
  // Eval epilogues as in conv3d_split.hip: the shifts live in registers for the whole kernel; the residual values of a tile are
  // fetched under the last taps of its last chunk where the registers allow (64 values per lane), else in the epilogue itself.
  constexpr bool ADD_AHEAD = EPI == 2 && MT * R * 16 <= 64;
  float shiftv[MT][16], addv[ADD_AHEAD ? MT : 1][ADD_AHEAD ? R : 1][16];
  const float relu_floor = (EPI && epi.relu) ? 0.f : -__builtin_inff();
  unsigned out_mag = 0;  // (EPI on fp16: the output's maximum for the next eval layer, epi.amax)
  // (the variant with the residual in its epilogue has no registers left for 32 shifts: kept across the tap loop they were spilled, and
  // every store of the epilogue waited for its reload; it requests them with the residual values instead)
  constexpr bool SHIFT_LATE = EPI == 2 && !ADD_AHEAD;
  if (EPI && !SHIFT_LATE) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) shiftv[m][qq] = epi.shift[min(d.o0 + m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half, d.Co - 1)];
  }

  // (ADD_AHEAD) per tile, not per chunk: the element offset of this lane's pixel of row r inside channel 0 of the tile's sample (a cached
  // dummy element for pixels outside the image), and once per kernel the channel part of the 16 offsets -- a request is one 32-bit add on
  // a uniform base (conv3d_split.hip, DESIGN 3s: computed per chunk this was ~100 vector instructions at every chunk's top)
  unsigned ep_off[ADD_AHEAD ? R : 1], ep_chan[ADD_AHEAD ? MT : 1][16];
  int ep_b = 0;
  auto ep_tile = [&](int k) {
    int b, h0, w0;
    tile_of(k, b, h0, w0);
    ep_b = __builtin_amdgcn_readfirstlane(b);
#pragma unroll
    for (int r = 0; r < (ADD_AHEAD ? R : 1); ++r) {
      const int gh = h0 + wave * R + r, gw = w0 + (lane & 31);
      ep_off[r] = (gh < d.H && gw < d.W) ? (unsigned)(gh * d.W + gw) : (unsigned)(lane & 31);
    }
  };
  if (ADD_AHEAD) {
    ep_tile(0);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int qq = 0; qq < 16; ++qq)
        ep_chan[m][qq] = (unsigned)min(d.o0 + m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half, d.Co - 1) * (unsigned)HWi;
  }

  // weight fragments: ring of 3 taps, fetched 2 taps ahead
  uint4 aring[3][MT][3];
  auto load_a = [&](int slot3, int ch, int tap) {
    const uint4* wq = wp + ((long long)ch * 9 + tap) * 192 + lane;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int p = 0; p < NP; ++p) aring[slot3][m][p] = wq[m * mstride + p * 64];
  };

  if (G > 0) {
    stage_begin();
#pragma unroll
    for (int k = 0; k < KIT; ++k) stage_load(k);
#pragma unroll
    for (int k = 0; k < KIT; ++k) stage_commit(0, k);
    load_a(0, 0, 0);
    load_a(1, 0, 1);
  }
  lds_barrier();

  int ch = 0, k_tile = 0;
  for (int g = 0; g < G; ++g) {
    const uint4* src = sm + (g & 1) * BUF;
    const int ch_next = ch + 1 < d.NCHUNK ? ch + 1 : 0;
    stage_advance(g + 1 < G ? 1 : 0);  // (after the last chunk: the same one once more into the idle buffer -- keeps the body free of branches)
    stage_begin();
    unsigned ep_cur[ADD_AHEAD ? R : 1];  // where this chunk's residual requests go: the tile's pixels in its last chunk, a cached dummy row else
    const float* ep_base = epi.add;
    if (ADD_AHEAD) {
      const bool last = ch == d.NCHUNK - 1;
      ep_base = epi.add + (long long)(last ? ep_b : 0) * d.Co * HWi;
#pragma unroll
      for (int r = 0; r < R; ++r) ep_cur[r] = last ? ep_off[r] : (unsigned)(lane & 31);
    }
    uint4 bq[2][R][3];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int p = 0; p < NP; ++p) bq[0][r][p] = src[2 * p * PIECE + rowpos[r]];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap + 1 < 9) {
        const int toff = ((tap + 1) / 3) * DIL * IW + ((tap + 1) % 3) * DIL;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int p = 0; p < NP; ++p) bq[(tap + 1) & 1][r][p] = src[2 * p * PIECE + rowpos[r] + toff];
      }
      if (tap + 2 < 9)
        load_a((tap + 2) % 3, ch, tap + 2);
      else
        load_a((tap + 2) % 3, ch_next, tap + 2 - 9);
      if (tap < KIT) stage_load(tap);
      if (ADD_AHEAD && tap >= 9 - 2 * R * MT && tap < 9) {  // 8 residual values under each of the last 2 * R * MT taps
        const int i = tap - (9 - 2 * R * MT), m = i / (2 * R), r = (i / 2) % R;
#pragma unroll
        for (int qq = 8 * (i & 1); qq < 8 * (i & 1) + 8; ++qq) addv[m][r][qq] = ep_base[ep_cur[r] + ep_chan[m][qq]];
      }
      if (tap >= 9 - KIT) stage_commit((g + 1) & 1, tap - (9 - KIT));
      if constexpr (F16) {
#define MODE_SPLIT_TERM(PA, PB)                                                      \
  _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int r = 0; r < R; ++r) \
      acc[m][r] = mfma_f16(aring[tap % 3][m][PA], bq[tap & 1][r][PB], acc[m][r]);
        MODE_SPLIT_TERM(1, 0)
        MODE_SPLIT_TERM(0, 1)
        MODE_SPLIT_TERM(0, 0)
#undef MODE_SPLIT_TERM
      } else {
#define MODE_SPLIT_TERM(PA, PB)                                                      \
  _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int r = 0; r < R; ++r) \
      acc[m][r] = mfma_bf16(aring[tap % 3][m][PA], bq[tap & 1][r][PB], acc[m][r]);
        MODE_SPLIT_TERM(2, 0)
        MODE_SPLIT_TERM(0, 2)
        MODE_SPLIT_TERM(1, 1)
        MODE_SPLIT_TERM(1, 0)
        MODE_SPLIT_TERM(0, 1)
        MODE_SPLIT_TERM(0, 0)
#undef MODE_SPLIT_TERM
      }
      // (F16: half the MFMAs carry two thirds of the staging work.  Per tap: 48 vector instructions of a position's split, R * NP
      // fragment reads + 4 stores, 16 + MT * NP loads -- dealt evenly over the tap's MFMA gaps, every kind with its own slot: a generous
      // allowance per gap let the scheduler fill a tap's first gaps with 6-16 instructions and leave the other two thirds empty)
      if constexpr (F16) {
        constexpr int NM = MT * R * 3;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, (48 + NM - 1) / NM + 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x080, (R * NP + 4 + NM - 1) / NM, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, (16 + MT * NP + NM - 1) / NM, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < MT * R * 6; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, MT * R <= 4 ? 5 : 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ch == d.NCHUNK - 1) {  // tile finished: D[i = o][j = w]
      int b, h0, w0;
      tile_of(k_tile, b, h0, w0);
      float* yb = y + (long long)b * d.Co * HWi;
      const int gw = w0 + (lane & 31);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int gh = h0 + wave * R + r;
        if (gh < d.H && gw < d.W) {
          const int sp = gh * d.W + gw;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            // (where the residual values were not fetched ahead: the 16 of this row and channel block requested together in front of its
            // stores -- read next to each store, `add` may alias y for all the compiler knows, every load waited for the store before it;
            // a block wholly inside the layer's channels stores without a test per channel)
            float res[16], shl[16];
            if (EPI == 2 && !ADD_AHEAD) {
              const float* ap = epi.add + (long long)b * d.Co * HWi + sp;
              int mo = d.o0 + m * 32 + 4 * half;  // opaque: or the shift requests are hoisted out of the tile loop again (and spilled)
              asm volatile("" : "+v"(mo));
#pragma unroll
              for (int qq = 0; qq < 16; ++qq) {
                const int oc = min(mo + (qq & 3) + 8 * (qq >> 2), d.Co - 1);
                shl[qq] = epi.shift[oc];
                res[qq] = ap[(long long)oc * HWi];
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            auto emit = [&](int qq) {
              const int o = d.o0 + m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half;
              float v = F16 ? acc[m][r][qq] * unscale : acc[m][r][qq];
              if (EPI) v += SHIFT_LATE ? shl[qq] : shiftv[m][qq];
              if (EPI == 2) v += ADD_AHEAD ? addv[ADD_AHEAD ? m : 0][ADD_AHEAD ? r : 0][qq] : res[qq];
              const float vo = (EPI && v < relu_floor) ? relu_floor : v;  // (NaN passes, as in torch.relu and the fp32 kernels)
              if (EPI && F16) out_mag = max(out_mag, mode::absmax_mag(vo));
              yb[(long long)o * HWi + sp] = vo;
            };
            if (d.o0 + m * 32 + 32 <= d.Co) {
#pragma unroll
              for (int qq = 0; qq < 16; ++qq) emit(qq);
            } else {
#pragma unroll
              for (int qq = 0; qq < 16; ++qq)
                if (d.o0 + m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half < d.Co) emit(qq);
            }
            if (EPI == 2 && !ADD_AHEAD) __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][r] = (f32x16){0};
      }
      ++k_tile;
      if (ADD_AHEAD) ep_tile(min(k_tile, max(mine - 1, 0)));
    }
    ch = ch_next;
    lds_barrier();
  }
  if (EPI && F16) {
    if (epi.amax) {  // (uniform)
      __syncthreads();
      mode::absmax_block_commit(out_mag, epi.amax, reinterpret_cast<unsigned*>(sm));
    }
  }
}

// The maximum buffer of an eval layer's FOLDED weights w[o][..] * scale[o] (what pack_w2d_split splits when fold == 1), one block: word 0
// the value, the slots zero.  n = elements per output row.  (conv3d_split.hip has the same kernel for its layers.)
__global__ __launch_bounds__(1024) void abs_max_folded2d_kernel(const float* __restrict__ w, mode_bn_epilogue bn, int rows, int n,
                                                                unsigned* __restrict__ out) {
  unsigned m = 0;
  for (int i = threadIdx.x; i < rows * n; i += 1024) m = max(m, mode::absmax_mag(w[i] * fold_scale(bn, i / n)));
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

template <int MT, int TH, int DIL>
int launch2(const float* x, const uint4* wp, float* y, S2Dims d, hipStream_t st, const char* who, Epi epi, const float* amax_x,
            const float* amax_w, int nblocks) {
  constexpr size_t LDS = Geo2<TH, DIL>::LDS_BYTES;
  d.nHt = mode::cdiv(d.H, TH);
  d.ntiles = d.B * d.nHt * d.nWt;
  const int grid = kNumCU;
#define MODE_C2D_LAUNCH(EPIV, F16V)                                                                                                  \
  {                                                                                                                                  \
    int rc = mode::allow_lds(conv2d_split_kernel<MT, TH, DIL, EPIV, F16V>, LDS, who);                                                \
    if (rc != MODE_OK) return rc;                                                                                                    \
    hipLaunchKernelGGL((conv2d_split_kernel<MT, TH, DIL, EPIV, F16V>), dim3(grid, nblocks), dim3(NT), LDS, st, x, wp, y, d, epi, amax_x, amax_w); \
  }
  if (epi.shift && epi.add) {
    if (amax_x) MODE_C2D_LAUNCH(2, true) else MODE_C2D_LAUNCH(2, false)
  } else if (epi.shift) {
    if (amax_x) MODE_C2D_LAUNCH(1, true) else MODE_C2D_LAUNCH(1, false)
  } else {
    if (amax_x) MODE_C2D_LAUNCH(0, true) else MODE_C2D_LAUNCH(0, false)
  }
#undef MODE_C2D_LAUNCH
  return mode::check_launch(who);
}

template <int MT>
int launch_tile(const float* x, const uint4* wp, float* y, const S2Dims& d, int dilation, hipStream_t st, const char* who, Epi epi,
                const float* ax, const float* aw, int nblocks) {
  // 16-row tiles unless that leaves fewer than two tiles per CU (a workgroup's first chunk is staged un-overlapped)
  const bool big = (long long)d.B * mode::cdiv(d.H, 16) * d.nWt * nblocks >= 2 * kNumCU;
  if (dilation == 1)
    return big ? launch2<MT, 16, 1>(x, wp, y, d, st, who, epi, ax, aw, nblocks) : launch2<MT, 8, 1>(x, wp, y, d, st, who, epi, ax, aw, nblocks);
  return big ? launch2<MT, 16, 2>(x, wp, y, d, st, who, epi, ax, aw, nblocks) : launch2<MT, 8, 2>(x, wp, y, d, st, who, epi, ax, aw, nblocks);
}

}  // namespace

namespace mode {

size_t conv2d_split_wpack_floats(int K, int rows) {
  // fragments + the folded BatchNorm shifts + (eval on the fp16 arithmetic) the maximum buffer of the folded weights
  return (size_t)cdiv(rows, 32) * cdiv(K, 16) * 9 * 3 * 64 * 4 + 32 * (size_t)cdiv(rows, 32) + MODE_BN_ABSMAX_FLOATS;
}

bool conv2d_split_supported(int K, int rows, int dilation) {
  // (any number of 64-channel output blocks: one launch each; 512 covers the fusion network's 256-channel bottleneck with room)
  return rows > 1 && rows <= 512 && K % 16 == 0 && (dilation == 1 || dilation == 2);
}

// rows = output channels of THIS GEMM (Co forward, Ci for the input gradient), K = its reduction channels; flip 0 / 1 as pack_w2d_split
int conv2d_split_run(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int H, int W, int dilation, int flip,
                     hipStream_t st, const char* who, const mode_bn_epilogue* bn, const float* acc_in, const float* amax_x,
                     const float* amax_w, float* amax_y) {
  MODE_REQUIRE(B >= 0 && K > 0 && rows > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  // eval mode on the fp16 arithmetic (bn, amax_x, amax_y; no amax_w): the weights' maximum is the FOLDED weights', taken with the pack
  // and kept in the workspace behind the shifts (conv3d_split.hip)
  const bool eval16 = amax_x && bn;
  MODE_REQUIRE(eval16 ? (!amax_w && amax_y && !acc_in && flip == 0) : ((amax_x == nullptr) == (amax_w == nullptr) && !amax_y), MODE_ERR_BAD_ARG,
               "%s: the fp16 arithmetic takes both operand maxima (eval epilogue: the input's and the output's)", who);
  MODE_REQUIRE(!(acc_in && bn), MODE_ERR_BAD_ARG, "%s: the accumulate form takes no BatchNorm epilogue", who);
  MODE_REQUIRE(!acc_in || acc_in != y, MODE_ERR_BAD_ARG, "%s: acc must not be the output tensor", who);
  MODE_REQUIRE(conv2d_split_supported(K, rows, dilation), MODE_ERR_UNSUPPORTED,
               "%s: %d output / %d reduction channels, dilation %d not covered by the split kernel", who, rows, K, dilation);
  MODE_REQUIRE((long long)std::max(K, rows) * H * W < (1ll << 29), MODE_ERR_UNSUPPORTED, "%s: a sample larger than 2^29 elements", who);
  if (B == 0) return amax_y ? mode::absmax_begin(amax_y, st, who) : MODE_OK;
  MODE_REQUIRE(x && w && y && wpack, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  S2Dims d;
  d.B = B; d.K = K; d.Co = rows; d.H = H; d.W = W;
  d.NCHUNK = K / 16;
  d.nWt = cdiv(W, 32);
  const int MT = cdiv(rows, 32);
  const long long npack = (long long)MT * d.NCHUNK * 9 * 64;
  uint4* wp = reinterpret_cast<uint4*>(wpack);
  float* wmax = wpack + npack * 3 * 4 + 32 * MT;  // (eval16)
  if (eval16) amax_w = wmax;
  if (mode::pack_needed()) {
    if (eval16) {
      hipLaunchKernelGGL(abs_max_folded2d_kernel, dim3(1), dim3(1024), 0, st, w, *bn, rows, K * 9, reinterpret_cast<unsigned*>(wmax));
      hipLaunchKernelGGL(pack_w2d_split<true>, dim3(cdiv(npack, 256)), dim3(256), 0, st, w, wp, rows, K, MT, d.NCHUNK, flip, 1, *bn, wmax);
    } else if (amax_x)
      hipLaunchKernelGGL(pack_w2d_split<true>, dim3(cdiv(npack, 256)), dim3(256), 0, st, w, wp, rows, K, MT, d.NCHUNK, flip, acc_in ? 2 : 0,
                         mode_bn_epilogue(), amax_w);
    else
      hipLaunchKernelGGL(pack_w2d_split<false>, dim3(cdiv(npack, 256)), dim3(256), 0, st, w, wp, rows, K, MT, d.NCHUNK, flip,
                         bn ? 1 : acc_in ? 2 : 0, bn ? *bn : mode_bn_epilogue(), (const float*)nullptr);
  }
  Epi epi = make_epi(bn, wpack + npack * 3 * 4);
  if (acc_in) {  // y = conv(x) + acc_in: the residual epilogue with zero shifts ((v + 0) + a is a + v exactly)
    epi.shift = wpack + npack * 3 * 4;
    epi.add = acc_in;
    epi.relu = 0;
  }
  if (amax_y) {
    int rc = mode::absmax_begin(amax_y, st, who);
    if (rc != MODE_OK) return rc;
    epi.amax = reinterpret_cast<unsigned*>(amax_y);
  }
  // two output-channel tiles per workgroup (64 channels); the layer's 64-channel blocks are the y-slices of ONE launch (round 6; a
  // launch per block before), an odd 32-channel block at the end a launch of its own
  d.o0 = 0;
  if (MT >= 2) {
    int rc = launch_tile<2>(x, wp, y, d, dilation, st, who, epi, amax_x, amax_w, MT / 2);
    if (rc != MODE_OK) return rc;
  }
  if (MT % 2) {
    d.o0 = 32 * (MT - 1);
    return launch_tile<1>(x, wp + (long long)(MT - 1) * d.NCHUNK * 9 * 192, y, d, dilation, st, who, epi, amax_x, amax_w, 1);
  }
  return MODE_OK;
}

}  // namespace mode
