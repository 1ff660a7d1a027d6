// ConvTranspose3d k3 s2 p1 op1 (hourglass conv5 / conv6, models/mode_disparity.py:23-25) -- and with it the input gradient of the
// stride-2 convolutions conv1 / conv3 (:17-19), which is the same operator on the same weight layout -- on the bf16 matrix pipe with
// fp32 operands split exactly into three bf16 pieces: the arithmetic and the structure of conv3d_split.hip (read that file first).
//
//   y[o][2 q + p] = sum_c sum_{taps of parity class p} w[c][o][k] * x[c][q + s(k)]        (q: low-resolution voxel, p in {0,1}^3)
// Per axis an even output index takes kernel index 1 from input q, an odd one kernel index 0 from input q + 1 and kernel index 2
// from input q: the 27 taps fall into 8 parity classes of 1, 2, 2, 2, 4, 4, 4, 8 taps, i.e. 14 tap PAIRS (one half-empty) -- so K of one
// v_mfma_f32_32x32x16_bf16 is again 8 input channels x 2 taps and a chunk is 14 pairs x 6 terms, only that a pair now adds into the
// accumulator of ITS class: D_class[i = o][j = 32 consecutive low-resolution columns].  A wave owns two low-resolution rows x one
// 32-channel output tile = 16 accumulators (256 registers: one wave per SIMD, like the stride-1 kernel), 168 MFMAs per chunk, and
// alternates between its two rows so that consecutive MFMAs never share an accumulator.
// The LDS tile is tiny: (TD + 1) x 5 rows x 33 columns of the low-resolution input per 8-channel chunk (330 / 495 positions, at most
// two per thread) against 672 MFMAs per workgroup and chunk -- the transposed convolution reads each input voxel for 27/8 taps of 8
// outputs each, so unlike the stride-2 forward it is the weights, not the staging, that set the tile: two rows per wave halve the
// weight-fragment traffic (42 KB per wave and chunk from L2) to what the L1 delivers beside the MFMAs.
// Output: the two classes pw = 0 / 1 of a (pd, ph) are stored as one float2 per lane (columns 2 j, 2 j + 1).
#include "common.h"

#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 256;
constexpr int TH = 4, IH = TH + 1, IW = 33;
constexpr int NPAIR = 14;
constexpr int KIT = 2;
constexpr int PIECE = KIT * NT;  // 512 >= 3 x 5 x 33 positions
constexpr int BUF = 3 * PIECE;
constexpr size_t LDS_BYTES = 2 * (size_t)BUF * sizeof(uint4);  // 49 152 B

// the 14 tap pairs: parity class (pd * 4 + ph * 2 + pw), kernel taps (kd * 9 + kh * 3 + kw; -1 = empty) and input shifts (sd, sh, sw)
__host__ __device__ constexpr int pair_cls(int p) {
  constexpr int v[NPAIR] = {0, 1, 2, 3, 3, 4, 5, 5, 6, 6, 7, 7, 7, 7};
  return v[p];
}
__host__ __device__ constexpr int pair_tap(int p, int h) {
  constexpr int a[NPAIR] = {13, 12, 10, 9, 15, 4, 3, 21, 1, 19, 0, 6, 18, 24};
  constexpr int b[NPAIR] = {-1, 14, 16, 11, 17, 22, 5, 23, 7, 25, 2, 8, 20, 26};
  return h ? b[p] : a[p];
}
__host__ __device__ constexpr int tap_shift_off(int tap) {  // LDS offset of the input voxel a tap reads, relative to the output's q
  // kernel index 0 along an axis reads q + 1, indices 1 and 2 read q
  return tap < 0 ? 0 : ((tap / 9 == 0 ? 1 : 0) * IH + ((tap / 3) % 3 == 0 ? 1 : 0)) * IW + (tap % 3 == 0 ? 1 : 0);
}

struct DcDims {
  int B, K, Co, D, H, W;  // low-resolution input volume; K = its channels (reduction), Co = output channels
  int nWt, nHt, nDt, NCHUNK, ntiles;
};

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = pack2(a, b);
  const float ra = a - __builtin_bit_cast(float, p1 << 16), rb = b - __builtin_bit_cast(float, p1 & 0xffff0000u);
  p2 = pack2(ra, rb);
  const float sa = ra - __builtin_bit_cast(float, p2 << 16), sb = rb - __builtin_bit_cast(float, p2 & 0xffff0000u);
  p3 = pack2(sa, sb);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// wp[(((m * NCHUNK + ch) * NPAIR + pair) * 3 + piece) * 64 + lane] = 8 bf16: piece of W[c = ch*8 + j][o = m*32 + (lane & 31)]
// [tap = pair_tap(pair, lane >> 5)], j = 0..7, from the (K, Co, 27) weight of the transposed convolution; zero for the empty tap, o >= Co, c >= K
__global__ void pack_w3d_deconv_split(const float* __restrict__ w, uint4* __restrict__ wp, int K, int Co, int MTr, int NCHUNK) {
  const long long total = (long long)MTr * NCHUNK * NPAIR * 64;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    long long r = idx >> 6;
    const int pair = (int)(r % NPAIR);
    r /= NPAIR;
    const int ch = (int)(r % NCHUNK);
    const int m = (int)(r / NCHUNK);
    const int o = m * 32 + (lane & 31);
    int tap = -1;
#pragma unroll
    for (int p = 0; p < NPAIR; ++p)
      if (p == pair) tap = (lane >> 5) ? pair_tap(p, 1) : pair_tap(p, 0);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ch * 8 + j;
      v[j] = (o < Co && c < K && tap >= 0) ? w[((long long)c * Co + o) * 27 + tap] : 0.f;
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

// MT = output-channel tiles per launch (1: <= 32 output channels, tile 2 x 4 rows; 2: 33..64, tile 1 x 4 rows)
template <int MT>
__global__ __launch_bounds__(NT, 1) void deconv3d_split_kernel(const float* __restrict__ x, const uint4* __restrict__ wp,
                                                                float* __restrict__ y, DcDims d) {
  constexpr int TD = 2 / MT, ID = TD + 1;
  constexpr int ITEMS = ID * IH * IW;
  static_assert(ITEMS <= PIECE, "tile does not fit the staging map");
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];  // [2][3][PIECE]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int m = wave % MT, rp = wave / MT;  // output-channel tile and row pair (tile rows 2 rp, 2 rp + 1) of this wave
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
    d0 = (t % d.nDt) * TD;
    b = t / d.nDt;
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
  unsigned okmask = 0;
  const float* st_xc = x;
  int st_base = 0, st_d0 = 0, st_h0 = 0, st_w0 = 0;
  auto stage_begin = [&](int g) {
    int b;
    const int k_tile = g / d.NCHUNK, ch = g - k_tile * d.NCHUNK;
    tile_of(k_tile, b, st_d0, st_h0, st_w0);
    st_xc = x + ((long long)b * d.K + ch * 8) * DHW;
    st_base = st_d0 * (int)HW + st_h0 * d.W + st_w0;
    okmask = 0;
  };
  auto stage_load = [&](int k) {
    const unsigned ok = (unsigned)(st_d0 + pdz[k] < d.D) & (unsigned)(st_h0 + phy[k] < d.H) & (unsigned)(st_w0 + pwx[k] < d.W);
    okmask |= ok << k;
    const unsigned off = ok ? (unsigned)(st_base + poff[k]) : 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float* xcc = st_xc + (long long)c * DHW;
      raw[k][c] = xcc[off];
    }
  };
  uint32_t sq[3][4];
  auto stage_commit = [&](int buf, int k, int h) {
    const bool ok = (okmask >> k) & 1;
#pragma unroll
    for (int j = 2 * h; j < 2 * h + 2; ++j) split2(ok ? raw[k][2 * j] : 0.f, ok ? raw[k][2 * j + 1] : 0.f, sq[0][j], sq[1][j], sq[2][j]);
    if (h == 1) {
      uint4* dst = sm + buf * BUF + tid + k * NT;
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[p * PIECE] = make_uint4(sq[p][0], sq[p][1], sq[p][2], sq[p][3]);
    }
  };

  f32x16 acc[2][8];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[r][c] = (f32x16){0};
  int rowpos[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int row = 2 * rp + r;
    rowpos[r] = ((row / TH) * IH + row % TH) * IW + (lane & 31);
  }
  const long long mstride = (long long)d.NCHUNK * NPAIR * 192;
  const uint4* wpm = wp + m * mstride;
  uint4 aring[7][3];
  auto load_a = [&](int slot7, int ch, int pair) {
    const uint4* wq = wpm + ((long long)ch * NPAIR + pair) * 192 + lane;
#pragma unroll
    for (int p = 0; p < 3; ++p) aring[slot7][p] = wq[p * 64];
  };

  if (G > 0) {
    stage_begin(0);
#pragma unroll
    for (int k = 0; k < KIT; ++k) stage_load(k);
#pragma unroll
    for (int k = 0; k < KIT; ++k) {
      stage_commit(0, k, 0);
      stage_commit(0, k, 1);
    }
#pragma unroll
    for (int pair = 0; pair < 3; ++pair) load_a(pair, 0, pair);
  }
  __syncthreads();

  int ch = 0, k_tile = 0;
  for (int g = 0; g < G; ++g) {
    const uint4* src = sm + (g & 1) * BUF;
    const int ch_next = ch + 1 < d.NCHUNK ? ch + 1 : 0;
    stage_begin(min(g + 1, G - 1));  // (after the last chunk it is staged once more into the idle buffer: no branch in the body)
    uint4 bq[2][2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[0][r][p] = src[p * PIECE + rowpos[r] + (half ? 0 : 0)];
    // One tap pair (P literal, so that every register array index is a compile-time constant): fragment reads of the next pair, the
    // weight fragments six pairs ahead, the staging work of this pair, then 12 MFMAs -- smallest terms first, the two rows alternating.
#define MODE_DC_PAIR(P, CLS, NOFFA, NOFFB)                                                                                     \
  {                                                                                                                            \
    if (P + 1 < NPAIR) {                                                                                                       \
      const int toff = half ? NOFFB : NOFFA;                                                                                   \
      _Pragma("unroll") for (int r = 0; r < 2; ++r) _Pragma("unroll") for (int p = 0; p < 3; ++p)                              \
          bq[(P + 1) & 1][r][p] = src[p * PIECE + rowpos[r] + toff];                                                           \
    }                                                                                                                          \
    if (P + 3 < NPAIR)                                                                                                         \
      load_a((P + 3) % 7, ch, P + 3);                                                                                          \
    else                                                                                                                       \
      load_a((P + 3) % 7, ch_next, P + 3 - NPAIR);                                                                             \
    if (P < KIT) stage_load(P);                                                                                                \
    if (P >= NPAIR - KIT) {                                                                                                    \
      stage_commit((g + 1) & 1, P - (NPAIR - KIT), 0);                                                                         \
      stage_commit((g + 1) & 1, P - (NPAIR - KIT), 1);                                                                         \
    }                                                                                                                          \
    MODE_DC_TERM(P, CLS, 2, 0) MODE_DC_TERM(P, CLS, 0, 2) MODE_DC_TERM(P, CLS, 1, 1) MODE_DC_TERM(P, CLS, 1, 0)                \
    MODE_DC_TERM(P, CLS, 0, 1) MODE_DC_TERM(P, CLS, 0, 0)                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                                                        \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                       \
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                                                                       \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                       \
    }                                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
  }
#define MODE_DC_TERM(P, CLS, PA, PB)                                                   \
  acc[0][CLS] = mfma_bf16(aring[(P) % 7][PA], bq[(P) & 1][0][PB], acc[0][CLS]);        \
  acc[1][CLS] = mfma_bf16(aring[(P) % 7][PA], bq[(P) & 1][1][PB], acc[1][CLS]);
    MODE_DC_PAIR(0, 0, 1, 0)
    MODE_DC_PAIR(1, 1, 33, 0)
    MODE_DC_PAIR(2, 2, 34, 33)
    MODE_DC_PAIR(3, 3, 1, 0)
    MODE_DC_PAIR(4, 3, 165, 0)
    MODE_DC_PAIR(5, 4, 166, 165)
    MODE_DC_PAIR(6, 5, 1, 0)
    MODE_DC_PAIR(7, 5, 198, 165)
    MODE_DC_PAIR(8, 6, 33, 0)
    MODE_DC_PAIR(9, 6, 199, 198)
    MODE_DC_PAIR(10, 7, 166, 165)
    MODE_DC_PAIR(11, 7, 34, 33)
    MODE_DC_PAIR(12, 7, 1, 0)
    MODE_DC_PAIR(13, 7, 0, 0)
#undef MODE_DC_PAIR
#undef MODE_DC_TERM
    if (ch == d.NCHUNK - 1) {  // tile finished: D_class[i = o][j = low-resolution column]
      int b, d0, h0, w0;
      tile_of(k_tile, b, d0, h0, w0);
      const int gw = w0 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int row = 2 * rp + r;
        const int gd = d0 + row / TH, gh = h0 + row % TH;
        if (gd < d.D && gh < d.H && gw < d.W) {
          // one running pointer per (pd, ph), advanced by whole channel planes and made opaque after every step: left to itself the
          // compiler computes all 128 store addresses of the two rows up front (256 registers, next to 256 accumulators)
          float* yb = y + ((long long)b * d.Co + m * 32 + 4 * half) * oDHW + (long long)(2 * gd) * oHW + (long long)(2 * gh) * oW + 2 * gw;
#pragma unroll
          for (int pdh = 0; pdh < 4; ++pdh) {  // (pd, ph): the two pw classes go out as one float2
            float* yc = yb + (pdh >> 1) * oHW + (pdh & 1) * oW;
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {
              const int o = m * 32 + (qq & 3) + 8 * (qq >> 2) + 4 * half;
              asm volatile("" : "+v"(yc));
              if (o < d.Co) *reinterpret_cast<float2*>(yc) = make_float2(acc[r][2 * pdh][qq], acc[r][2 * pdh + 1][qq]);
              yc += ((qq & 3) == 3 ? 5 : 1) * oDHW;
            }
          }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[r][c] = (f32x16){0};
      }
      ++k_tile;
    }
    ch = ch_next;
    lds_barrier();
  }
}

}  // namespace

namespace mode {

size_t deconv3d_split_wpack_floats(int K, int Co) { return (size_t)cdiv(Co, 32) * cdiv(K, 8) * NPAIR * 3 * 64 * 4; }

bool deconv3d_split_supported(int K, int Co) { return Co > 1 && Co <= 64 && K > 0 && K % 8 == 0; }

// x (B, K, D, H, W), w (K, Co, 27) -> y (B, Co, 2D, 2H, 2W)
int deconv3d_split(const float* x, const float* w, float* y, float* wpack, int B, int K, int Co, int D, int H, int W, hipStream_t st,
                   const char* who) {
  MODE_REQUIRE(deconv3d_split_supported(K, Co), MODE_ERR_UNSUPPORTED, "%s: %d output / %d input channels not supported by the split kernel", who, Co, K);
  MODE_REQUIRE((long long)D * H * W * 8 * std::max(Co, 8) < (1ll << 31) && (long long)D * H * W < (1ll << 27), MODE_ERR_UNSUPPORTED,
               "%s: volume beyond the 32-bit offsets of the split kernel", who);
  DcDims d;
  d.B = B; d.K = K; d.Co = Co; d.D = D; d.H = H; d.W = W;
  const int MTr = cdiv(Co, 32), TD = 2 / MTr;
  d.nWt = cdiv(W, 32); d.nHt = cdiv(H, TH); d.nDt = cdiv(D, TD);
  d.NCHUNK = cdiv(K, 8);
  d.ntiles = B * d.nDt * d.nHt * d.nWt;
  const long long npack = (long long)MTr * d.NCHUNK * NPAIR * 64;
  hipLaunchKernelGGL(pack_w3d_deconv_split, dim3(cdiv(npack, 256)), dim3(256), 0, st, w, reinterpret_cast<uint4*>(wpack), K, Co, MTr, d.NCHUNK);
  int rc;
  if (MTr == 2) {
    rc = mode::allow_lds(deconv3d_split_kernel<2>, LDS_BYTES, who);
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL(deconv3d_split_kernel<2>, dim3(kNumCU), dim3(NT), LDS_BYTES, st, x, reinterpret_cast<const uint4*>(wpack), y, d);
  } else {
    rc = mode::allow_lds(deconv3d_split_kernel<1>, LDS_BYTES, who);
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL(deconv3d_split_kernel<1>, dim3(kNumCU), dim3(NT), LDS_BYTES, st, x, reinterpret_cast<const uint4*>(wpack), y, d);
  }
  return mode::check_launch(who);
}

}  // namespace mode
