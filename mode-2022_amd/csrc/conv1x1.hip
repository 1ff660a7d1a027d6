// 1x1 convolutions (stride 1 and 2, no bias) of the 2-D feature extractor, gfx950 / fp32 MFMA.
//
// Reference: the nn.Conv2d(kernel_size=1) layers of models/submodule.py -- the `downsample` branches of the first block of
// layer1 / layer2 / layer4 (:167-174: 32->64, 64->64 stride 2, 64->128) and lastconv[0] / lastconv[4] (:162: 256->128, 128->32);
// cuDNN GEMMs there.
//
// A 1x1 convolution over NCHW planes is a plain GEMM whose N dimension (pixels) is contiguous in memory:
//     y[b, o, q] = sum_c W[o, c] * x[b, c, s*q]                      D[i = o][j = 32 consecutive pixels]  (v_mfma_f32_32x32x2_f32)
//   forward / input gradient:  B[k = c][j = pixel] is ONE coalesced 128-byte load per half-wave straight from HBM -- no LDS, no
//     barrier; A[i = o][k = c] comes pre-packed in fragment order from L2.  A wave owns 32 pixels and ALL output channels
//     (<= 128 per launch row), so x is read exactly once.  The input gradient is the same kernel on W^T (stride 2: scattered
//     into a zero-filled gx).
//   weight gradient:  gW[o, c] = sum_{b, q} gy[b, o, q] * x[b, c, s*q]   D[i = o][j = c], K = pixels.  Both operands are
//     K-contiguous, the layout MFMA does not like: 64-row x 64-pixel tiles of gy and x are staged in LDS with coalesced 16-byte row
//     loads (double-buffered, the next tile in flight under the MFMAs) and read back transposed.  (A first version let every lane
//     load 16 bytes of ITS row straight from global memory -- 32 cache lines per instruction: 1 TB/s.)  4 waves = 4 K-slices
//     summed through LDS in wave order; split-K partials reduced in fixed order (deterministic).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 256;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct P1 {
  int B, K, rows;   // K = reduction channels, rows = output channels of THIS GEMM
  int H, W, Ho, Wo; // planes of the K-side tensor (H, W) and of the rows-side tensor (Ho, Wo); forward: K side = x
  int s;            // forward: y[q] reads x[s*q]
  int NK4;          // ceil(K / 8)
  int MT;           // ceil(rows / 32)
};

// wp[((mt*NK4 + k4)*64 + lane)][j] = Wsrc(o = mt*32 + (lane&31), c = 8*k4 + 2*j + (lane>>5));  flip == 0: w[o][c] (w is (rows, K));
// flip == 1: w[c][o] (w is (K, rows): the input gradient).  fold: row o scaled by the folded BatchNorm scale, shifts at wp[total + o].
__global__ void pack_w1(const float* __restrict__ w, float* __restrict__ wp, int rows, int K, int MT, int NK4, int flip, int fold,
                        mode_bn_epilogue bn) {
  const long long total = (long long)MT * NK4 * 64 * 4;
  if (fold && blockIdx.x == 0)
    for (int o = threadIdx.x; o < rows; o += blockDim.x) wp[total + o] = fold_shift(bn, o);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    const long long r = idx >> 8;
    const int k4 = (int)(r % NK4), mt = (int)(r / NK4);
    const int o = mt * 32 + (lane & 31), c = 8 * k4 + 2 * j + (lane >> 5);
    float v = 0.f;
    if (o < rows && c < K) v = flip ? w[(long long)c * rows + o] : w[(long long)o * K + c];
    if (fold && o < rows) v *= fold_scale(bn, o);
    wp[idx] = v;
  }
}

// One wave = one segment of 32 output pixels (of the Ho x Wo plane) x MTB*32 output channels.  grid = (ceil(B*tiles/4), ceil(MT/MTB)).
// SCATTER (input gradient of a stride-2 layer): the 32 "pixels" are low-resolution positions of gy, the result goes to position
// (s*ho, s*wo) of the zero-filled high-resolution gx.
template <int MTB, bool EPI, bool SCATTER>
__global__ __launch_bounds__(NT) void conv1x1_kernel(const float* __restrict__ x, const float4* __restrict__ wp, float* __restrict__ y,
                                                     P1 d, Epi epi) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int No = d.Ho * d.Wo;
  const int tpi = (No + 31) / 32;
  const long long tile = (long long)blockIdx.x * 4 + wave;
  if (tile >= (long long)d.B * tpi) return;
  const int b = (int)(tile / tpi);
  const int q = (int)(tile % tpi) * 32 + (lane & 31);
  const bool ok = q < No;
  const int ho = q / d.Wo, wo = q - ho * d.Wo;
  // forward: the K-side plane is (H, W) and is read at (s*ho, s*wo); SCATTER: the K-side plane is (Ho, Wo) itself
  const long long kplane = SCATTER ? (long long)No : (long long)d.H * d.W;
  const long long koff = !ok ? 0 : (SCATTER ? (long long)q : (long long)ho * d.s * d.W + (long long)wo * d.s);
  const float* xb = x + (long long)b * d.K * kplane + koff;
  const int half = lane >> 5;
  const int mt0 = blockIdx.y * MTB;

  f32x16 acc[MTB];
#pragma unroll
  for (int m = 0; m < MTB; ++m) acc[m] = (f32x16){0};
  const float4* wq = wp + ((long long)mt0 * d.NK4) * 64 + lane;
  const int kmax = d.K - 1;

  float bv[4];
  float4 av[MTB];
  auto load = [&](int k4, float (&b4)[4], float4 (&a4)[MTB]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = min(8 * k4 + 2 * j + half, kmax);  // (channels past K meet zero weights)
      b4[j] = xb[(long long)c * kplane];
    }
#pragma unroll
    for (int m = 0; m < MTB; ++m) a4[m] = (mt0 + m < d.MT) ? wq[((long long)m * d.NK4 + k4) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  load(0, bv, av);
  for (int k4 = 0; k4 < d.NK4; ++k4) {
    float bn_[4];
    float4 an_[MTB];
    if (k4 + 1 < d.NK4) load(k4 + 1, bn_, an_);  // next 8 channels in flight under the MFMAs below
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < MTB; ++m) {
        const float a = j == 0 ? av[m].x : j == 1 ? av[m].y : j == 2 ? av[m].z : av[m].w;
        acc[m] = mfma32(a, bv[j], acc[m]);
      }
    if (k4 + 1 < d.NK4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = bn_[j];
#pragma unroll
      for (int m = 0; m < MTB; ++m) av[m] = an_[m];
    }
  }

  if (!ok) return;
  const long long oplane = SCATTER ? (long long)d.H * d.W : (long long)No;
  const long long ooff = SCATTER ? (long long)ho * d.s * d.W + (long long)wo * d.s : (long long)q;
  float* yb = y + (long long)b * d.rows * oplane + ooff;
#pragma unroll
  for (int m = 0; m < MTB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = (mt0 + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (o < d.rows) {
        const long long idx = (long long)o * oplane;
        yb[idx] = EPI ? apply_epi(epi, acc[m][r], o, (yb - y) + idx) : acc[m][r];
      }
    }
}

template <bool EPI, bool SCATTER>
int launch1(const float* x, const float* wpack, float* y, const P1& d, hipStream_t st, const Epi& epi, const char* who) {
  const long long tiles = (long long)d.B * mode::cdiv(d.Ho * d.Wo, 32);
  const int gx = mode::cdiv(tiles, 4);
  const float4* wp4 = reinterpret_cast<const float4*>(wpack);
  if (d.MT == 1)
    hipLaunchKernelGGL((conv1x1_kernel<1, EPI, SCATTER>), dim3(gx, 1), dim3(NT), 0, st, x, wp4, y, d, epi);
  else if (d.MT == 2)
    hipLaunchKernelGGL((conv1x1_kernel<2, EPI, SCATTER>), dim3(gx, 1), dim3(NT), 0, st, x, wp4, y, d, epi);
  else
    hipLaunchKernelGGL((conv1x1_kernel<4, EPI, SCATTER>), dim3(gx, mode::cdiv(d.MT, 4)), dim3(NT), 0, st, x, wp4, y, d, epi);
  return mode::check_launch(who);
}

int check1(const void* a, const void* b, const void* c, const void* wp, int B, int Ci, int H, int W, int Co, int stride, const char* who) {
  MODE_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(stride == 1 || stride == 2, MODE_ERR_UNSUPPORTED, "%s: stride %d not implemented (1 or 2)", who, stride);
  MODE_REQUIRE((long long)std::max(Ci, Co) * H * W < (1ll << 31), MODE_ERR_UNSUPPORTED, "%s: a sample larger than 2^31 elements", who);
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(a && b && c && wp, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  return MODE_OK;
}

int fwd1(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int H, int W, int Co, int stride, hipStream_t st,
         const mode_bn_epilogue* bn, const char* who) {
  P1 d;
  d.B = B; d.K = Ci; d.rows = Co; d.H = H; d.W = W; d.s = stride;
  d.Ho = (H - 1) / stride + 1; d.Wo = (W - 1) / stride + 1;
  d.NK4 = mode::cdiv(Ci, 8); d.MT = mode::cdiv(Co, 32);
  const long long npack = (long long)d.MT * d.NK4 * 256;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w1, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, Co, Ci, d.MT, d.NK4, 0, bn ? 1 : 0,
                     bn ? *bn : mode_bn_epilogue());
  const Epi epi = make_epi(bn, wpack + npack);
  return bn ? launch1<true, false>(x, wpack, y, d, st, epi, who) : launch1<false, false>(x, wpack, y, d, st, epi, who);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient.  grid = (S, ceil(MTo/2), ceil(MTc/2)); a workgroup owns a 64 x 64 (o, c) block and the K-slice s; its 4 waves
// take every 4th group of 4 pixels of the slice.  part[s][Co][Ci].
struct W1 {
  int B, Ci, Co, H, W, Ho, Wo, s;
  int groups;  // groups of 4 output pixels per image
  int S;
};

// Both operands are staged through LDS with coalesced 16-byte row loads (a 64-row x 64-pixel tile of gy and of x per step, double
// buffered: the next tile travels global -> registers under the MFMAs) and read back transposed (lane = row, odd row pitch).
constexpr int BW_PX = 64;            // pixels per tile
constexpr int BW_PITCH = BW_PX + 1;  // odd: a fragment read (32 rows, one pixel) hits 32 banks
constexpr int BW_TILE = 128 * BW_PITCH;  // 64 gy rows + 64 x rows
constexpr int BW_LDS_FLOATS = 2 * BW_TILE > 3 * 4 * 1024 ? 2 * BW_TILE : 3 * 4 * 1024;

template <int S2>
__global__ __launch_bounds__(NT) void conv1x1_bww_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ part,
                                                         W1 d) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // BW_LDS_FLOATS: the two tiles; at the end: waves 1..3 -> wave 0
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = lane & 31, half = lane >> 5;
  const int o0 = blockIdx.y * 64, c0 = blockIdx.z * 64;
  const long long No = (long long)d.Ho * d.Wo, HW = (long long)d.H * d.W;
  const int tiles_img = (int)((No + BW_PX - 1) / BW_PX);
  const long long total = (long long)d.B * tiles_img;
  const long long per = (total + d.S - 1) / d.S;
  const long long t_lo = (long long)blockIdx.x * per, t_hi = min(total, t_lo + per);

  // staging map: thread -> (row r = tid / 16 + 16 u, pixel group of 4 = tid % 16), u = 0..7: rows 0..63 gy, 64..127 x
  const int sr = threadIdx.x >> 4, sp = (threadIdx.x & 15) * 4;
  float4 v[8];
  auto issue = [&](long long t) {
    const int b = (int)(t / tiles_img);
    const long long q = (t % tiles_img) * BW_PX + sp;  // first of this thread's 4 output pixels (No % 4 == 0: all in or all out)
    const bool ok = q < No;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = sr + 16 * u;
      if (r < 64) {
        const int o = o0 + r;
        v[u] = *reinterpret_cast<const float4*>(gy + ((long long)b * d.Co + (o < d.Co ? o : 0)) * No + (ok ? q : 0));
        if (!ok || o >= d.Co) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int c = c0 + r - 64;
        const float* xr = x + ((long long)b * d.Ci + (c < d.Ci ? c : 0)) * HW;
        if (S2 == 1) {
          v[u] = *reinterpret_cast<const float4*>(xr + (ok ? q : 0));
        } else {  // x at (2*ho, 2*wo .. 2*wo + 6): the even elements of 8 consecutive floats
          const int ho = (int)((ok ? q : 0) / d.Wo), wo = (int)((ok ? q : 0) - (long long)ho * d.Wo);
          const float* p = xr + (long long)ho * 2 * d.W + 2 * wo;
          const float4 u0 = *reinterpret_cast<const float4*>(p), u1 = *reinterpret_cast<const float4*>(p + 4);
          v[u] = make_float4(u0.x, u0.z, u1.x, u1.z);
        }
        if (!ok || c >= d.Ci) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float* p = buf + (sr + 16 * u) * BW_PITCH + sp;
      p[0] = v[u].x;
      p[1] = v[u].y;
      p[2] = v[u].z;
      p[3] = v[u].w;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i >> 1][i & 1] = (f32x16){0};
  if (t_lo < t_hi) {
    issue(t_lo);
    commit(sm);
  }
  __syncthreads();
  int buf = 0;
  for (long long t = t_lo; t < t_hi; ++t) {
    const bool more = t + 1 < t_hi;
    if (more) issue(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    // wave w takes pixels 16 w .. 16 w + 15 of the tile: 8 k-steps of 2 pixels (lanes 0..31: even pixel, 32..63: odd pixel)
    const float* gt = sm + buf * BW_TILE + row * BW_PITCH + wave * 16 + half;
    const float* xt = gt + 64 * BW_PITCH;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float a0 = gt[2 * ks], a1 = gt[32 * BW_PITCH + 2 * ks];
      const float b0 = xt[2 * ks], b1 = xt[32 * BW_PITCH + 2 * ks];
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
    if (more) commit(sm + (buf ^ 1) * BW_TILE);
    __syncthreads();
    buf ^= 1;
  }
  float (*red)[4][1024] = reinterpret_cast<float (*)[4][1024]>(sm);
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][t][r * 64 + lane] = acc[t >> 1][t & 1][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* pb = part + (long long)blockIdx.x * d.Co * d.Ci;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v2 = acc[t >> 1][t & 1][r];
        v2 += red[0][t][r * 64 + lane];
        v2 += red[1][t][r * 64 + lane];
        v2 += red[2][t][r * 64 + lane];
        const int o = o0 + (t >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;  // D[i = o][j = c = lane & 31]
        const int c = c0 + (t & 1) * 32 + row;
        if (o < d.Co && c < d.Ci) pb[(long long)o * d.Ci + c] = v2;
      }
  }
}

int bww_splits(int B, int groups, int Co, int Ci) {  // groups = groups of 4 output pixels per image
  const long long blocks = (long long)mode::cdiv(Co, 64) * mode::cdiv(Ci, 64);
  long long S = (4LL * kNumCU + blocks - 1) / blocks;
  const long long total = (long long)B * mode::cdiv(groups, 16);  // tiles of 64 pixels
  if (S > (total + 3) / 4) S = (total + 3) / 4;  // at least 4 tiles per slice (the double buffer pays from the second on)
  return (int)std::max(1LL, S);
}

}  // namespace

extern "C" size_t mode_conv1x1_wpack_bytes(int Ci, int Co) {
  if (Ci <= 0 || Co <= 0) return 0;
  const size_t f = (size_t)mode::cdiv(Co, 32) * mode::cdiv(Ci, 8) * 256, b = (size_t)mode::cdiv(Ci, 32) * mode::cdiv(Co, 8) * 256;
  return ((f > b ? f : b) + (size_t)(Co > Ci ? Co : Ci)) * sizeof(float);
}

extern "C" int mode_conv1x1_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int H, int W, int Co, int stride,
                                mode_stream_t stream) {
  const char* who = "mode_conv1x1_fwd";
  int rc = check1(x, w, y, wpack, B, Ci, H, W, Co, stride, who);
  if (rc != MODE_OK || B == 0) return rc;
  return fwd1(x, w, y, wpack, B, Ci, H, W, Co, stride, mode::as_stream(stream), nullptr, who);
}

extern "C" int mode_conv1x1_fwd_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci, int H,
                                   int W, int Co, int stride, mode_stream_t stream) {
  const char* who = "mode_conv1x1_fwd_bn";
  int rc = check1(x, w, y, wpack, B, Ci, H, W, Co, stride, who);
  if (rc == MODE_OK) rc = mode::check_bn(bn, who);
  if (rc != MODE_OK || B == 0) return rc;
  return fwd1(x, w, y, wpack, B, Ci, H, W, Co, stride, mode::as_stream(stream), bn, who);
}

// gx (B, Ci, H, W) is WRITTEN: gx[c, s*q] = sum_o w[o, c] * gy[o, q], zero at the positions a stride-2 layer never read.
extern "C" int mode_conv1x1_bwd_data(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int H, int W, int Co,
                                     int stride, mode_stream_t stream) {
  const char* who = "mode_conv1x1_bwd_data";
  int rc = check1(gy, w, gx, wpack, B, Ci, H, W, Co, stride, who);
  if (rc != MODE_OK || B == 0) return rc;
  hipStream_t st = mode::as_stream(stream);
  P1 d;
  d.B = B; d.K = Co; d.rows = Ci; d.H = H; d.W = W; d.s = stride;
  d.Ho = (H - 1) / stride + 1; d.Wo = (W - 1) / stride + 1;
  d.NK4 = mode::cdiv(Co, 8); d.MT = mode::cdiv(Ci, 32);
  const long long npack = (long long)d.MT * d.NK4 * 256;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w1, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, Ci, Co, d.MT, d.NK4, 1, 0, mode_bn_epilogue());
  const Epi none = make_epi(nullptr, nullptr);
  if (stride == 1) {  // the forward kernel on W^T: the "input" is gy (planes Ho x Wo = H x W)
    return launch1<false, false>(gy, wpack, gx, d, st, none, who);
  }
  rc = mode::zero_floats(gx, (size_t)B * Ci * H * W, st, who);  // (a kernel, not hipMemsetAsync: common.h)
  if (rc != MODE_OK) return rc;
  return launch1<false, true>(gy, wpack, gx, d, st, none, who);
}

extern "C" size_t mode_conv1x1_bwd_weight_workspace_bytes(int B, int Ci, int H, int W, int Co, int stride) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0 || stride < 1) return 0;
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  return (size_t)bww_splits(B, Ho * Wo / 4, Co, Ci) * Co * Ci * sizeof(float);
}

// gw (Co, Ci) written (accumulate = 0) or added to (accumulate = 1).  Needs Wo % 4 == 0 (stride 2: also W % 8 == 0) -- the
// 16-byte row loads; other shapes: MODE_ERR_UNSUPPORTED (the caller uses the gather kernels).
extern "C" int mode_conv1x1_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int H, int W, int Co,
                                       int stride, int accumulate, mode_stream_t stream) {
  const char* who = "mode_conv1x1_bwd_weight";
  int rc = check1(gy, x, gw, workspace, B, Ci, H, W, Co, stride, who);
  if (rc != MODE_OK) return rc;
  W1 d;
  d.B = B; d.Ci = Ci; d.Co = Co; d.H = H; d.W = W; d.s = stride;
  d.Ho = (H - 1) / stride + 1; d.Wo = (W - 1) / stride + 1;
  MODE_REQUIRE(d.Wo % 4 == 0 && (stride == 1 || W % 8 == 0), MODE_ERR_UNSUPPORTED,  // (any H: the last row read is 2 (Ho - 1) <= H - 1)
               "%s: needs an output width divisible by 4 (stride 2: input width divisible by 8), got %dx%d stride %d", who, H, W, stride);
  MODE_REQUIRE(((size_t)gy % 16) == 0 && ((size_t)x % 16) == 0, MODE_ERR_UNSUPPORTED, "%s: tensors must be 16-byte aligned", who);
  hipStream_t st = mode::as_stream(stream);
  if (B == 0) {
    if (!accumulate) return mode::zero_floats(gw, (size_t)Co * Ci, st, who);
    return MODE_OK;
  }
  d.groups = d.Ho * d.Wo / 4;
  d.S = bww_splits(B, d.groups, Co, Ci);
  const dim3 grid(d.S, mode::cdiv(Co, 64), mode::cdiv(Ci, 64));
  const size_t lds = (size_t)BW_LDS_FLOATS * sizeof(float);
  rc = stride == 1 ? mode::allow_lds(conv1x1_bww_kernel<1>, lds, who) : mode::allow_lds(conv1x1_bww_kernel<2>, lds, who);
  if (rc != MODE_OK) return rc;
  if (stride == 1)
    hipLaunchKernelGGL(conv1x1_bww_kernel<1>, grid, dim3(NT), lds, st, gy, x, workspace, d);
  else
    hipLaunchKernelGGL(conv1x1_bww_kernel<2>, grid, dim3(NT), lds, st, gy, x, workspace, d);
  const int n = Co * Ci;
  hipLaunchKernelGGL(reduce_slices_kernel, dim3(mode::cdiv(n, 4)), dim3(256), 0, st, workspace, gw, n, d.S, accumulate);
  return mode::check_launch(who);
}
