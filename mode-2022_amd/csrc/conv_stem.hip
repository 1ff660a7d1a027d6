// The stem of the 2-D feature extractor: Conv2d(3 -> 32, kernel 7, stride 2, padding 3, no bias) on the full-resolution image
// (reference: firstconv[0] of sphere_feature_extraction, models/submodule.py:155; cuDNN there).  gfx950 / fp32 MFMA.
//
// Few input channels and many taps: the contraction index is kk = (c, kh, kw), K = 49 * Ci = 147.
//   forward:          y[o][ho][wo] = sum_kk W[o][kk] * x[c][2 ho + kh - 3][2 wo + kw - 3]           D[i = o][j = 32 consecutive wo]
//   weight gradient:  gW[o][kk]    = sum_{b,ho,wo} gy[o][ho][wo] * x[c][2 ho + kh - 3][2 wo + kw - 3]  D[i = o][j = kk], K = pixels
// Both read the image through a haloed LDS tile [Ci][13 rows][69 columns] (4 output rows x 32 output columns); a lane's operand
// address is (offset of its tap) + (offset of its pixel), the tap offset being a compile-time constant per k-step in the forward
// and a per-lane register per 32-tap block in the weight gradient.  The image has no gradient, so there is no input-gradient
// kernel.  HBM-side the layer is tiny (25 MB in, 67 MB out for 4 images); what this file buys is not having to run it as
// 2 x 49-tap gathers (0.7 ms forward, 0.6 ms weight gradient -> ~0.1 ms each).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 256;
constexpr int KS = 7, ST = 2, PD = 3;
constexpr int TR = 4, TC = 32;                     // output tile: one row per wave, 32 columns
constexpr int IR = (TR - 1) * ST + KS;             // 13 input rows
constexpr int IC = (TC - 1) * ST + KS;             // 69 input columns (odd pitch)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct SDims {
  int B, Ci, Co, H, W, Ho, Wo;
  int nHt, nWt, ntiles;
  int KK;    // Ci * 49
  int NK4;   // ceil(ceil(KK / 2) / 4): float4 groups of k-steps
};

__host__ __device__ constexpr int tap_off(int kk) {  // LDS offset of tap kk = (c, kh, kw) relative to the tile's pixel origin
  return (kk / 49) * (IR * IC) + ((kk % 49) / KS) * IC + (kk % KS);
}

// wp[(t4*64 + lane)][j] = W[o = lane & 31][kk = 2*(4*t4 + j) + (lane >> 5)], zero past Co / KK; fold: see common.h
__global__ void pack_w_stem(const float* __restrict__ w, float* __restrict__ wp, int Co, int KK, int NK4, int fold, mode_bn_epilogue bn) {
  const int total = NK4 * 64 * 4;
  if (fold && blockIdx.x == 0)
    for (int o = threadIdx.x; o < Co; o += blockDim.x) wp[total + o] = fold_shift(bn, o);
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int j = idx & 3, lane = (idx >> 2) & 63, t4 = idx >> 8;
    const int o = lane & 31, kk = 2 * (4 * t4 + j) + (lane >> 5);
    float v = 0.f;
    if (o < Co && kk < KK) v = w[(long long)o * KK + kk];
    if (fold && o < Co) v *= fold_scale(bn, o);
    wp[idx] = v;
  }
}

__device__ __forceinline__ void stage_image_tile(const float* __restrict__ xb, float* __restrict__ tile, const SDims& d, int h0, int w0) {
  const long long HW = (long long)d.H * d.W;
  const int n = d.Ci * IR * IC;
  for (int idx = threadIdx.x; idx < n; idx += NT) {
    const int c = idx / (IR * IC), rem = idx - c * (IR * IC);
    const int r = rem / IC, col = rem - r * IC;
    const int gh = h0 * ST - PD + r, gw = w0 * ST - PD + col;
    const bool ok = gh >= 0 && gh < d.H && gw >= 0 && gw < d.W;
    const float v = xb[ok ? c * HW + (long long)gh * d.W + gw : 0];
    tile[idx] = ok ? v : 0.f;
  }
}

template <int CI, bool EPI>
__global__ __launch_bounds__(NT) void stem_fwd_kernel(const float* __restrict__ x, const float4* __restrict__ wp, float* __restrict__ y,
                                                      SDims d, Epi epi) {
  __shared__ float tile[CI * IR * IC];
  int t = blockIdx.x;
  const int wt = t % d.nWt;
  t /= d.nWt;
  const int ht = t % d.nHt;
  const int b = t / d.nHt;
  const int h0 = ht * TR, w0 = wt * TC;
  stage_image_tile(x + (long long)b * d.Ci * d.H * d.W, tile, d, h0, w0);
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5;
  const float* bp = tile + wave * ST * IC + ST * (lane & 31);
  constexpr int KK = CI * 49, NK = (KK + 1) / 2, NK4 = (NK + 3) / 4;
  f32x16 acc = (f32x16){0};
  float4 a_nxt = wp[lane];
#pragma unroll
  for (int t4 = 0; t4 < NK4; ++t4) {
    const float4 a4 = a_nxt;
    if (t4 + 1 < NK4) a_nxt = wp[(t4 + 1) * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ks = 4 * t4 + j;
      if (ks < NK) {
        const int o0 = tap_off(2 * ks), o1 = tap_off(2 * ks + 1 < KK ? 2 * ks + 1 : 2 * ks);  // (a tap past KK meets a zero weight)
        const float bv = bp[half ? o1 : o0];
        const float av = j == 0 ? a4.x : j == 1 ? a4.y : j == 2 ? a4.z : a4.w;
        acc = mfma32(av, bv, acc);
      }
    }
  }
  const int gh = h0 + wave, gw = w0 + (lane & 31);
  if (gh < d.Ho && gw < d.Wo) {
    const long long oHW = (long long)d.Ho * d.Wo;
    float* yb = y + (long long)b * d.Co * oHW + (long long)gh * d.Wo + gw;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (o < d.Co) yb[o * oHW] = EPI ? apply_epi(epi, acc[r], o, (yb - y) + o * oHW) : acc[r];
    }
  }
}

// Weight gradient: persistent workgroups over the tiles, accumulators in registers; the 4 waves (= the 4 rows of a tile) are summed
// through LDS in wave order at the end; part[workgroup][Co][KK], reduced in fixed order by reduce_slices_kernel (common.h).
template <int CI>
__global__ __launch_bounds__(NT) void stem_bww_kernel(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ part,
                                                      SDims d) {
  constexpr int KK = CI * 49, NTL = (KK + 31) / 32;
  constexpr int GP = TR * TC + 1;  // gy tile pitch per output channel (odd)
  constexpr int XT = CI * IR * IC;
  constexpr int STAGE = XT + 32 * GP, RED = 3 * NTL * 1024;
  __shared__ float sm[STAGE > RED ? STAGE : RED];
  float* xt = sm;
  float* gt = sm + XT;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, row = lane & 31;
  int tb[NTL];
#pragma unroll
  for (int n = 0; n < NTL; ++n) {
    const int kk = n * 32 + row;
    tb[n] = tap_off(kk < KK ? kk : 0);
  }
  f32x16 acc[NTL];
#pragma unroll
  for (int n = 0; n < NTL; ++n) acc[n] = (f32x16){0};
  const long long oHW = (long long)d.Ho * d.Wo;
  for (int tile = blockIdx.x; tile < d.ntiles; tile += gridDim.x) {
    int t = tile;
    const int wt = t % d.nWt;
    t /= d.nWt;
    const int ht = t % d.nHt;
    const int b = t / d.nHt;
    const int h0 = ht * TR, w0 = wt * TC;
    stage_image_tile(x + (long long)b * d.Ci * d.H * d.W, xt, d, h0, w0);
    const float* gb = gy + (long long)b * d.Co * oHW;
    for (int idx = threadIdx.x; idx < 32 * TR * TC; idx += NT) {
      const int o = idx / (TR * TC), rem = idx - o * (TR * TC);
      const int r = rem / TC, col = rem - r * TC;
      const int gh = h0 + r, gw = w0 + col;
      const bool ok = o < d.Co && gh < d.Ho && gw < d.Wo;
      const float v = gb[ok ? o * oHW + (long long)gh * d.Wo + gw : 0];
      gt[o * GP + rem] = ok ? v : 0.f;
    }
    __syncthreads();
    const float* ap = gt + row * GP + wave * TC + half;
    const float* xp = xt + wave * ST * IC + ST * half;
#pragma unroll 4
    for (int ks = 0; ks < TC / 2; ++ks) {
      const float av = ap[2 * ks];
      const int po = 2 * ST * ks;
#pragma unroll
      for (int n = 0; n < NTL; ++n) acc[n] = mfma32(av, xp[tb[n] + po], acc[n]);
    }
    __syncthreads();
  }
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < NTL; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) sm[((wave - 1) * NTL + n) * 1024 + r * 64 + lane] = acc[n][r];
  }
  __syncthreads();
  if (wave == 0) {
    float* pb = part + (long long)blockIdx.x * d.Co * KK;
#pragma unroll
    for (int n = 0; n < NTL; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[n][r];
        v += sm[(0 * NTL + n) * 1024 + r * 64 + lane];
        v += sm[(1 * NTL + n) * 1024 + r * 64 + lane];
        v += sm[(2 * NTL + n) * 1024 + r * 64 + lane];
        const int o = (r & 3) + 8 * (r >> 2) + 4 * half, kk = n * 32 + row;
        if (o < d.Co && kk < KK) pb[o * KK + kk] = v;
      }
  }
}

int make_dims(SDims& d, int B, int Ci, int H, int W, int Co, const char* who) {
  MODE_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(Ci == 3 && Co <= 32, MODE_ERR_UNSUPPORTED, "%s: built for 3 input and <= 32 output channels (got %d -> %d)", who, Ci, Co);
  MODE_REQUIRE((long long)32 * H * W < (1ll << 31), MODE_ERR_UNSUPPORTED, "%s: a sample larger than 2^31 elements", who);
  d.B = B; d.Ci = Ci; d.Co = Co; d.H = H; d.W = W;
  d.Ho = (H + 2 * PD - KS) / ST + 1;
  d.Wo = (W + 2 * PD - KS) / ST + 1;
  d.nHt = mode::cdiv(d.Ho, TR);
  d.nWt = mode::cdiv(d.Wo, TC);
  d.ntiles = B * d.nHt * d.nWt;
  d.KK = Ci * 49;
  d.NK4 = mode::cdiv(mode::cdiv(d.KK, 2), 4);
  return MODE_OK;
}

int stem_groups(const SDims& d) { return std::max(1, std::min(d.ntiles, 2 * kNumCU)); }

int stem_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int H, int W, int Co, hipStream_t st,
             const mode_bn_epilogue* bn, const char* who) {
  SDims d;
  int rc = make_dims(d, B, Ci, H, W, Co, who);
  if (rc != MODE_OK || B == 0) return rc;
  MODE_REQUIRE(x && w && y && wpack, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  const int npack = d.NK4 * 256;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w_stem, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, Co, d.KK, d.NK4, bn ? 1 : 0,
                     bn ? *bn : mode_bn_epilogue());
  const Epi epi = make_epi(bn, wpack + npack);
  const float4* wp4 = reinterpret_cast<const float4*>(wpack);
  if (bn)
    hipLaunchKernelGGL((stem_fwd_kernel<3, true>), dim3(d.ntiles), dim3(NT), 0, st, x, wp4, y, d, epi);
  else
    hipLaunchKernelGGL((stem_fwd_kernel<3, false>), dim3(d.ntiles), dim3(NT), 0, st, x, wp4, y, d, epi);
  return mode::check_launch(who);
}

}  // namespace

extern "C" size_t mode_conv_stem_wpack_bytes(int Ci, int Co) {
  if (Ci <= 0 || Co <= 0) return 0;
  return ((size_t)mode::cdiv(mode::cdiv(Ci * 49, 2), 4) * 256 + 32) * sizeof(float);
}

extern "C" int mode_conv_stem_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int H, int W, int Co,
                                  mode_stream_t stream) {
  return stem_fwd(x, w, y, wpack, B, Ci, H, W, Co, mode::as_stream(stream), nullptr, "mode_conv_stem_fwd");
}

extern "C" int mode_conv_stem_fwd_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci,
                                     int H, int W, int Co, mode_stream_t stream) {
  int rc = mode::check_bn(bn, "mode_conv_stem_fwd_bn");
  if (rc != MODE_OK) return rc;
  return stem_fwd(x, w, y, wpack, B, Ci, H, W, Co, mode::as_stream(stream), bn, "mode_conv_stem_fwd_bn");
}

extern "C" size_t mode_conv_stem_bwd_weight_workspace_bytes(int B, int Ci, int H, int W, int Co) {
  SDims d;
  if (B <= 0 || make_dims(d, B, Ci, H, W, Co, "mode_conv_stem_bwd_weight_workspace_bytes") != MODE_OK) return 0;
  return (size_t)stem_groups(d) * Co * d.KK * sizeof(float);
}

extern "C" int mode_conv_stem_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int H, int W, int Co,
                                         int accumulate, mode_stream_t stream) {
  const char* who = "mode_conv_stem_bwd_weight";
  SDims d;
  int rc = make_dims(d, B, Ci, H, W, Co, who);
  if (rc != MODE_OK) return rc;
  hipStream_t st = mode::as_stream(stream);
  MODE_REQUIRE(gw, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  if (B == 0) {
    if (!accumulate) return mode::zero_floats(gw, (size_t)Co * d.KK, st, "mode_conv_stem_bwd_weight");
    return MODE_OK;
  }
  MODE_REQUIRE(gy && x && workspace, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  const int S = stem_groups(d);
  hipLaunchKernelGGL(stem_bww_kernel<3>, dim3(S), dim3(NT), 0, st, gy, x, workspace, d);
  const int n = Co * d.KK;
  hipLaunchKernelGGL(reduce_slices_kernel, dim3(mode::cdiv(n, 4)), dim3(256), 0, st, workspace, gw, n, S, accumulate);
  return mode::check_launch(who);
}
