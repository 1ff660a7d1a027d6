// First layer of the 3-D regulariser WITHOUT the cost volume (reference: models/mode_disparity.py:104-116 -- the concatenation
// volume followed by dres0[0] = Conv3d(64 -> 32, k3 s1 p1)).
//
// cost[c][d'][h][w'] is ref[c][h][w'] (c < C) or tgt[c-C][h][w'-d'] for w' >= d', else 0: the reference half does not depend on
// d' at all and the target half only on w'-d'.  So with the per-(kd,kw) partial convolutions over (channel, kh)
//     R_t[o][h][w'] = sum_{c,kh} W[o][c  ][kd][kh][kw] * ref[c][h+kh-1][w']          t = kd*3 + kw
//     T_t[o][h][u ] = sum_{c,kh} W[o][C+c][kd][kh][kw] * tgt[c][h+kh-1][u ]
// (18 small 2-D products, a GEMM of K = 3C over the feature maps -- 7 GFLOP instead of the layer's 261) the layer is
//     out[o][d][h][w] = sum_t [0 <= d' < D][d' <= w' < W] ( R_t[o][h][w'] + T_t[o][h][w'-d'] ),   d' = d+kd-1, w' = w+kw-1,
// exactly (the masks are the volume's zero triangle and the convolution's zero padding in d and w; the padding in h is inside
// R_t / T_t).  These two kernels do that assembly and its adjoint; the 402.7 MB volume per sample never exists.
// HBM-bound: the forward writes B*Co*D*H*W floats once (9 LDS reads per element), the adjoint reads them once.
#include "common.h"

#include "bn_internal.h"

namespace {

constexpr int NT = 128;
constexpr int NTB = 256;  // the backward assembly: two halves of 128 threads share the columns

// grid = B*Co*H blocks; LDS = 9*(W+2) + 9*W floats.  R, T: (B, 9*Co, H, W) with channel = t*Co + o.
// EPI: eval-mode BatchNorm (+ ReLU) of dres0[0] applied to the assembled value on its way out (scale and shift of channel o from
// the four BatchNorm vectors, common.h); `add` is not supported here (dres0[0] has no residual).
template <bool EPI>
__global__ __launch_bounds__(NT) void cost_conv_assemble_fwd_kernel(const float* __restrict__ R, const float* __restrict__ T,
                                                                    float* __restrict__ out, int B, int Co, int D, int H, int W,
                                                                    mode_bn_epilogue bn, unsigned* __restrict__ amax) {
  // (EPI) amax: the maximum buffer of `out` for the fp16 arithmetic of the next eval layer (bn_internal.h; zeroed by the host), or null
  extern __shared__ float sm[];
  __shared__ unsigned amax_sh[NT / 64];
  unsigned out_mag = 0;
  float* Rl = sm;                  // [9][W+2], columns -1 and W are zeros
  float* Tl = sm + 9 * (W + 2);    // [9][W]
  int t0 = blockIdx.x;
  const int h = t0 % H;
  t0 /= H;
  const int o = t0 % Co;
  const int b = t0 / Co;
  const long long HW = (long long)H * W;
  for (int idx = threadIdx.x; idx < 9 * W; idx += NT) {
    const int t = idx / W, w = idx - t * W;
    const long long src = (((long long)b * 9 + t) * Co + o) * HW + (long long)h * W + w;
    Rl[t * (W + 2) + 1 + w] = R[src];
    Tl[idx] = T[src];
  }
  if (threadIdx.x < 18) Rl[(threadIdx.x >> 1) * (W + 2) + ((threadIdx.x & 1) ? W + 1 : 0)] = 0.f;
  __syncthreads();
  float* ob = out + (((long long)b * Co + o) * D) * HW + (long long)h * W;
  float sc = 1.f, sh = 0.f;
  if (EPI) {
    sc = fold_scale(bn, o);
    sh = fold_shift(bn, o);
  }
  for (int w = threadIdx.x; w < W; w += NT) {
    float r[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) r[t] = Rl[t * (W + 2) + w + (t % 3)];  // column w' = w + kw - 1, stored at index w' + 1
    for (int d = 0; d < D; ++d) {  // (unrolling this loop by 4 was measured: 0.170 against 0.172 ms -- the stores bound it, not the LDS reads)
      float acc = 0.f;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int dp = d + kd - 1;
        if (dp < 0 || dp >= D) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int wp = w + kw - 1;
          if (wp >= dp && wp < W) acc += r[kd * 3 + kw] + Tl[(kd * 3 + kw) * W + wp - dp];
        }
      }
      if (EPI) {
        acc = fmaf(acc, sc, sh);
        if (bn.relu) acc = relu_nan(acc);
        out_mag = max(out_mag, mode::absmax_mag(acc));
      }
      ob[(long long)d * HW + w] = acc;
    }
  }
  if (EPI) {
    if (amax) mode::absmax_block_commit(out_mag, amax, amax_sh);  // (uniform)
  }
}

// Adjoint: gR_t[o][h][w'] = sum_d [0 <= d' < D][d' <= w'] gout[o][d][h][w'-kw+1],
//          gT_t[o][h][u ] = sum_d [0 <= d' < D][u+d' < W] gout[o][d][h][u+d+kd-kw]        (terms with a column outside [0, W) vanish).
// grid = B*Co*H blocks.  LDS = 4 + D*(W+4) floats: the gout rows of this (b, o, h) with 4 zero words behind every row (column W,
// and column -1 of the next row) and in front of the first.  The rows are fetched with all loads of a thread in flight (the
// first version staged them one dependent load at a time and took 0.62 ms; the 18 sums themselves are cheap): the three kd of
// one kw share a pass over the column for gR, and the (kd, kw) with the same kd - kw share a pass over the diagonal for gT.
// Sums over d ascending: deterministic.
__global__ __launch_bounds__(NTB) void cost_conv_assemble_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gR,
                                                                    float* __restrict__ gT, int B, int Co, int D, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int S = W + 4;
  float* gl = sm + 4;  // [D][S]
  int t0 = blockIdx.x;
  const int h = t0 % H;
  t0 /= H;
  const int o = t0 % Co;
  const int b = t0 / Co;
  const long long HW = (long long)H * W;
  const float* gb = gout + (((long long)b * Co + o) * D) * HW + (long long)h * W;
  if ((W & 3) == 0 && (reinterpret_cast<size_t>(gout) & 15) == 0) {
    const int W4 = W >> 2, total = D * W4;
    for (int base = 0; base < total; base += NTB * 8) {
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = base + j * NTB + threadIdx.x;
        const int d = idx / W4, w4 = idx - d * W4;
        v[j] = *reinterpret_cast<const float4*>(gb + (idx < total ? (long long)d * HW + w4 * 4 : 0));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int idx = base + j * NTB + threadIdx.x;
        const int d = idx / W4, w4 = idx - d * W4;
        if (idx < total) *reinterpret_cast<float4*>(gl + d * S + w4 * 4) = v[j];
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < D * W; idx += NTB) {
      const int d = idx / W, w = idx - d * W;
      gl[d * S + w] = gb[(long long)d * HW + w];
    }
  }
  for (int i = threadIdx.x; i < 4 * (D + 1); i += NTB) sm[(i >> 2) * S + (i & 3)] = 0.f;  // the pads: sm[0..3] and behind every row
  __syncthreads();
  // Two halves of the block share a column x: the first takes gR and the diagonal delta = -2, the second the diagonals -1 .. 2 (192
  // LDS reads each instead of 384 in one thread; every (kd, kw) sum belongs to exactly one of them).  The d loops are unrolled by 8: as
  // rolled loops every read waited for its own round trip to the LDS (0.41 ms per launch against 0.08 ms of HBM time for the 403 MB).
  const int part = threadIdx.x / (NTB / 2);
  for (int x = threadIdx.x % (NTB / 2); x < W; x += NTB / 2) {
    const long long dst = (((long long)b * 9) * Co + o) * HW + (long long)h * W + x;
    const long long tstride = (long long)Co * HW;
    // gR: one pass over column x - kw + 1 for the three kd
    if (part == 0) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float* col = gl + (x - kw + 1);
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 8
      for (int d = 0; d < D; ++d) {
        const float v = col[d * S];
        // d' = d + kd - 1 must lie in [0, D) and be <= x
        if (d >= 1 && d - 1 <= x) s0 += v;
        if (d <= x) s1 += v;
        if (d + 1 < D && d + 1 <= x) s2 += v;
      }
      gR[dst + (0 * 3 + kw) * tstride] = s0;
      gR[dst + (1 * 3 + kw) * tstride] = s1;
      gR[dst + (2 * 3 + kw) * tstride] = s2;
    }
    }
    // gT: one pass over the diagonal w = x + d + delta for every (kd, kw) with kd - kw = delta
    float st[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) st[t] = 0.f;
#pragma unroll
    for (int delta = -2; delta <= 2; ++delta) {
      if ((delta == -2) != (part == 0)) continue;  // (uniform per half of the block)
#pragma unroll 8
      for (int d = 0; d < D; ++d) {
        const int w = x + d + delta;
        const float v = (w >= -1 && w <= W) ? gl[d * S + w] : 0.f;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
          const int kw = kd - delta;
          if (kw < 0 || kw > 2) continue;
          const int dp = d + kd - 1;
          if (dp >= 0 && dp < D && x + dp < W) st[kd * 3 + kw] += v;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int delta = t / 3 - t % 3;  // kd - kw
      if ((delta == -2) == (part == 0)) gT[dst + t * tstride] = st[t];
    }
  }
}

int check_args(const void* a, const void* b, const void* c, int B, int Co, int D, int H, int W, const char* who) {
  MODE_REQUIRE(B >= 0 && Co > 0 && D > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE((long long)B * Co * H < (1ll << 31), MODE_ERR_UNSUPPORTED, "%s: too many rows", who);
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(a && b && c, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  return MODE_OK;
}

}  // namespace

namespace {
template <bool EPI>
int assemble_fwd(const float* R, const float* T, float* out, int B, int Co, int D, int H, int W, mode_stream_t stream,
                 const mode_bn_epilogue& bn, const char* who, float* out_absmax = nullptr) {
  int rc = check_args(R, T, out, B, Co, D, H, W, who);
  if (rc == MODE_OK && out_absmax) rc = mode::absmax_begin(out_absmax, mode::as_stream(stream), who);
  if (rc != MODE_OK || B == 0) return rc;
  const size_t lds = (size_t)(9 * (W + 2) + 9 * W) * sizeof(float);
  MODE_REQUIRE(lds <= 160 * 1024, MODE_ERR_UNSUPPORTED, "%s: W = %d too wide for the row buffers", who, W);
  rc = mode::allow_lds(cost_conv_assemble_fwd_kernel<EPI>, lds, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(cost_conv_assemble_fwd_kernel<EPI>, dim3(B * Co * H), dim3(NT), lds, mode::as_stream(stream), R, T, out, B, Co, D,
                     H, W, bn, reinterpret_cast<unsigned*>(out_absmax));
  return mode::check_launch(who);
}
}  // namespace

extern "C" int mode_cost_conv_assemble_fwd(const float* R, const float* T, float* out, int B, int Co, int D, int H, int W,
                                           mode_stream_t stream) {
  return assemble_fwd<false>(R, T, out, B, Co, D, H, W, stream, mode_bn_epilogue(), "mode_cost_conv_assemble_fwd");
}

extern "C" int mode_cost_conv_assemble_fwd_bn_amax(const float* R, const float* T, const mode_bn_epilogue* bn, float* out, float* out_absmax,
                                                   int B, int Co, int D, int H, int W, mode_stream_t stream) {
  const char* who = "mode_cost_conv_assemble_fwd_bn";
  int rc = mode::check_bn(bn, who);
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(bn->add == nullptr, MODE_ERR_UNSUPPORTED, "%s: no residual input on this layer", who);
  return assemble_fwd<true>(R, T, out, B, Co, D, H, W, stream, *bn, who, out_absmax);
}

extern "C" int mode_cost_conv_assemble_fwd_bn(const float* R, const float* T, const mode_bn_epilogue* bn, float* out, int B, int Co,
                                              int D, int H, int W, mode_stream_t stream) {
  return mode_cost_conv_assemble_fwd_bn_amax(R, T, bn, out, nullptr, B, Co, D, H, W, stream);
}

extern "C" int mode_cost_conv_assemble_bwd(const float* gout, float* gR, float* gT, int B, int Co, int D, int H, int W,
                                           mode_stream_t stream) {
  const char* who = "mode_cost_conv_assemble_bwd";
  int rc = check_args(gout, gR, gT, B, Co, D, H, W, who);
  if (rc != MODE_OK || B == 0) return rc;
  const size_t lds = (size_t)(4 + (size_t)D * (W + 4)) * sizeof(float);
  MODE_REQUIRE(lds <= 160 * 1024, MODE_ERR_UNSUPPORTED, "%s: D x W = %d x %d too large for the row buffer", who, D, W);
  rc = mode::allow_lds(cost_conv_assemble_bwd_kernel, lds, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(cost_conv_assemble_bwd_kernel, dim3(B * Co * H), dim3(NTB), lds, mode::as_stream(stream), gout, gR, gT, B, Co, D, H, W);
  return mode::check_launch(who);
}
