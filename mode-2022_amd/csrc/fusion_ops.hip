// The layers of the fusion network that are not a 3x3 convolution or a BatchNorm: 2x2 max-pooling, the
// rearrangement half of the 2x2 stride-2 transposed convolution, and the single-channel 1x1 head with its sigmoid.  gfx950; all of them
// stream their tensors once (HBM-bound), fp32.
//
// Reference: models/mode_fusion.py -- nn.MaxPool2d(2, stride=2) in front of the encoder blocks (:146, :161, :190), nn.ConvTranspose2d(planes,
// planes / 2, 2, 2) + BatchNorm2d + ReLU behind the decoder blocks (:195-197, :212-214), nn.Conv2d(planes, 1, 1, bias=True) + nn.Sigmoid at
// the end (:228-229); MIOpen / ATen there.
//
//   * A transposed convolution with kernel 2 and stride 2 has no overlapping taps: y[b, o, 2h+i, 2w+j] = bias[o] + sum_c x[b, c, h, w] *
//     W[c, o, i, j] is ONE 1x1 convolution with 4 Co output channels (row r = 4 o + 2 i + j of the weight's own (Ci, 4 Co) storage) followed
//     by a depth-to-space rearrangement.  The GEMM runs on csrc/conv1x1.hip (MFMA; forward, input gradient and weight gradient exist
//     there); this file holds the rearrangement with the per-channel affine (bias, or the folded eval-mode BatchNorm) and ReLU in its
//     store, and its inverse for the backward pass.
//   * Max-pooling picks like torch (max_pool2d_with_indices): scan order (0,0), (0,1), (1,0), (1,1), a later value replaces the
//     current one when it is greater or NaN; the backward recomputes that choice from x (no index tensor).
//   * The head: s[b, q] = sigmoid(bias + sum_c w[c] x[b, c, q]) over NCHW planes -- one pass over x; its backward one pass that writes
//     gx and leaves per-block partial sums of gw / gbias, reduced in fixed order by a second launch (deterministic, no atomics).
#include "common.h"

#include <algorithm>

namespace {

constexpr int NT = 256;

__device__ __forceinline__ bool takes(float v, float m) { return v > m || v != v; }

// one thread per output element pair (2 consecutive wo): x (N, H, W) -> y (N, Ho, Wo), Ho = H / 2, Wo = W / 2 (floor)
__global__ __launch_bounds__(NT) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long N, int H, int W, int Ho,
                                                         int Wo) {
  const int Wp = (Wo + 1) >> 1;
  const long long total = N * Ho * Wp;
  for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < total; idx += (long long)gridDim.x * NT) {
    const int wp = (int)(idx % Wp);
    const long long r = idx / Wp;
    const int ho = (int)(r % Ho);
    const long long n = r / Ho;
    const float* r0 = x + (n * H + 2 * ho) * W + 4 * wp;
    const float* r1 = r0 + W;
    float* out = y + (n * Ho + ho) * Wo + 2 * wp;
    const bool two = 2 * wp + 1 < Wo;
    float a[4], b[4];
    if (two && ((W & 3) == 0) && ((reinterpret_cast<size_t>(x) & 15) == 0)) {
      const float4 u = *reinterpret_cast<const float4*>(r0), v = *reinterpret_cast<const float4*>(r1);
      a[0] = u.x; a[1] = u.y; a[2] = u.z; a[3] = u.w;
      b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = j < 2 || two;
        a[j] = in ? r0[j] : 0.f;
        b[j] = in ? r1[j] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k == 1 && !two) break;
      float m = a[2 * k];
      if (takes(a[2 * k + 1], m)) m = a[2 * k + 1];
      if (takes(b[2 * k], m)) m = b[2 * k];
      if (takes(b[2 * k + 1], m)) m = b[2 * k + 1];
      out[k] = m;
    }
  }
}

// one thread per 2 x 2 block of x (including the incomplete blocks of an odd last row / column, which get zeros)
__global__ __launch_bounds__(NT) void maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                                                         long long N, int H, int W, int Ho, int Wo) {
  const int Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
  const long long total = N * Hb * Wb;
  for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < total; idx += (long long)gridDim.x * NT) {
    const int wb = (int)(idx % Wb);
    const long long r = idx / Wb;
    const int hb = (int)(r % Hb);
    const long long n = r / Hb;
    const long long base = (n * H + 2 * hb) * W + 2 * wb;
    const bool full = hb < Ho && wb < Wo;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (full) {
      const float v[4] = {x[base], x[base + 1], x[base + W], x[base + W + 1]};
      int arg = 0;
      float m = v[0];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (takes(v[k], m)) {
          m = v[k];
          arg = k;
        }
      const float go = gy[(n * Ho + hb) * Wo + wb];
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = k == arg ? go : 0.f;
    }
    const bool c1 = 2 * wb + 1 < W, r1 = 2 * hb + 1 < H;
    gx[base] = g[0];
    if (c1) gx[base + 1] = g[1];
    if (r1) gx[base + W] = g[2];
    if (r1 && c1) gx[base + W + 1] = g[3];
  }
}

// y[b, o, 2h+i, 2w+j] = act(scale[o] * y4[b, 4 o + 2 i + j, h, w] + shift[o]); one thread per (b, o, h, w); scale may be null (1)
__global__ __launch_bounds__(NT) void shuffle2_kernel(const float* __restrict__ y4, const float* __restrict__ scale, const float* __restrict__ shift,
                                                     float* __restrict__ y, long long BC, int Co, int H, int W, int relu) {
  const long long HW = (long long)H * W, total = BC * HW;
  for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < total; idx += (long long)gridDim.x * NT) {
    const int w = (int)(idx % W);
    const long long r = idx / W;
    const int h = (int)(r % H);
    const long long bo = r / H;
    const int o = (int)(bo % Co);
    const float a = scale ? scale[o] : 1.f, s = shift ? shift[o] : 0.f;
    const float* src = y4 + (bo * 4) * HW + (long long)h * W + w;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float t = __builtin_fmaf(a, src[k * HW], s);
      v[k] = relu ? relu_nan(t) : t;
    }
    float* dst = y + (bo * 2 * H + 2 * h) * (2LL * W) + 2 * w;
    *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
    *reinterpret_cast<float2*>(dst + 2 * W) = make_float2(v[2], v[3]);
  }
}

// g4[b, 4 o + 2 i + j, h, w] = gy[b, o, 2h+i, 2w+j]
__global__ __launch_bounds__(NT) void unshuffle2_kernel(const float* __restrict__ gy, float* __restrict__ g4, long long BC, int H, int W) {
  const long long HW = (long long)H * W, total = BC * HW;
  for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < total; idx += (long long)gridDim.x * NT) {
    const int w = (int)(idx % W);
    const long long r = idx / W;
    const int h = (int)(r % H);
    const long long bo = r / H;
    const float* src = gy + (bo * 2 * H + 2 * h) * (2LL * W) + 2 * w;
    const float2 t0 = *reinterpret_cast<const float2*>(src), t1 = *reinterpret_cast<const float2*>(src + 2 * W);
    float* dst = g4 + (bo * 4) * HW + (long long)h * W + w;
    dst[0] = t0.x;
    dst[HW] = t0.y;
    dst[2 * HW] = t1.x;
    dst[3 * HW] = t1.y;
  }
}

constexpr int HEAD_MAXC = 64;

// s[b, q] = sigmoid(bias + sum_c w[c] x[b, c, q]); one thread per 4 consecutive pixels (S % 4 == 0, 16-byte aligned planes)
__global__ __launch_bounds__(NT) void head1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ s, int B, int C, long long S) {
  __shared__ float wl[HEAD_MAXC];
  if (threadIdx.x < C) wl[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const float b0 = bias ? bias[0] : 0.f;
  const long long S4 = S >> 2, total = (long long)B * S4;
  for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < total; idx += (long long)gridDim.x * NT) {
    const long long q4 = idx % S4;
    const int b = (int)(idx / S4);
    const float4* xp = reinterpret_cast<const float4*>(x + (long long)b * C * S) + q4;
    float4 z = make_float4(b0, b0, b0, b0);
    for (int c = 0; c < C; ++c) {
      const float4 v = xp[(long long)c * S4];
      const float wc = wl[c];
      z.x = __builtin_fmaf(wc, v.x, z.x);
      z.y = __builtin_fmaf(wc, v.y, z.y);
      z.z = __builtin_fmaf(wc, v.z, z.z);
      z.w = __builtin_fmaf(wc, v.w, z.w);
    }
    float4 o;
    o.x = 1.f / (1.f + __expf(-z.x));
    o.y = 1.f / (1.f + __expf(-z.y));
    o.z = 1.f / (1.f + __expf(-z.z));
    o.w = 1.f / (1.f + __expf(-z.w));
    reinterpret_cast<float4*>(s + (long long)b * S)[q4] = o;
  }
}

// g = gs * s (1 - s);  gx[b, c, q] = g w[c];  partial[block][c] = sum over the block's pixels of g x[b, c, q], partial[block][C] = sum g
__global__ __launch_bounds__(NT) void head1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ s,
                                                      const float* __restrict__ gs, float* __restrict__ gx, float* __restrict__ partial, int B,
                                                      int C, long long S) {
  __shared__ float wl[HEAD_MAXC];
  __shared__ float red[NT / 64][HEAD_MAXC + 1];
  if (threadIdx.x < C) wl[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const long long S4 = S >> 2, total = (long long)B * S4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c0 = 0; c0 <= C; c0 += 16) {  // 16 channel sums per sweep (register budget); sweep 0 also writes gx
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < total; idx += (long long)gridDim.x * NT) {
      const long long q4 = idx % S4;
      const int b = (int)(idx / S4);
      const float4 sv = reinterpret_cast<const float4*>(s + (long long)b * S)[q4];
      const float4 gv = reinterpret_cast<const float4*>(gs + (long long)b * S)[q4];
      float4 g;
      g.x = gv.x * sv.x * (1.f - sv.x);
      g.y = gv.y * sv.y * (1.f - sv.y);
      g.z = gv.z * sv.z * (1.f - sv.z);
      g.w = gv.w * sv.w * (1.f - sv.w);
      const float4* xp = reinterpret_cast<const float4*>(x + (long long)b * C * S) + q4;
      float4* gp = reinterpret_cast<float4*>(gx + (long long)b * C * S) + q4;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int c = c0 + k;
        if (c < C) {
          const float4 v = xp[(long long)c * S4];
          acc[k] += g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w;
          if (gx) {
            const float wc = wl[c];
            gp[(long long)c * S4] = make_float4(g.x * wc, g.y * wc, g.z * wc, g.w * wc);
          }
        } else if (c == C) {
          acc[k] += g.x + g.y + g.z + g.w;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a = acc[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
      if (lane == 0 && c0 + k <= C) red[wave][c0 + k] = a;
    }
  }
  __syncthreads();
  if (threadIdx.x <= C) {
    float a = 0.f;
#pragma unroll
    for (int wv = 0; wv < NT / 64; ++wv) a += red[wv][threadIdx.x];
    partial[(long long)blockIdx.x * (HEAD_MAXC + 1) + threadIdx.x] = a;
  }
}

__global__ __launch_bounds__(64) void head1_reduce_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ gw,
                                                         float* __restrict__ gbias, int C, int accumulate) {
  const int c = blockIdx.x;  // 0..C
  double a = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) a += (double)partial[(long long)i * (HEAD_MAXC + 1) + c];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
  if (threadIdx.x == 0) {
    float* dst = c < C ? gw + c : gbias;
    if (dst) *dst = accumulate ? *dst + (float)a : (float)a;
  }
}

inline int grid_for(long long items) { return (int)std::max<long long>(1, std::min<long long>(mode::cdiv(items, NT), 32LL * kNumCU)); }

}  // namespace

extern "C" int mode_maxpool2x2_fwd(const float* x, float* y, long long N, int H, int W, mode_stream_t stream) {
  MODE_REQUIRE(N >= 0 && H >= 2 && W >= 2, MODE_ERR_BAD_ARG, "mode_maxpool2x2_fwd: planes of at least 2 x 2");
  if (N == 0) return MODE_OK;
  MODE_REQUIRE(x && y, MODE_ERR_BAD_ARG, "mode_maxpool2x2_fwd: null pointer");
  const int Ho = H / 2, Wo = W / 2;
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(grid_for(N * Ho * ((Wo + 1) / 2))), dim3(NT), 0, mode::as_stream(stream), x, y, N, H, W, Ho, Wo);
  return mode::check_launch("mode_maxpool2x2_fwd");
}

extern "C" int mode_maxpool2x2_bwd(const float* x, const float* gy, float* gx, long long N, int H, int W, mode_stream_t stream) {
  MODE_REQUIRE(N >= 0 && H >= 2 && W >= 2, MODE_ERR_BAD_ARG, "mode_maxpool2x2_bwd: planes of at least 2 x 2");
  if (N == 0) return MODE_OK;
  MODE_REQUIRE(x && gy && gx, MODE_ERR_BAD_ARG, "mode_maxpool2x2_bwd: null pointer");
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(grid_for(N * ((H + 1) / 2) * ((W + 1) / 2))), dim3(NT), 0, mode::as_stream(stream), x, gy, gx, N, H,
                     W, H / 2, W / 2);
  return mode::check_launch("mode_maxpool2x2_bwd");
}

extern "C" int mode_depth_to_space2(const float* y4, const float* scale, const float* shift, float* y, int B, int Co, int H, int W, int relu,
                                    mode_stream_t stream) {
  MODE_REQUIRE(B >= 0 && Co > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "mode_depth_to_space2: non-positive size");
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(y4 && y, MODE_ERR_BAD_ARG, "mode_depth_to_space2: null pointer");
  MODE_REQUIRE((reinterpret_cast<size_t>(y) & 7) == 0, MODE_ERR_UNSUPPORTED, "mode_depth_to_space2: the output must be 8-byte aligned");
  hipLaunchKernelGGL(shuffle2_kernel, dim3(grid_for((long long)B * Co * H * W)), dim3(NT), 0, mode::as_stream(stream), y4, scale, shift, y,
                     (long long)B * Co, Co, H, W, relu);
  return mode::check_launch("mode_depth_to_space2");
}

extern "C" int mode_space_to_depth2(const float* gy, float* g4, int B, int Co, int H, int W, mode_stream_t stream) {
  MODE_REQUIRE(B >= 0 && Co > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "mode_space_to_depth2: non-positive size");
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(gy && g4, MODE_ERR_BAD_ARG, "mode_space_to_depth2: null pointer");
  MODE_REQUIRE((reinterpret_cast<size_t>(gy) & 7) == 0, MODE_ERR_UNSUPPORTED, "mode_space_to_depth2: the input must be 8-byte aligned");
  hipLaunchKernelGGL(unshuffle2_kernel, dim3(grid_for((long long)B * Co * H * W)), dim3(NT), 0, mode::as_stream(stream), gy, g4, (long long)B * Co, H,
                     W);
  return mode::check_launch("mode_space_to_depth2");
}

static int check_head1(const char* who, int B, int C, long long S) {
  MODE_REQUIRE(B >= 0 && C > 0 && S > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(C <= HEAD_MAXC, MODE_ERR_UNSUPPORTED, "%s: %d input channels (at most %d)", who, C, HEAD_MAXC);
  MODE_REQUIRE(S % 4 == 0, MODE_ERR_UNSUPPORTED, "%s: planes of a multiple of 4 elements", who);
  return MODE_OK;
}

extern "C" int mode_conv1x1_sigmoid_fwd(const float* x, const float* w, const float* bias, float* s, int B, int C, long long S,
                                        mode_stream_t stream) {
  int rc = check_head1("mode_conv1x1_sigmoid_fwd", B, C, S);
  if (rc != MODE_OK || B == 0) return rc;
  MODE_REQUIRE(x && w && s, MODE_ERR_BAD_ARG, "mode_conv1x1_sigmoid_fwd: null pointer");
  MODE_REQUIRE(((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(s)) & 15) == 0, MODE_ERR_UNSUPPORTED,
               "mode_conv1x1_sigmoid_fwd: tensors must be 16-byte aligned");
  hipLaunchKernelGGL(head1_fwd_kernel, dim3(grid_for((long long)B * (S / 4))), dim3(NT), 0, mode::as_stream(stream), x, w, bias, s, B, C, S);
  return mode::check_launch("mode_conv1x1_sigmoid_fwd");
}

extern "C" size_t mode_conv1x1_sigmoid_bwd_workspace_bytes(int B, long long S) {
  if (B <= 0 || S <= 0) return 0;
  return (size_t)std::min<long long>(grid_for((long long)B * (S / 4)), 4 * kNumCU) * (HEAD_MAXC + 1) * sizeof(float);
}

extern "C" int mode_conv1x1_sigmoid_bwd(const float* x, const float* w, const float* s, const float* gs, float* gx, float* gw, float* gbias,
                                        int accumulate, float* workspace, int B, int C, long long S, mode_stream_t stream) {
  int rc = check_head1("mode_conv1x1_sigmoid_bwd", B, C, S);
  if (rc != MODE_OK || B == 0) return rc;
  MODE_REQUIRE(x && w && s && gs && gw && workspace, MODE_ERR_BAD_ARG, "mode_conv1x1_sigmoid_bwd: null pointer");
  MODE_REQUIRE(((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(s) | reinterpret_cast<size_t>(gs) | reinterpret_cast<size_t>(gx)) & 15) == 0,
               MODE_ERR_UNSUPPORTED, "mode_conv1x1_sigmoid_bwd: tensors must be 16-byte aligned");
  const int nb = (int)std::min<long long>(grid_for((long long)B * (S / 4)), 4 * kNumCU);
  hipStream_t st = mode::as_stream(stream);
  hipLaunchKernelGGL(head1_bwd_kernel, dim3(nb), dim3(NT), 0, st, x, w, s, gs, gx, workspace, B, C, S);
  rc = mode::check_launch("mode_conv1x1_sigmoid_bwd");
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(head1_reduce_kernel, dim3(C + 1), dim3(64), 0, st, workspace, nb, gw, gbias, C, accumulate);
  return mode::check_launch("mode_conv1x1_sigmoid_bwd(reduce)");
}
