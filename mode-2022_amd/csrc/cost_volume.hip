// Concatenation cost volume for MODE's disparity stage (reference: models/mode_disparity.py:104-113).
//
// HBM-bound data movement.  Forward reads 2 feature maps (B,C,H,W) and writes the (B,2C,D4,H,W) volume:
// algorithmic bytes = 2*B*C*H*W*4 + B*2C*D4*H*W*4 (411.0 MB per sample at C=32, D4=48, 256x128).  Every
// output element is written exactly once with a 16-byte store; a wave's store instruction covers 1 KiB of
// contiguous volume.  The reference instead zero-fills the volume on the host, copies it to the device and
// issues 2*D4 strided slice copies.
#include "common.h"

namespace {

// One thread owns 4 consecutive w of one (b, c, h) row and streams all D4 disparity planes of both halves.
__global__ __launch_bounds__(256) void cost_volume_fwd_v4(const float* __restrict__ ref, const float* __restrict__ tgt,
                                                          float* __restrict__ cost, int B, int C, int D4, int H, int W) {
  const int W4 = W >> 2;
  const long long total = (long long)B * C * H * W4;
  const long long plane = (long long)H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(idx % W4) * 4;
    long long r = idx / W4;
    const int h = (int)(r % H);
    r /= H;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const long long row = (((long long)b * C + c) * H + h) * W;
    const float4 rv = *reinterpret_cast<const float4*>(ref + row + w);
    const float* trow = tgt + row;
    float* o_ref = cost + ((((long long)b * 2 * C + c) * D4) * H + h) * W + w;
    float* o_tgt = cost + ((((long long)b * 2 * C + C + c) * D4) * H + h) * W + w;
#pragma unroll 4
    for (int i = 0; i < D4; ++i) {
      float4 a, t;
      a.x = (w + 0 >= i) ? rv.x : 0.f;
      a.y = (w + 1 >= i) ? rv.y : 0.f;
      a.z = (w + 2 >= i) ? rv.z : 0.f;
      a.w = (w + 3 >= i) ? rv.w : 0.f;
      t.x = (w + 0 >= i) ? trow[w + 0 - i] : 0.f;
      t.y = (w + 1 >= i) ? trow[w + 1 - i] : 0.f;
      t.z = (w + 2 >= i) ? trow[w + 2 - i] : 0.f;
      t.w = (w + 3 >= i) ? trow[w + 3 - i] : 0.f;
      *reinterpret_cast<float4*>(o_ref + i * plane) = a;
      *reinterpret_cast<float4*>(o_tgt + i * plane) = t;
    }
  }
}

// Ragged widths (W % 4 != 0) or unaligned bases: one element per thread.
__global__ __launch_bounds__(256) void cost_volume_fwd_scalar(const float* __restrict__ ref, const float* __restrict__ tgt,
                                                              float* __restrict__ cost, int B, int C, int D4, int H, int W) {
  const long long total = (long long)B * C * H * W;
  const long long plane = (long long)H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(idx % W);
    long long r = idx / W;
    const int h = (int)(r % H);
    r /= H;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const long long row = (((long long)b * C + c) * H + h) * W;
    const float rv = ref[row + w];
    float* o_ref = cost + ((((long long)b * 2 * C + c) * D4) * H + h) * W + w;
    float* o_tgt = cost + ((((long long)b * 2 * C + C + c) * D4) * H + h) * W + w;
    for (int i = 0; i < D4; ++i) {
      o_ref[i * plane] = (w >= i) ? rv : 0.f;
      o_tgt[i * plane] = (w >= i) ? tgt[row + w - i] : 0.f;
    }
  }
}

// Backward: g_ref[w] = sum_{i<=w} g[c][i][w];  g_tgt[w] = sum_{i<W-w} g[C+c][i][w+i].
__global__ __launch_bounds__(256) void cost_volume_bwd_v4(const float* __restrict__ g, float* __restrict__ g_ref,
                                                          float* __restrict__ g_tgt, int B, int C, int D4, int H, int W) {
  const int W4 = W >> 2;
  const long long total = (long long)B * C * H * W4;
  const long long plane = (long long)H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(idx % W4) * 4;
    long long r = idx / W4;
    const int h = (int)(r % H);
    r /= H;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const float* gr = g + ((((long long)b * 2 * C + c) * D4) * H + h) * W + w;
    const float* gt = g + ((((long long)b * 2 * C + C + c) * D4) * H + h) * W;
    float4 sr = make_float4(0.f, 0.f, 0.f, 0.f), st = sr;
#pragma unroll 4
    for (int i = 0; i < D4; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(gr + i * plane);
      sr.x += (w + 0 >= i) ? v.x : 0.f;
      sr.y += (w + 1 >= i) ? v.y : 0.f;
      sr.z += (w + 2 >= i) ? v.z : 0.f;
      sr.w += (w + 3 >= i) ? v.w : 0.f;
      const float* p = gt + i * plane + i;
      st.x += (w + 0 + i < W) ? p[w + 0] : 0.f;
      st.y += (w + 1 + i < W) ? p[w + 1] : 0.f;
      st.z += (w + 2 + i < W) ? p[w + 2] : 0.f;
      st.w += (w + 3 + i < W) ? p[w + 3] : 0.f;
    }
    const long long row = (((long long)b * C + c) * H + h) * W + w;
    *reinterpret_cast<float4*>(g_ref + row) = sr;
    *reinterpret_cast<float4*>(g_tgt + row) = st;
  }
}

__global__ __launch_bounds__(256) void cost_volume_bwd_scalar(const float* __restrict__ g, float* __restrict__ g_ref,
                                                              float* __restrict__ g_tgt, int B, int C, int D4, int H, int W) {
  const long long total = (long long)B * C * H * W;
  const long long plane = (long long)H * W;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int w = (int)(idx % W);
    long long r = idx / W;
    const int h = (int)(r % H);
    r /= H;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    const float* gr = g + ((((long long)b * 2 * C + c) * D4) * H + h) * W + w;
    const float* gt = g + ((((long long)b * 2 * C + C + c) * D4) * H + h) * W + w;
    float sr = 0.f, st = 0.f;
    for (int i = 0; i < D4; ++i) {
      sr += (w >= i) ? gr[i * plane] : 0.f;
      st += (w + i < W) ? gt[i * plane + i] : 0.f;
    }
    const long long row = (((long long)b * C + c) * H + h) * W + w;
    g_ref[row] = sr;
    g_tgt[row] = st;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int mode_cost_volume_fwd(const float* ref, const float* tgt, float* cost, int B, int C, int D4, int H, int W,
                                    mode_stream_t stream) {
  MODE_REQUIRE(B >= 0 && C > 0 && D4 > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG,
               "mode_cost_volume_fwd: bad sizes B=%d C=%d D4=%d H=%d W=%d", B, C, D4, H, W);
  if (B == 0) return MODE_OK;  // empty batch: nothing to do (pointers may be null)
  MODE_REQUIRE(ref && tgt && cost, MODE_ERR_BAD_ARG, "mode_cost_volume_fwd: null pointer");
  const bool v4 = (W % 4 == 0) && aligned16(ref) && aligned16(tgt) && aligned16(cost);
  const long long n = (long long)B * C * H * (v4 ? W / 4 : W);
  const int grid = (int)std::min<long long>(mode::cdiv(n, 256), 1 << 20);
  if (v4)
    hipLaunchKernelGGL(cost_volume_fwd_v4, dim3(grid), dim3(256), 0, mode::as_stream(stream), ref, tgt, cost, B, C, D4, H, W);
  else
    hipLaunchKernelGGL(cost_volume_fwd_scalar, dim3(grid), dim3(256), 0, mode::as_stream(stream), ref, tgt, cost, B, C, D4, H,
                       W);
  return mode::check_launch("mode_cost_volume_fwd");
}

extern "C" int mode_cost_volume_bwd(const float* gcost, float* g_ref, float* g_tgt, int B, int C, int D4, int H, int W,
                                    mode_stream_t stream) {
  MODE_REQUIRE(B >= 0 && C > 0 && D4 > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG,
               "mode_cost_volume_bwd: bad sizes B=%d C=%d D4=%d H=%d W=%d", B, C, D4, H, W);
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(gcost && g_ref && g_tgt, MODE_ERR_BAD_ARG, "mode_cost_volume_bwd: null pointer");
  const bool v4 = (W % 4 == 0) && aligned16(gcost) && aligned16(g_ref) && aligned16(g_tgt);
  const long long n = (long long)B * C * H * (v4 ? W / 4 : W);
  const int grid = (int)std::min<long long>(mode::cdiv(n, 256), 1 << 20);
  if (v4)
    hipLaunchKernelGGL(cost_volume_bwd_v4, dim3(grid), dim3(256), 0, mode::as_stream(stream), gcost, g_ref, g_tgt, B, C, D4, H,
                       W);
  else
    hipLaunchKernelGGL(cost_volume_bwd_scalar, dim3(grid), dim3(256), 0, mode::as_stream(stream), gcost, g_ref, g_tgt, B, C, D4,
                       H, W);
  return mode::check_launch("mode_cost_volume_bwd");
}
