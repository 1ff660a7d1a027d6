// Export-stage geometry of the disparity network: disparity -> depth, Cassini re-projections and
// the depth view transform with its z-buffer, gfx950.
//
// Reference (host numpy + a CPU<->GPU ping-pong per call + a sequential numba loop):
//   save_output_disparity_stage.py:105-160   disp2depth
//   utils/geometry.py:7-45, 48-96, 160-198   cassini2Equirec / rotateCassini / erp2rect_cassini: angle maps on the host,
//                                            then F.grid_sample(bilinear, align_corners=True, padding_mode='border')
//   utils/geometry.py:99-145                 depthViewTransWithConf: 3D re-projection of every pixel
//   utils/geometry.py:148-156                __iterPixels_with_conf: sequential z-buffer scatter (numba)
// Here the maps the reference builds with numpy stay on the host (they depend on the image size only and are cached by the
// caller); everything per pixel runs on the GPU with no host round trip.  All three kernels are HBM-bound elementwise /
// gather / scatter work: 4-16 bytes per pixel and direction.
#include "common.h"

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------------------------------------------------------------
// disparity -> depth by the sine rule (save_output_disparity_stage.py:118-135), float32 arithmetic like numpy's:
//   phi_l = float(start + j * (-step))   (np.arange in float64, then .astype(float32))
//   phi_r = disp * pi / W + phi_l        depth = baseline * sin(pi/2 - phi_r) / sin(phi_r - phi_l)
//   disp == 0 -> 1000 (masked, filled);  depth > 1000 -> 1000;  depth < 0 -> 0
__global__ __launch_bounds__(NT) void disp2depth_kernel(const float* __restrict__ disp, float* __restrict__ depth, long long n, int W,
                                                        float baseline) {
  const float pi_f = 3.14159265358979323846f, half_pi_f = 1.57079632679489661923f;
  const double start = 0.5 * 3.14159265358979323846 - (0.5 * 3.14159265358979323846 / W);
  const double step = 3.14159265358979323846 / W;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
    const int j = (int)(i % W);
    const float d = disp[i];
    float out = 1000.f;
    if (d != 0.f) {
      const float phi_l = (float)(start + (double)j * (-step));
      const float phi_r = d * pi_f / (float)W + phi_l;
      out = baseline * sinf(half_pi_f - phi_r) / sinf(phi_r - phi_l);
      if (out > 1000.f) out = 1000.f;
      if (out < 0.f) out = 0.f;  // NaN stays NaN, as in numpy
    }
    depth[i] = out;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// F.grid_sample(src, grid, mode='bilinear', padding_mode='border', align_corners=True) for (N, C, Hs, Ws) -> (N, C, Ho, Wo);
// grid (Ng, Ho, Wo, 2) with Ng = N or 1 (shared by all samples: the reference repeat_interleaves one grid).
__global__ __launch_bounds__(NT) void grid_sample_border_kernel(const float* __restrict__ src, const float* __restrict__ grid,
                                                                float* __restrict__ dst, int N, int C, int Hs, int Ws, int Ho,
                                                                int Wo, int Ng) {
  const long long npix = (long long)Ho * Wo;
  const long long total = (long long)N * npix;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
    const int n = (int)(i / npix);
    const long long p = i - (long long)n * npix;
    const float2 g = reinterpret_cast<const float2*>(grid)[(Ng == 1 ? 0 : (long long)n * npix) + p];
    // unnormalise (align_corners=True), clip to the border
    float x = (g.x + 1.f) * 0.5f * (float)(Ws - 1);
    float y = (g.y + 1.f) * 0.5f * (float)(Hs - 1);
    x = fminf(fmaxf(x, 0.f), (float)(Ws - 1));
    y = fminf(fmaxf(y, 0.f), (float)(Hs - 1));
    const float xf = floorf(x), yf = floorf(y);
    const int x0 = (int)xf, y0 = (int)yf;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = x - xf, wy1 = y - yf;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    // corner weights in torch's order (nw, ne, sw, se); corners outside contribute nothing
    const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    const bool x1ok = x1 <= Ws - 1, y1ok = y1 <= Hs - 1;
    const float* sp = src + (long long)n * C * Hs * Ws;
    float* dp = dst + (long long)n * C * npix + p;
    for (int c = 0; c < C; ++c) {
      const float* s = sp + (long long)c * Hs * Ws;
      float v = s[(long long)y0 * Ws + x0] * nw;
      if (x1ok) v += s[(long long)y0 * Ws + x1] * ne;
      if (y1ok) v += s[(long long)y1 * Ws + x0] * sw;
      if (x1ok && y1ok) v += s[(long long)y1 * Ws + x1] * se;
      dp[(long long)c * npix] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// depthViewTransWithConf.  Pass 1, one thread per SOURCE pixel (i, j) with r1 > 0:
//   X1 = r1 * dir(i, j)  (float32 products in numpy's order)  X2 = R (X1 - t)  (float64)       r2 = |X2|
//   I = clip(rint(H/2 - H * atan2(X2.y, X2.z) / (2 pi)), 0, H-1)      J = clip(rint(W/2 - W * asin(clip(X2.x / r2)) / pi), 0, W-1)
// and a 64-bit atomic min of a key into key[I][J].
// The reference scans the sources in row-major order and overwrites the target when r2 < view2[target] -- r2 in float64 against
// the float32 value stored so far (initial value 100000).  Let m be the smallest float32(r2) over the sources of a target
// (positive floats order like their bit patterns) and S the sources that round to m.  The first source of S always gets
// stored (rounding is monotonic); a later source of S replaces it only if its float64 r2 is strictly below the float32 m.
// Hence the survivor is the LAST source of L = {k in S : r2_k < m}, or the FIRST source of S when L is empty, and
//   key = bits(m) << 32 | (k in L ? 0 : 1) << 31 | (k in L ? N-1-k : k)
// has exactly that source as its minimum -- bit-identical to the sequential loop, in any execution order.
// Pass 2, one thread per TARGET pixel: view2 = r2 of the winner (0 if none; capped at 1000), conf2 = conf1[winner] (0 if none).
struct ViewXform {
  double R[9];
  double t[3];
};

// projection of one source pixel: returns false if it takes no part (r1 <= 0, r2 not below the initial 100000, r2 == 0)
__device__ __forceinline__ bool project_pixel(float r1, float sin_phi, float cos_phi, float sin_theta, float cos_theta,
                                              const ViewXform& xf, int H, int W, double& r2, long long& tgt) {
#pragma clang fp contract(off)  // numpy's matmul / sum of squares round every product: no fused multiply-adds here
  const double PI = 3.14159265358979323846;
  if (!(r1 > 0.f)) return false;
  // float32 products in numpy's order (geometry.py:126-128): r * sin(phi);  (r * cos(phi)) * sin(theta);  (r * cos(phi)) * cos(theta)
  const float rc = r1 * cos_phi;
  const float x1 = r1 * sin_phi, y1 = rc * sin_theta, z1 = rc * cos_theta;
  const double ax = (double)x1 - xf.t[0], ay = (double)y1 - xf.t[1], az = (double)z1 - xf.t[2];
  const double X = xf.R[0] * ax + xf.R[1] * ay + xf.R[2] * az;
  const double Y = xf.R[3] * ax + xf.R[4] * ay + xf.R[5] * az;
  const double Z = xf.R[6] * ax + xf.R[7] * ay + xf.R[8] * az;
  r2 = sqrt(X * X + Y * Y + Z * Z);
  if (!(r2 < 100000.0)) return false;  // never below the initial value (also drops NaN)
  const double theta = atan2(Y, Z);
  double sphi = X / r2;
  sphi = sphi < -1.0 ? -1.0 : (sphi > 1.0 ? 1.0 : sphi);
  const double phi = asin(sphi);
  double fi = rint((double)H / 2 - (double)H * theta / (2 * PI));
  double fj = rint((double)W / 2 - (double)W * phi / PI);
  fi = fi < 0.0 ? 0.0 : (fi > (double)(H - 1) ? (double)(H - 1) : fi);
  fj = fj < 0.0 ? 0.0 : (fj > (double)(W - 1) ? (double)(W - 1) : fj);
  if (!(fi == fi) || !(fj == fj)) return false;  // r2 == 0: NaN angles; numpy's int16 cast of NaN is platform noise
  tgt = (long long)fi * W + (long long)fj;
  return true;
}

// z-buffer key of source `idx` (of n) with radius r2, see above
__device__ __forceinline__ unsigned long long zkey(double r2, long long idx, long long n) {
  const float m = (float)r2;
  const bool inL = r2 < (double)m;
  const unsigned lo = inL ? (unsigned)(n - 1 - idx) : (0x80000000u | (unsigned)idx);
  return ((unsigned long long)__float_as_uint(m) << 32) | lo;
}

__global__ __launch_bounds__(NT) void view_trans_scatter_kernel(const float* __restrict__ view1, const float* __restrict__ trig,
                                                                unsigned long long* __restrict__ keys, int H, int W, ViewXform xf) {
  const long long n = (long long)H * W;
  for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < n; idx += (long long)gridDim.x * NT) {
    double r2;
    long long tgt;
    const int i = (int)(idx / W), j = (int)(idx - (long long)i * W);
    if (project_pixel(view1[idx], trig[j], trig[W + j], trig[2 * W + i], trig[2 * W + H + i], xf, H, W, r2, tgt))
      atomicMin(keys + tgt, zkey(r2, idx, n));
  }
}

// the two halves on their own (tests, and callers that bring their own projection): projection to (r2, target index or -1) ...
__global__ __launch_bounds__(NT) void view_project_kernel(const float* __restrict__ view1, const float* __restrict__ trig,
                                                          double* __restrict__ r2_out, int* __restrict__ tgt_out, int H, int W, ViewXform xf) {
  const long long n = (long long)H * W;
  for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < n; idx += (long long)gridDim.x * NT) {
    double r2 = 0.0;
    long long tgt = -1;
    const int i = (int)(idx / W), j = (int)(idx - (long long)i * W);
    const bool ok = project_pixel(view1[idx], trig[j], trig[W + j], trig[2 * W + i], trig[2 * W + H + i], xf, H, W, r2, tgt);
    r2_out[idx] = r2;
    tgt_out[idx] = ok ? (int)tgt : -1;
  }
}

// ... and the scatter of given (r2, target) pairs
__global__ __launch_bounds__(NT) void zbuffer_scatter_kernel(const double* __restrict__ r2, const int* __restrict__ tgt,
                                                             unsigned long long* __restrict__ keys, long long n) {
  for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < n; idx += (long long)gridDim.x * NT) {
    const int t = tgt[idx];
    if (t >= 0 && t < n && r2[idx] < 100000.0) atomicMin(keys + t, zkey(r2[idx], idx, n));
  }
}

__global__ __launch_bounds__(NT) void view_trans_resolve_kernel(const unsigned long long* __restrict__ keys,
                                                                const float* __restrict__ conf1, float* __restrict__ view2,
                                                                float* __restrict__ conf2, long long n) {
  for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < n; idx += (long long)gridDim.x * NT) {
    const unsigned long long k = keys[idx];
    float v = 0.f, c = 0.f;
    if (k != ~0ull) {
      v = __uint_as_float((unsigned)(k >> 32));
      const unsigned lo = (unsigned)(k & 0xffffffffull);
      c = conf1[(lo & 0x80000000u) ? (long long)(lo & 0x7fffffffu) : n - 1 - (long long)lo];
      if (v == 100000.f) v = 0.f;  // "view_2[view_2 == 100000] = 0"
      if (v > 1000.f) v = 1000.f;
    }
    view2[idx] = v;
    conf2[idx] = c;
  }
}

int grid_for(long long n) { return (int)std::min<long long>(mode::cdiv(n, NT), 8LL * kNumCU); }

}  // namespace

extern "C" int mode_disp2depth(const float* disp, float* depth, int H, int W, float baseline, mode_stream_t stream) {
  MODE_REQUIRE(H >= 0 && W > 0, MODE_ERR_BAD_ARG, "mode_disp2depth: bad size %dx%d", H, W);
  if (H == 0) return MODE_OK;
  MODE_REQUIRE(disp && depth, MODE_ERR_BAD_ARG, "mode_disp2depth: null pointer");
  const long long n = (long long)H * W;
  hipLaunchKernelGGL(disp2depth_kernel, dim3(grid_for(n)), dim3(NT), 0, mode::as_stream(stream), disp, depth, n, W, baseline);
  return mode::check_launch("mode_disp2depth");
}

extern "C" int mode_grid_sample_border(const float* src, const float* grid, float* dst, int N, int C, int Hs, int Ws, int Ho, int Wo,
                                       int grids, mode_stream_t stream) {
  MODE_REQUIRE(N >= 0 && C > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, MODE_ERR_BAD_ARG, "mode_grid_sample_border: non-positive size");
  MODE_REQUIRE(grids == 1 || grids == N, MODE_ERR_BAD_ARG, "mode_grid_sample_border: %d grids for %d samples", grids, N);
  if (N == 0) return MODE_OK;
  MODE_REQUIRE(src && grid && dst, MODE_ERR_BAD_ARG, "mode_grid_sample_border: null pointer");
  MODE_REQUIRE((reinterpret_cast<uintptr_t>(grid) & 7) == 0, MODE_ERR_UNSUPPORTED, "mode_grid_sample_border: grid must be 8-byte aligned");
  const long long n = (long long)N * Ho * Wo;
  hipLaunchKernelGGL(grid_sample_border_kernel, dim3(grid_for(n)), dim3(NT), 0, mode::as_stream(stream), src, grid, dst, N, C, Hs, Ws,
                     Ho, Wo, grids);
  return mode::check_launch("mode_grid_sample_border");
}

extern "C" size_t mode_depth_view_trans_workspace_bytes(int H, int W) {
  return H > 0 && W > 0 ? (size_t)H * W * sizeof(unsigned long long) : 0;
}

extern "C" int mode_depth_view_trans(const float* view1, const float* conf1, const float* trig, const double* R, const double* t,
                                     float* view2, float* conf2, void* workspace, int H, int W, mode_stream_t stream) {
  MODE_REQUIRE(H > 0 && W > 0 && (long long)H * W < (1LL << 31), MODE_ERR_BAD_ARG, "mode_depth_view_trans: bad size %dx%d", H, W);
  MODE_REQUIRE(view1 && conf1 && trig && R && t && view2 && conf2, MODE_ERR_BAD_ARG, "mode_depth_view_trans: null pointer");
  MODE_REQUIRE(workspace, MODE_ERR_WORKSPACE, "mode_depth_view_trans: workspace required");
  MODE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 7) == 0, MODE_ERR_UNSUPPORTED, "mode_depth_view_trans: unaligned workspace");
  hipStream_t st = mode::as_stream(stream);
  const long long n = (long long)H * W;
  ViewXform xf;
  for (int i = 0; i < 9; ++i) xf.R[i] = R[i];
  for (int i = 0; i < 3; ++i) xf.t[i] = t[i];
  int frc = mode::fill_words(workspace, 0xffffffffu, 2 * (size_t)n, st, "mode_depth_view_trans");  // all keys = +inf (a kernel, not hipMemsetAsync: common.h)
  if (frc != MODE_OK) return frc;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);
  hipLaunchKernelGGL(view_trans_scatter_kernel, dim3(grid_for(n)), dim3(NT), 0, st, view1, trig, keys, H, W, xf);
  hipLaunchKernelGGL(view_trans_resolve_kernel, dim3(grid_for(n)), dim3(NT), 0, st, keys, conf1, view2, conf2, n);
  return mode::check_launch("mode_depth_view_trans");
}

// The two halves of mode_depth_view_trans on their own: per-source projection (r2 as float64, target index i*W + j, or -1 for a
// source that takes no part), and the z-buffer over given (r2, target) pairs with the reference's exact overwrite rule.
extern "C" int mode_depth_view_project(const float* view1, const float* trig, const double* R, const double* t, double* r2, int32_t* target,
                                       int H, int W, mode_stream_t stream) {
  MODE_REQUIRE(H > 0 && W > 0 && (long long)H * W < (1LL << 31), MODE_ERR_BAD_ARG, "mode_depth_view_project: bad size %dx%d", H, W);
  MODE_REQUIRE(view1 && trig && R && t && r2 && target, MODE_ERR_BAD_ARG, "mode_depth_view_project: null pointer");
  ViewXform xf;
  for (int i = 0; i < 9; ++i) xf.R[i] = R[i];
  for (int i = 0; i < 3; ++i) xf.t[i] = t[i];
  hipLaunchKernelGGL(view_project_kernel, dim3(grid_for((long long)H * W)), dim3(NT), 0, mode::as_stream(stream), view1, trig, r2, target, H,
                     W, xf);
  return mode::check_launch("mode_depth_view_project");
}

extern "C" int mode_zbuffer(const double* r2, const int32_t* target, const float* conf1, float* view2, float* conf2, void* workspace,
                            long long n, mode_stream_t stream) {
  MODE_REQUIRE(n > 0 && n < (1LL << 31), MODE_ERR_BAD_ARG, "mode_zbuffer: bad size");
  MODE_REQUIRE(r2 && target && conf1 && view2 && conf2, MODE_ERR_BAD_ARG, "mode_zbuffer: null pointer");
  MODE_REQUIRE(workspace, MODE_ERR_WORKSPACE, "mode_zbuffer: workspace required");
  hipStream_t st = mode::as_stream(stream);
  int frc = mode::fill_words(workspace, 0xffffffffu, 2 * (size_t)n, st, "mode_zbuffer");  // all keys = +inf (a kernel, not hipMemsetAsync: common.h)
  if (frc != MODE_OK) return frc;
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(workspace);
  hipLaunchKernelGGL(zbuffer_scatter_kernel, dim3(grid_for(n)), dim3(NT), 0, st, r2, target, keys, n);
  hipLaunchKernelGGL(view_trans_resolve_kernel, dim3(grid_for(n)), dim3(NT), 0, st, keys, conf1, view2, conf2, n);
  return mode::check_launch("mode_zbuffer");
}
