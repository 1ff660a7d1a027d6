// Times libmode_hip.so's stride-1 3x3x3 convolution on the split-bf16 matrix path against its fp32 MFMA path and measures the
// error of both against an exact (double) evaluation at sampled outputs.  No Python, seconds per run:
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/experiments/conv3d_split_bench.cpp -Lmode-2022_amd/mode_hip -lmode_hip \
//         -Wl,-rpath,$PWD/mode-2022_amd/mode_hip -o /tmp/conv3d_split_bench && /tmp/conv3d_split_bench [Ci Co D H W B]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "mode_hip.h"

extern "C" int mode_conv3d_fwd_split(const float*, const float*, const mode_bn_epilogue*, float*, float*, int, int, int, int, int, int,
                                     mode_stream_t);
extern "C" int mode_conv3d_bwd_weight_split(const float*, const float*, float*, float*, int, int, int, int, int, int, int, mode_stream_t);
extern "C" int mode_conv3d_bwd_data_split(const float*, const float*, float*, float*, int, int, int, int, int, int, mode_stream_t);

#define CK(e)                                                                 \
  do {                                                                        \
    hipError_t _e = (e);                                                      \
    if (_e != hipSuccess) {                                                   \
      printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e));        \
      return 1;                                                               \
    }                                                                         \
  } while (0)

int main(int argc, char** argv) {
  int Ci = 32, Co = 32, D = 48, H = 256, W = 128, B = 2;
  if (argc > 6) {
    Ci = atoi(argv[1]); Co = atoi(argv[2]); D = atoi(argv[3]); H = atoi(argv[4]); W = atoi(argv[5]); B = atoi(argv[6]);
  }
  const long long DHW = (long long)D * H * W;
  const size_t nx = (size_t)B * Ci * DHW, ny = (size_t)B * Co * DHW, nw = (size_t)Co * Ci * 27;
  std::vector<float> hx(nx), hw(nw);
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& v : hx) v = nd(rng);
  for (auto& v : hw) v = nd(rng) * 0.05f;
  float *x, *w, *y, *y32, *wpack;
  CK(hipMalloc(&x, nx * 4));
  CK(hipMalloc(&w, nw * 4));
  CK(hipMalloc(&y, ny * 4));
  CK(hipMalloc(&y32, ny * 4));
  CK(hipMalloc(&wpack, mode_conv3d_wpack_bytes(Ci, Co)));
  CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
  CK(hipMemset(y, 0xff, ny * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double flop = 2.0 * 27 * Ci * Co * (double)B * DHW;
  for (int which = 0; which < 2; ++which) {
    auto run = [&]() {
      return which == 0 ? mode_conv3d_fwd(x, w, y32, wpack, B, Ci, D, H, W, Co, 1, nullptr)
                        : mode_conv3d_fwd_split(x, w, nullptr, y, wpack, B, Ci, D, H, W, Co, nullptr);
    };
    for (int i = 0; i < 3; ++i)
      if (run() != MODE_OK) {
        printf("%s failed: %s\n", which ? "split" : "fp32", mode_last_error());
        return 1;
      }
    CK(hipDeviceSynchronize());
    const int n = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) run();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("conv3d %d->%d @%dx%dx%d B=%d  %-22s %.3f ms per launch (incl. weight packing) = %.1f TFLOP/s fp32-equivalent\n", Ci, Co, D, H,
           W, B, which ? "split bf16 x 6:" : "fp32 MFMA:", ms / n, flop / (ms / n * 1e-3) / 1e12);
  }
  std::vector<float> hy(ny), hy32(ny);
  CK(hipMemcpy(hy.data(), y, ny * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hy32.data(), y32, ny * 4, hipMemcpyDeviceToHost));
  // exact evaluation at sampled outputs (corners and edges included)
  std::uniform_int_distribution<long long> pick(0, (long long)ny - 1);
  double m_s = 0, m_f = 0, s_s = 0, s_f = 0, ymax = 0;
  const int NS = 20000;
  for (int s = 0; s < NS; ++s) {
    long long idx = s < 8 ? (s & 1 ? (long long)ny - 1 - s : s) : pick(rng);
    long long r = idx;
    const int wq = (int)(r % W); r /= W;
    const int hq = (int)(r % H); r /= H;
    const int dq = (int)(r % D); r /= D;
    const int o = (int)(r % Co);
    const int b = (int)(r / Co);
    double acc = 0;
    for (int c = 0; c < Ci; ++c)
      for (int t = 0; t < 27; ++t) {
        const int dd = dq + t / 9 - 1, hh = hq + (t / 3) % 3 - 1, ww = wq + t % 3 - 1;
        if (dd < 0 || dd >= D || hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
        acc += (double)hw[((size_t)o * Ci + c) * 27 + t] * (double)hx[((size_t)b * Ci + c) * DHW + (size_t)dd * H * W + (size_t)hh * W + ww];
      }
    const double es = std::fabs(hy[idx] - acc), ef = std::fabs(hy32[idx] - acc);
    m_s = std::max(m_s, es); m_f = std::max(m_f, ef);
    s_s += es * es; s_f += ef * ef;
    ymax = std::max(ymax, std::fabs(acc));
  }
  printf("error against an exact evaluation at %d sampled outputs (|y| up to %.2f):\n  split bf16 x 6 : max %.3e rms %.3e\n"
         "  fp32 MFMA      : max %.3e rms %.3e\n", NS, ymax, m_s, std::sqrt(s_s / NS), m_f, std::sqrt(s_f / NS));
  // ---- weight gradient: gW = sum gy * x over all voxels (x doubles as the operand, y32 as gy)
  float *gw, *gw32, *ws;
  CK(hipMalloc(&gw, nw * 4));
  CK(hipMalloc(&gw32, nw * 4));
  CK(hipMalloc(&ws, mode_conv3d_bwd_weight_workspace_bytes(B, Ci, D, H, W, Co, 1)));
  for (int which = 0; which < 2; ++which) {
    auto run = [&]() {
      return which == 0 ? mode_conv3d_bwd_weight(y32, x, gw32, ws, B, Ci, D, H, W, Co, 1, 0, nullptr)
                        : mode_conv3d_bwd_weight_split(y32, x, gw, ws, B, Ci, D, H, W, Co, 0, nullptr);
    };
    for (int i = 0; i < 3; ++i)
      if (run() != MODE_OK) {
        printf("weight gradient %s failed: %s\n", which ? "split" : "fp32", mode_last_error());
        return 1;
      }
    CK(hipDeviceSynchronize());
    const int n = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) run();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("conv3d weight gradient %d->%d  %-22s %.3f ms per launch (incl. the split-K reduction) = %.1f TFLOP/s fp32-equivalent\n", Ci, Co,
           which ? "split bf16 x 6:" : "fp32 MFMA:", ms / n, flop / (ms / n * 1e-3) / 1e12);
  }
  std::vector<float> hgw(nw), hgw32(nw);
  CK(hipMemcpy(hgw.data(), gw, nw * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hgw32.data(), gw32, nw * 4, hipMemcpyDeviceToHost));
  std::uniform_int_distribution<long long> pickw(0, (long long)nw - 1);
  double wm_s = 0, wm_f = 0, wmax = 0;
  for (int sidx = 0; sidx < 64; ++sidx) {
    const long long idx = sidx < 27 ? sidx : pickw(rng);  // all 27 taps of (o, c) = (0, 0), then random entries
    const int t = (int)(idx % 27), c = (int)((idx / 27) % Ci), o = (int)(idx / 27 / Ci);
    double acc = 0;
    for (int b = 0; b < B; ++b)
      for (int dq = 0; dq < D; ++dq)
        for (int hq = 0; hq < H; ++hq) {
          const int dd = dq + t / 9 - 1, hh = hq + (t / 3) % 3 - 1;
          if (dd < 0 || dd >= D || hh < 0 || hh >= H) continue;
          const float* gp = &hy32[((size_t)b * Co + o) * DHW + (size_t)dq * H * W + (size_t)hq * W];
          const float* xp = &hx[((size_t)b * Ci + c) * DHW + (size_t)dd * H * W + (size_t)hh * W];
          for (int wq = 0; wq < W; ++wq) {
            const int ww = wq + t % 3 - 1;
            if (ww >= 0 && ww < W) acc += (double)gp[wq] * (double)xp[ww];
          }
        }
    wm_s = std::max(wm_s, std::fabs(hgw[idx] - acc));
    wm_f = std::max(wm_f, std::fabs(hgw32[idx] - acc));
    wmax = std::max(wmax, std::fabs(acc));
  }
  printf("weight gradient against an exact evaluation at 64 entries (|gW| up to %.1f): split max %.3e, fp32 MFMA max %.3e\n", wmax, wm_s, wm_f);
  return (m_s < 5e-5 && m_f < 5e-5 && wm_s < 1e-4 * std::max(1.0, wmax)) ? 0 : 2;
}
