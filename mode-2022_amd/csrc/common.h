// Shared host-side helpers for libmode_hip.so (gfx950 only; no CUDA path, no dual-platform macros).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <utility>

#include "mode_hip.h"

namespace mode {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return MODE_OK;
}

inline hipStream_t as_stream(mode_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Dynamic LDS above 64 KiB must be opted into per kernel (gfx950 has 160 KiB per CU).
template <typename K>
inline int allow_lds(K kernel, size_t bytes, const char* what) {
  if (bytes <= 64 * 1024) return MODE_OK;
  if (bytes > 160 * 1024) {
    set_error("%s: needs %zu B of LDS (> 160 KiB)", what, bytes);
    return MODE_ERR_UNSUPPORTED;
  }
  // once per (kernel, size): keeps the call out of the steady state (and out of hipGraph captures)
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> granted;
  const void* fn = reinterpret_cast<const void*>(kernel);
  int device = 0;
  (void)hipGetDevice(&device);
  const std::pair<const void*, int> key(fn, device);
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = granted.find(key);
    if (it != granted.end() && it->second >= bytes) return MODE_OK;
  }
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> g(mu);
    granted[key] = bytes;
  }
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(%zu B LDS): %s", what, bytes, hipGetErrorString(e));
    return (int)e;
  }
  return MODE_OK;
}

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace mode

#define MODE_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      ::mode::set_error(__VA_ARGS__);  \
      return (code);                   \
    }                                  \
  } while (0)

// MI355X: 256 CUs in 8 XCDs; block b is dispatched to XCD b % 8 (performance hint only).
constexpr int kNumCU = 256;
constexpr int kNumXCD = 8;

#ifdef __HIPCC__
// XCD-aware bijective remap of a block id in [0, n): consecutive ids are dispatched round-robin over the 8 XCDs, each with its
// own L2; this gives every XCD a contiguous range of work items, so that neighbouring tiles (shared halo rows, overlapping
// gather footprints) meet in one L2.  Measured on the ring weight-gradient kernel: 1.20 GB -> 0.42 GB fetched per launch.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n / kNumXCD, r = n % kNumXCD;
  const int xcd = bid % kNumXCD, k = bid / kNumXCD;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
#endif
