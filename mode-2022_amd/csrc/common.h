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

// Fill `nwords` 32-bit words at `dst` (4-byte aligned) with `pattern` by a KERNEL on `st`.  The library never calls hipMemsetAsync:
// captured into a hipGraph on this ROCm stack a memset node takes effect on the first launch of the graph only -- from the second
// replay on the buffer keeps whatever it held (tools/experiments/graph_memset_probe.py; round 4: 3/4 of the stride-2 1x1 input
// gradient came back as garbage from every replayed training step but the first).  A kernel node replays like any other.
int fill_words(void* dst, unsigned pattern, size_t nwords, hipStream_t st, const char* what);
// false after mode_weight_pack_reuse(1) on this thread: the caller's wpack workspace already holds the packed weights
bool pack_needed();
inline int zero_floats(float* dst, size_t n, hipStream_t st, const char* what) { return fill_words(dst, 0u, n, st, what); }

}  // namespace mode

#define MODE_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      ::mode::set_error(__VA_ARGS__);  \
      return (code);                   \
    }                                  \
  } while (0)

namespace mode {
inline int check_bn(const mode_bn_epilogue* bn, const char* who) {
  MODE_REQUIRE(bn && bn->gamma && bn->beta && bn->mean && bn->var, MODE_ERR_BAD_ARG, "%s: null BatchNorm epilogue / vector", who);
  return MODE_OK;
}
}  // namespace mode

// MI355X: 256 CUs in 8 XCDs; block b is dispatched to XCD b % 8 (performance hint only).
constexpr int kNumCU = 256;
constexpr int kNumXCD = 8;

// Device-side view of mode_bn_epilogue (include/mode_hip.h): out = relu?(acc + shift[o] [+ add]); the BatchNorm scale is folded
// into the packed weights by the packing kernel, which also writes `shift` (fold_scale / fold_shift below).  shift == nullptr:
// plain store.
struct Epi {
  const float* shift;
  const float* add;
  int relu;
  unsigned* amax;  // eval mode: where the kernel leaves the largest finite magnitude it stored (bn_internal.h's buffer, zeroed), or null
};

inline Epi make_epi(const mode_bn_epilogue* e, const float* shift) {
  Epi r;
  r.shift = e ? shift : nullptr;
  r.add = e ? e->add : nullptr;
  r.relu = e ? e->relu : 0;
  r.amax = nullptr;
  return r;
}

#ifdef __HIPCC__
// Eval-mode BatchNorm folded into the convolution in front of it (torch semantics: y = (x - mean) / sqrt(var + eps) * gamma + beta):
// the packing kernels scale output channel o of the weights by fold_scale and write fold_shift(o) next to the packed weights.
__device__ __forceinline__ float fold_scale(const mode_bn_epilogue& e, int o) { return e.gamma[o] / sqrtf(e.var[o] + e.eps); }
__device__ __forceinline__ float fold_shift(const mode_bn_epilogue& e, int o) { return e.beta[o] - e.mean[o] * fold_scale(e, o); }
// Buffer-addressed loads (a 128-bit descriptor in scalar registers + a 32-bit lane offset + a scalar offset): no 64-bit vector
// address arithmetic per request, and a lane offset at or beyond `bytes` reads as ZERO -- the zero padding of a haloed tile costs one
// select of the OFFSET per position instead of one select per loaded value.  The descriptor must be built from wave-uniform values.
// kBufOOB is the offset of a lane that must read zero (bytes < 2^31 is the caller's contract).
constexpr unsigned kBufOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned lane_off_bytes, unsigned scalar_off_bytes) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off_bytes, scalar_off_bytes, 0));
}
// ReLU as torch computes it: NaN stays NaN (fmaxf(NaN, 0) would return 0 and hide a diverged activation).
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float apply_epi(const Epi& e, float v, int o, long long idx) {
  v += e.shift[o];
  if (e.add) v += e.add[idx];
  return e.relu ? relu_nan(v) : v;
}
// Register-lean form for the MFMA epilogues (D[i = o][j]: accumulator register q of lane l holds output channel
// o = tile * 32 + (q & 3) + 8 * (q >> 2) + 4 * (l >> 5)).  Fetching shift[o] per element made the compiler preload 16 values per
// M-tile next to the accumulators (conv2d 61 -> 80 VGPRs, deconv3d 94 -> 148: one wave per SIMD less).  Instead every lane loads
// ONE shift per M-tile -- lane l that of channel tile * 32 + (l & 31) -- and an element takes its value from the lane that holds it
// with a cross-lane read (ds_bpermute: no memory access, nothing to keep live).
__device__ __forceinline__ float epi_tile_shift(const Epi& e, int tile_channel0, int nchan) {
  const int c = tile_channel0 + (int)(threadIdx.x & 31);
  return e.shift[c < nchan ? c : nchan - 1];
}
__device__ __forceinline__ float epi_apply_q(const Epi& e, float v, float tile_shift, int q, long long idx) {
  const int src = (q & 3) + 8 * (q >> 2) + 4 * (int)((threadIdx.x & 63) >> 5);  // lane (= channel within the tile) holding this element's shift
  v += __shfl(tile_shift, src, 64);
  if (e.add) v += e.add[idx];
  return e.relu ? relu_nan(v) : v;
}

// Split-K reduction out[i] (+)= sum_s part[s * n + i], one wave per output element: lane l takes the slices l, l + 64, ... in
// ascending order, then a fixed butterfly over the lanes -- deterministic, and S loads deep instead of S loads long (a thread
// per element walked its S slices one dependent load after the other: 40 us for a 4 704-element gradient with 512 slices).
// grid = ceil(n / 4) blocks of 256 threads.
static __global__ __launch_bounds__(256) void reduce_slices_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int S,
                                                                   int accumulate) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  float v = 0.f;
  for (int s = lane; s < S; s += 64) v += part[(long long)s * n + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if (lane == 0) out[i] = accumulate ? out[i] + v : v;
}

// XCD-aware bijective remap of a block id in [0, n): consecutive ids are dispatched round-robin over the 8 XCDs, each with its
// own L2; this gives every XCD a contiguous range of work items, so that neighbouring tiles (shared halo rows, overlapping
// gather footprints) meet in one L2.  Measured on the ring weight-gradient kernel: 1.20 GB -> 0.42 GB fetched per launch.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n / kNumXCD, r = n % kNumXCD;
  const int xcd = bid % kNumXCD, k = bid / kNumXCD;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}
#endif
