// Internal (non-ABI) entry points of bn_act.hip used by other translation units.
#pragma once
#include "common.h"

namespace mode {

// Training-mode BatchNorm WITHOUT its apply pass: the statistics pass over y (B, C, S) and the per-channel finalisation of
// mode_bn_train_fwd -- batch mean / invstd, the float32 affine coefficients scale = gamma * invstd, shift = beta - mean * scale (C floats
// each), the running-statistics update and the batch count -- for a consumer that normalises while it stages its operand.
// workspace >= mode_bn_workspace_bytes(C).
int bn_train_coefficients(const float* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          long long* num_batches_tracked, float momentum, float eps, float* save_mean, float* save_invstd,
                          float* save_scale, float* save_shift, float* workspace, int B, int C, long long S, hipStream_t st,
                          const char* who);

// ---- the largest finite magnitude of a tensor, as its fp16-arithmetic consumers read it (mode_abs_max, the out_absmax / gy_absmax
// arguments of the `_amax` entries) ---------------------------------------------------------------------------------------------------------------
// `amax` points at MODE_BN_ABSMAX_FLOATS words, all zero when the producing pass starts; the value is the maximum over word 0 and the
// ABSMAX_SLOTS words at 16 * (1 + s) (bit patterns of non-negative floats: an unsigned maximum).  One address for the whole launch does
// not work: tens of thousands of waves end within microseconds of each other, and their requests to ONE word -- the atomics, and just as
// much the loads that guard them -- queue up behind each other at the memory side (device scope: not served by the per-XCD L2s): an atomic
// per wave was 3 x a BatchNorm pass's time, a guarded one still +60 % on the 403 MB layers.  So: one request per BLOCK (its waves meet
// in LDS), spread over 128 words of 128 different cache lines.  Nobody folds them: a kernel of its own behind every pass was 5 us x 100
// per step; every wave of a consumer reads the 129 words itself (two loads per lane and a wave reduction, once per kernel).
constexpr int ABSMAX_SLOTS = 128;
static_assert(MODE_BN_ABSMAX_FLOATS == 16 * (1 + ABSMAX_SLOTS), "include/mode_hip.h and bn_internal.h disagree about the maximum's buffer");

// bit pattern of |f| when f is finite, else 0 (the scale of an fp16 consumer fits the finite data; NaN / Inf stay where they are)
__device__ __forceinline__ unsigned absmax_mag(float f) {
  const unsigned u = __builtin_bit_cast(unsigned, f) & 0x7fffffffu;
  return u < 0x7f800000u ? u : 0u;
}
// every thread of the block calls this once, behind its last store; sh: blockDim.x / 64 words of LDS nobody reads any more
__device__ __forceinline__ void absmax_block_commit(unsigned mx, unsigned* amax, unsigned* sh) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mx = max(mx, sh[w]);
    unsigned* slot = amax + 16 * (1 + (int)((blockIdx.y * gridDim.x + blockIdx.x) % ABSMAX_SLOTS));
    // (the value only grows: a block whose maximum is not above what is already there has nothing to add; a stale read costs one spare atomic)
    if (mx > __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMax(slot, mx);
  }
}
// what a consumer does with the pointer: every lane of a FULL wave calls it (kernel prologue) and gets the maximum
__device__ __forceinline__ float absmax_load(const float* amax) {
  const unsigned* u = reinterpret_cast<const unsigned*>(amax);
  const int lane = threadIdx.x & 63;
  unsigned mx = max(max(u[16 * (1 + lane)], u[16 * (1 + lane + 64)]), u[0]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, 64));
  return __builtin_bit_cast(float, mx);
}
// zero the buffer in front of the pass (where no kernel of the pass does it on the way)
int absmax_begin(float* amax, hipStream_t st, const char* who);

}  // namespace mode
