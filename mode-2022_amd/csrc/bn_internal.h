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

}  // namespace mode
