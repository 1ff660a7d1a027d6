// Internal links between sphere_conv.hip (general gather kernels) and sphere_conv_win.hip (windowed kernels).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace mode {

size_t sphere_bwd_weight_general_workspace(int B, int Ci, int Co, int Kh, int Kw, int Ho, int Wo, int groups, int nsel);

int sphere_bwd_weight_general(const float* gy, const float* pos, const float* x, float* gw, float* workspace, int B, int Ci, int H,
                              int W, int Co, int Kh, int Kw, int sH, int sW, int Ho, int Wo, int groups, const int* pixmap, int nsel,
                              hipStream_t st, const char* who);

}  // namespace mode
