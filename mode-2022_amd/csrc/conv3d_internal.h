// Internal (non-ABI) entry points shared between the conv3d translation units.
#pragma once
#include "common.h"

namespace mode {

// 3x3x3, pad 1, stride 1 convolution with a SINGLE output channel (the classifier heads, mode_disparity.py:76-80).
// The MFMA tile would be 31/32 padding for Co = 1; these use the vector ALU (forward) and an MFMA formulation with the 27
// taps as the GEMM-N dimension (weight gradient) or the GEMM-K dimension (input gradient).
int conv3d_co1_fwd(const float* x, const float* w, float* y, int B, int Ci, int D, int H, int W, hipStream_t st, const char* who);

// classif_head.hip: the same forward for Ci <= 32 and samples below 2^30 elements on the kernel of the fused classifier head (every
// request of the plane loop unconditional and in a fixed order, a position group's fragments re-requested as soon as its MFMAs are issued)
int conv3d_co1_fwd_small(const float* x, const float* w, float* y, int B, int Ci, int D, int H, int W, hipStream_t st, const char* who);

// gx (B,Ci,D,H,W) = input gradient for gy (B,1,D,H,W); overwrites gx.
int conv3d_co1_bwd_data(const float* gy, const float* w, float* gx, int B, int Ci, int D, int H, int W, hipStream_t st, const char* who);

size_t conv3d_co1_bwd_weight_workspace_floats(int B, int Ci, int D, int H, int W);
int conv3d_co1_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int D, int H, int W,
                          int accumulate, hipStream_t st, const char* who);

// conv3d_split.hip: stride-1 3x3x3 convolution on the bf16 matrix pipe with three-way split fp32 operands (fp32 accuracy).  Same
// argument meaning as conv3d_s1 in conv3d.hip (rows = output channels of the GEMM, K = its reduction channels, flip 0 forward /
// 1 backward-data); wpack >= conv3d_split_wpack_floats(K, rows) floats.
bool conv3d_split_supported(int K, int rows);
size_t conv3d_split_wpack_floats(int K, int rows);
// stats != nullptr (training, bn == nullptr): the kernel also leaves the BatchNorm batch statistics of y as conv3d_split_stat_partials()
// partial pairs per channel + the pivots of the shifted sums (layout: conv3d_split_kernel, EPI 3).
int conv3d_split_stat_partials();
int conv3d_s1_split(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int D, int H, int W, int flip,
                    hipStream_t st, const char* who, const mode_bn_epilogue* bn, float* stats = nullptr, const float* acc_in = nullptr,
                    const float* amax_x = nullptr, const float* amax_w = nullptr, float* amax_y = nullptr);
// (acc_in: y = conv(x) + acc_in -- a gradient that is already there added in the store, instead of a separate pass; not with bn / stats.
//  amax_x / amax_w: device floats max |x|, max |w| (both or neither) -> the two-piece fp16 arithmetic (three MFMAs per product) with
//  power-of-two scales; not with stats.  With bn (eval mode): amax_x and amax_y, NO amax_w -- the maximum of the folded weights is taken
//  where they are packed and kept in wpack; amax_y (MODE_BN_ABSMAX_FLOATS floats) receives the maximum of the stored output)

// out[0] (device) = the largest magnitude in x[0..n): the scale source of the fp16 arithmetic (order-independent, graph-capturable)
int abs_max(const float* x, long long n, float* out, hipStream_t st, const char* who);
int abs_max_batch(const float* const* ptrs, const long long* sizes, int n, float* out, hipStream_t st, const char* who);

// conv3d_split_s2.hip: the stride-2 forward (= input gradient of the transposed convolution) on the same arithmetic; rows = output
// channels (33..64), K = reduction channels (multiple of 8); w is (rows, K, 27); wpack >= conv3d_s2_split_wpack_floats(K, rows).
bool conv3d_s2_split_supported(int K, int rows);
size_t conv3d_s2_split_wpack_floats(int K, int rows);
int conv3d_s2_split(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int D, int H, int W, hipStream_t st,
                    const char* who, const mode_bn_epilogue* bn = nullptr, float* amax_y = nullptr);
// bn: optional folded-BatchNorm epilogue (eval mode); amax_y (with bn): receives the maximum buffer of the stored output

// conv3d_split_deconv.hip: ConvTranspose3d k3 s2 p1 op1 (= the input gradient of the stride-2 convolution) on the same arithmetic;
// x (B, K, D, H, W), w (K, Co, 27) -> y (B, Co, 2D, 2H, 2W); K a multiple of 8, 2..64 output channels.
bool deconv3d_split_supported(int K, int Co);
bool deconv3d_split_bn_supported(int K, int Co);  // with the folded-BatchNorm epilogue: whole 32-channel output tiles
size_t deconv3d_split_wpack_floats(int K, int Co);
int deconv3d_split(const float* x, const float* w, float* y, float* wpack, int B, int K, int Co, int D, int H, int W, hipStream_t st,
                   const char* who, const mode_bn_epilogue* bn = nullptr, const float* acc_in = nullptr, float* amax_y = nullptr);
// bn: optional folded-BatchNorm epilogue (eval mode; amax_y: receives the maximum buffer of the stored output); acc_in: y = deconv(x) +
// acc_in (whole 32-channel output tiles, not with bn)

// conv3d_split_wgrad.hip: split-K partials of the stride-1 weight gradient on the split-bf16 matrix path, written in the layout of
// conv3d.hip's weight-gradient kernels (part[s][o / 32][c / 32][tap][o % 32][c % 32]); the caller reduces them.
struct WgradSplitDims {
  int Ci, Co, D, H, W;
  int nWt, nHt, nDc, ring_dc, units;  // units = B * nHt * nWt * nDc work units of ring_dc depths x 2 rows x 32 voxels
  int S, MTo, MTc;                    // workgroups along the split-K axis, 32-channel blocks of gy / x
};
int conv3d_bww_split_launch(const float* gy, const float* x, float* part, const WgradSplitDims& d, hipStream_t st, const char* who,
                            const float* amax_g = nullptr, const float* amax_x = nullptr);  // device floats max |gy|, max |x| -> fp16 arithmetic

// conv3d_split_wgrad_s2.hip: the same for the stride-2 convolution (x at D x H x W, gy at Do x Ho x Wo = half of it; even D, H and
// W a multiple of 8); a workgroup owns BOTH 32-row blocks of a 64-channel gy block (grid y = cdiv(MTo, 2)).
struct WgradS2SplitDims {
  int Ci, Co, D, H, W, Do, Ho, Wo;
  int nWt, nDc, ring_dc, units;  // units = B * Ho * nWt * nDc work units of ring_dc output depths x 1 output row x 16 output voxels
  int S, MTo, MTc;
};
int conv3d_bww_s2_split_launch(const float* gy, const float* x, float* part, const WgradS2SplitDims& d, hipStream_t st, const char* who);

// conv2d_split.hip: the regular 3x3 Conv2d layers (stride 1, dilation 1 / 2) on the split-bf16 matrix path; arguments as conv2d.hip's
// run() (rows = output channels of the GEMM, K = its reduction channels, flip 0 forward / 1 input gradient).
bool conv2d_split_supported(int K, int rows, int dilation);
size_t conv2d_split_wpack_floats(int K, int rows);
int conv2d_split_run(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int H, int W, int dilation, int flip,
                     hipStream_t st, const char* who, const mode_bn_epilogue* bn, const float* acc_in = nullptr,
                     const float* amax_x = nullptr, const float* amax_w = nullptr, float* amax_y = nullptr);
// acc_in: y = conv(x) + acc_in.  amax_x / amax_w: the two-piece fp16 arithmetic.  With bn (eval): amax_x and amax_y (the stored output's
// maximum, out), no amax_w -- the folded weights' maximum is taken with the pack and kept in wpack.

// conv2d_split_wgrad.hip: split-K partials of the 3x3 Conv2d weight gradient on the split-bf16 path, in the layout of conv2d_wgrad.hip
// (part[s][o / 32][c / 32][tap][o % 32][c % 32]); the caller reduces them.
struct Wgrad2SplitDims {
  int Ci, Co, H, W;
  int nWt, nGroups, run_groups, nRun, units;  // units = B * nWt * nRun runs of run_groups 4-row groups x 32 columns
  int S, MTo, MTc;
};
int conv2d_bww_split_launch(const float* gy, const float* x, float* part, const Wgrad2SplitDims& d, int dilation, hipStream_t st,
                            const char* who, const float* amax_g = nullptr, const float* amax_x = nullptr);

}  // namespace mode
