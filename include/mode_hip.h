/*
 * mode_hip.h -- C-ABI of libmode_hip.so: the MI355X (gfx950) kernels of MODE's disparity-stage hot path.
 *
 * Drop-in boundary.  The reference binds its native code through a two-function pybind11 module
 * (models/basic/spherical_conv/src/sphere_conv_cuda.cpp:339-345) that takes ATen tensors.  This
 * library is what a reference-side binding (ctypes / pybind / cgo) binds instead: plain device
 * pointers, sizes and a HIP stream -- no torch types.  Each entry point cites the reference
 * interface it replaces; paths are relative to the upstream repository root.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer to a contiguous fp32 buffer allocated by the caller
 *     (the reference requires contiguous tensors too: sphere_conv_cuda.cpp:48, 138-140);
 *   - the library never allocates or keeps device memory; scratch is passed in as `workspace`;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream) and
 *     the call returns without synchronising (reference: at::cuda::getCurrentCUDAStream(),
 *     sphere_conv_cuda_kernel.cu:280, 374);
 *   - the return value is MODE_OK (0) or a negative MODE_ERR_* code for rejected arguments /
 *     a positive hipError_t for a launch failure; mode_last_error() returns a description.  The
 *     reference throws c10::Error on bad shapes (sphere_conv_cuda.cpp:43-125) but only printf()s
 *     launch errors (sphere_conv_cuda_kernel.cu:286-289); the Python binding raises RuntimeError
 *     for every non-zero code;
 *   - re-entrant across devices and threads: no global mutable state except the thread-local
 *     error string (nn.DataParallel calls forward from one thread per GPU).
 */
#ifndef MODE_HIP_H_
#define MODE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mode_stream_t; /* hipStream_t */

enum {
  MODE_OK = 0,
  MODE_ERR_BAD_ARG = -1,      /* NULL pointer, non-positive size, inconsistent shapes */
  MODE_ERR_UNSUPPORTED = -2,  /* valid request outside what the kernels implement */
  MODE_ERR_WORKSPACE = -3     /* workspace missing or too small */
};

/* Eval-mode epilogue of the *_bn entry points: the BatchNorm that follows a convolution in every convbn / convbn_3d /
 * sphereConvbn block (models/submodule.py:15-22, 61-74), in inference mode (running statistics), folded into the convolution:
 *     out = relu?( conv(x, w)[o] * gamma[o] / sqrt(var[o] + eps) + beta[o] - mean[o] * gamma[o] / sqrt(var[o] + eps) [+ add] )
 * The scale goes into the packed weights, the shift / residual add / ReLU into the store of the convolution kernel: the layer
 * is one launch and the un-normalised convolution result never reaches HBM.  All pointers are device pointers; `add` (same
 * shape as the output) may be NULL.  The struct itself lives in host memory and is read during the call. */
typedef struct mode_bn_epilogue {
  const float* gamma; /* (Co) BatchNorm weight */
  const float* beta;  /* (Co) BatchNorm bias */
  const float* mean;  /* (Co) running_mean */
  const float* var;   /* (Co) running_var */
  float eps;
  const float* add; /* residual added before the ReLU, or NULL */
  int relu;
} mode_bn_epilogue;

/* Version / diagnostics. */
#define MODE_HIP_ABI_VERSION 31 /* bumped whenever a signature below changes */
int mode_hip_abi_version(void);
const char* mode_last_error(void);

/* Test utility, not part of the operator seam (no reference counterpart): fills the LDS of every CU and the vector / accumulator
 * register files with `pattern`, on `stream`.  A kernel whose output depends on LDS words or registers it never wrote gives
 * pattern-dependent results; tests/test_gpu_repeat.py runs every operator of the training step behind it. */
int mode_debug_poison(unsigned pattern, mode_stream_t stream);

/* Inference with fixed weights (no reference counterpart: the reference re-reads its weights on every call, test_disparity.py:136).
 * Every forward entry that takes a `wpack` workspace first launches a small kernel that repacks the weights (with the folded
 * BatchNorm scale) into it -- 113 launches, 0.65 ms of a 10.3 ms eval forward at one pair.  After mode_weight_pack_reuse(1) those
 * entries SKIP the repacking on the calling thread and use `wpack` as it is: the caller guarantees that it still holds what the same
 * entry wrote for the same weights, BatchNorm parameters and shapes (mode_hip.functional keeps such workspaces per layer and keys
 * them on the tensors' versions).  mode_weight_pack_reuse(0) restores the default.  Returns the previous setting. */
int mode_weight_pack_reuse(int on);

/* ---------------------------------------------------------------------------------------------
 * Spherical convolution.
 *
 * Replaces sphere_conv_forward_cuda (sphere_conv_cuda.cpp:129-210 = per-sample im2col kernel
 * sphere_conv_cuda_kernel.cu:195-262 + addmm_) with ONE fused launch: the bilinear gather writes
 * the column tile to LDS and the contraction over Ci*Kh*Kw runs on fp32 MFMA; no HBM column buffer.
 *
 *   x   (B, Ci, H, W)            pos (1, 2*Kh*Kw, H, W): channel 2k = row coordinate, 2k+1 = column
 *   w   (Co, Ci/groups, Kh, Kw)  coordinate of tap k, sampled at (h_out*sH, w_out*sW)
 *   y   (B, Co, Ho, Wo)          written (not accumulated)
 *
 * `wpack` is scratch for the MFMA-fragment-ordered copy of the weights, >= mode_sphere_conv_wpack_bytes().
 * padding / dilation only enter the output-size formula (sphere_conv.py:112-113), so the caller
 * passes Ho, Wo.
 */
size_t mode_sphere_conv_wpack_bytes(int Ci, int Co, int Kh, int Kw, int groups);

int mode_sphere_conv_fwd(const float* x, const float* pos, const float* w, float* y, float* wpack,
                         int B, int Ci, int H, int W, int Co, int Kh, int Kw, int sH, int sW,
                         int Ho, int Wo, int groups, mode_stream_t stream);

/* sphereConvbn (models/submodule.py:61-74) -- and, through an integer table, any other convbn -- in eval mode as one launch:
 * mode_sphere_conv_fwd with the folded-BatchNorm epilogue (mode_bn_epilogue above). */
int mode_sphere_conv_fwd_bn(const float* x, const float* pos, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack,
                            int B, int Ci, int H, int W, int Co, int Kh, int Kw, int sH, int sW, int Ho, int Wo, int groups,
                            mode_stream_t stream);

/* Replaces the grad_input half of sphere_conv_backward_cuda (sphere_conv_cuda.cpp:275-294:
 * addmm_(W^T, gO) + col2im kernel sphere_conv_cuda_kernel.cu:293-356).  ACCUMULATES into gx, which
 * the caller zero-fills first, exactly like the reference (sphere_conv.py:62). */
int mode_sphere_conv_bwd_data(const float* gy, const float* pos, const float* w, float* gx, float* wpack,
                              int B, int Ci, int H, int W, int Co, int Kh, int Kw, int sH, int sW,
                              int Ho, int Wo, int groups, mode_stream_t stream);

/* The same gradient in gather form: deterministic, no atomics, ~7x faster.  It needs the ADJOINT of the sampling table,
 * which depends on the table only (a constant of the module, sphere_conv.py:150) and is built once on the HOST:
 *   mode_sphere_adjoint_build(pos_host, ...) fills rowptr_host[Kh*Kw*H*W + 1] and entries_host[2 * n] (n <=
 *   mode_sphere_adjoint_max_entries()) -- for tap k and input pixel q, entries rowptr[k*H*W+q] .. rowptr[k*H*W+q+1] are
 *   (output pixel p, float bits of the bilinear weight) pairs.  The caller uploads both arrays and passes the device
 *   copies to mode_sphere_conv_bwd_data_adj, which adds to gx like the scatter form (accumulate = 1) or overwrites it
 *   (accumulate = 0: no zero-fill and no read of gx needed). */
size_t mode_sphere_adjoint_max_entries(int Kh, int Kw, int Ho, int Wo);

int mode_sphere_adjoint_build(const float* pos_host, int H, int W, int Kh, int Kw, int sH, int sW, int Ho, int Wo,
                              int32_t* rowptr_host, int32_t* entries_host, int64_t* n_entries);

int mode_sphere_conv_bwd_data_adj(const float* gy, const float* w, float* gx, float* wpack, const int32_t* adj_rowptr,
                                  const int32_t* adj_entries, int B, int Ci, int H, int W, int Co, int Kh, int Kw,
                                  int Ho, int Wo, int groups, int accumulate, mode_stream_t stream);

/* The gather form restricted to a LIST of 64-pixel tiles of gx (tile t = linear pixels 64 t .. 64 t + 63 of the H x W image as it is
 * stored; n_list tiles per sample; 3x3 kernels): everything outside the list is left untouched. */
int mode_sphere_conv_bwd_data_adj_list(const float* gy, const float* w, float* gx, float* wpack, const int32_t* adj_rowptr,
                                       const int32_t* adj_entries, int B, int Ci, int H, int W, int Co, int Kh, int Kw,
                                       int Ho, int Wo, int groups, int accumulate, const int32_t* tile_list, int n_list,
                                       mode_stream_t stream);

/* Windowed input gradient on the split-bf16 matrix path (csrc/sphere_conv_win.hip; replaces the col2im scatter of
 * sphere_conv_cuda_kernel.cu:293-356 + the GEMM of sphere_conv_cuda.cpp:275-315 for stride 1, 3x3 taps, output grid = input grid).
 * Host planning, once per table: mode_sphere_adjplan_build(pos_host, H, W, Kh, Kw, good_tiles[4 n], bad_tiles[2 n], counts[2],
 *   rec_off_host[n * 9 * 256 * 4], rec_w_host[same], rec_off2_host[n * 9 * 256 * 2], rec_w2_host[same]) with n =
 *   mode_sphere_plan_max_tiles(H, W): for every 64 x 4 tile of INPUT pixels whose adjoint lists all have <= 6 entries inside one 81-row
 *   x 8-column window of gy it writes (h0, w0, rbase, cbase | six << 16) and, per (tile, tap, pixel), 4 (window offset, weight) slots
 *   + 2 more in the second pair of arrays (six = 1 when any list of the tile uses them); the other tiles go to bad_tiles as (h0, w0).
 *   counts = (good, bad).
 * mode_sphere_conv_bwd_data_win_split WRITES gx on the good tiles (fp32 operands split exactly into 3 bf16 pieces, 6 bf16 MFMAs per
 *   product, fp32 accumulation); the caller runs mode_sphere_conv_bwd_data_adj_list (accumulate = 0) on the bad ones.  `transposed`: gy
 *   and gx are plane-transposed (B, C, W, H).  Needs mode_sphere_conv_bwd_data_win_supported(Ci, Co, groups) == 1 (output channels per
 *   group a multiple of 16); `wpack` >= mode_sphere_conv_bwd_data_win_wpack_bytes(). */
int mode_sphere_adjplan_build(const float* pos_host, int H, int W, int Kh, int Kw, int32_t* good_tiles, int32_t* bad_tiles,
                              int32_t* counts, int32_t* rec_off_host, float* rec_w_host, int32_t* rec_off2_host, float* rec_w2_host);

size_t mode_sphere_conv_bwd_data_win_wpack_bytes(int Ci, int Co, int Kh, int Kw, int groups);

int mode_sphere_conv_bwd_data_win_supported(int Ci, int Co, int groups);

int mode_sphere_conv_bwd_data_win_split(const float* gy, const float* w, float* gx, float* wpack, const int32_t* tiles, int n_tiles,
                                        const int32_t* rec_off, const float* rec_w, const int32_t* rec_off2, const float* rec_w2, int B,
                                        int Ci, int H, int W, int Co, int Kh, int Kw, int groups, int transposed, mode_stream_t stream);
/* The same on the two-piece fp16 arithmetic of mode_sphere_conv_fwd_win_split_f16 (a backward pass is a training step): amax_g / amax_w =
 * the maximum buffers (MODE_BN_ABSMAX_FLOATS floats; mode_abs_max, mode_bn_train_bwd_amax) of gy and of w. */
int mode_sphere_conv_bwd_data_win_split_f16(const float* gy, const float* w, const float* amax_g, const float* amax_w, float* gx, float* wpack,
                                            const int32_t* tiles, int n_tiles, const int32_t* rec_off, const float* rec_w,
                                            const int32_t* rec_off2, const float* rec_w2, int B, int Ci, int H, int W, int Co, int Kh,
                                            int Kw, int groups, int transposed, mode_stream_t stream);

/* Windowed forward (csrc/sphere_conv_win.hip): same result as mode_sphere_conv_fwd for stride 1 and 3x3 taps, ~2x faster on
 * tables whose samples are spatially compact (the gnomonic tables of the network).  The caller plans the table once on the HOST:
 *   mode_sphere_plan_build(pos_host, ...) fills tiles_host[4 * mode_sphere_plan_max_tiles(H, W)] with (h0, w0, rbase, cbase)
 *   per 64x4 tile of output pixels (the window class is packed into bits 16.. of the 4th word) and counts[4] = tiles per
 *   class (81-row window, 145-row window, whole-axis window, does-not-fit).  If counts[3] != 0 the table is not compact: use mode_sphere_conv_fwd.
 * Otherwise upload the tile list and pass it with counts[0..2]; `wpack` >= mode_sphere_conv_win_wpack_bytes(). */
size_t mode_sphere_plan_max_tiles(int H, int W);

int mode_sphere_plan_build(const float* pos_host, int H, int W, int Kh, int Kw, int32_t* tiles_host, int32_t* counts);

size_t mode_sphere_conv_win_wpack_bytes(int Ci, int Co, int Kh, int Kw, int groups);

int mode_sphere_conv_fwd_win(const float* x, const float* pos, const float* w, float* y, float* wpack, const int32_t* tiles,
                             int n_small, int n_mid, int n_wrap, int B, int Ci, int H, int W, int Co, int Kh, int Kw,
                             int groups, int transposed, mode_stream_t stream);

int mode_sphere_conv_fwd_win_bn(const float* x, const float* pos, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack,
                                const int32_t* tiles, int n_small, int n_mid, int n_wrap, int B, int Ci, int H, int W, int Co, int Kh,
                                int Kw, int groups, int transposed, mode_stream_t stream);
/* The same call with the small-window tiles on the split-bf16 matrix path (needs Ci / groups % 16 == 0, else it is the
 * call above); bn may be NULL (plain convolution). */
int mode_sphere_conv_fwd_win_split(const float* x, const float* pos, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack,
                                const int32_t* tiles, int n_small, int n_mid, int n_wrap, int B, int Ci, int H, int W, int Co, int Kh,
                                int Kw, int groups, int transposed, mode_stream_t stream);

/* The plain (no epilogue) call of a TRAINING step with the small-window tiles on the two-piece fp16 arithmetic of the stride-1 3-D layers
 * (two fp16 pieces per value, three MFMAs per product, a power-of-two scale per operand): amax_x / amax_w = device scalars
 * holding the largest finite magnitude of x and of w (mode_abs_max, or mode_bn_train_fwd_amax of the pass that wrote x).
 * Ci / groups % 16 != 0: the call above with bn = NULL. */
int mode_sphere_conv_fwd_win_split_f16(const float* x, const float* pos, const float* w, const float* amax_x, const float* amax_w, float* y,
                                       float* wpack, const int32_t* tiles, int n_small, int n_mid, int n_wrap, int B, int Ci, int H, int W,
                                       int Co, int Kh, int Kw, int groups, int transposed, mode_stream_t stream);

/* `transposed` != 0: x and y are stored plane-transposed, (B, C, W, H) contiguous, i.e. with the h axis contiguous
 * (mode_transpose_planes converts).  For the Cassini tables of the network h is the shift-invariant longitude axis, and in
 * this storage every global access of the windowed kernels is a full contiguous segment.  H, W, the table and the plan
 * always refer to the logical (untransposed) image. */
int mode_transpose_planes(const float* in, float* out, long long planes, int H, int W, mode_stream_t stream);

/* Windowed weight gradient: the 81-row-window tiles of the plan run on the LDS-window kernel; the other tiles on the polar
 * kernel (mode_sphere_plan_polar; pass n_rest_pixels = 0 then) or, if they could not be planned, their pixels
 * (mode_sphere_plan_rest_pixels: linear indices h*W + w, sorted; at most H*W; n_polar_items = 0 then) on the general kernels.  ADDS to gw like
 * mode_sphere_conv_bwd_weight; deterministic.  `workspace` >= mode_sphere_conv_bwd_weight_win_workspace_bytes().
 * gy_t / x_t (both or neither): plane-transposed copies of gy / x for the windowed kernel (see mode_transpose_planes); the
 * general kernels always read gy / x (which may be null when gy_t / x_t are given and n_rest_pixels = 0). */
int mode_sphere_plan_rest_pixels(const int32_t* tiles_host, const int32_t* counts, int H, int W, int32_t* pix_host,
                                 int32_t* n_pix);

/* Sampling records of the 81-row-window tiles (window offset + 4 corner weights per tap and pixel), which the windowed
 * weight-gradient kernel reads instead of re-deriving them from the table: rec_w_host[4 * n], rec_off_host[n],
 * n = mode_sphere_plan_records_count(counts[0]).  Built once per table on the host, uploaded by the caller. */
size_t mode_sphere_plan_records_count(int n_small);

int mode_sphere_plan_records(const float* pos_host, const int32_t* tiles_host, const int32_t* counts, int H, int W,
                             float* rec_w_host, int32_t* rec_off_host);

/* Column items of the tiles that are not of the 81-row class, for the polar weight-gradient kernel (one output column of 32 pixels
 * per item, nine per-tap windows of 34 rows x 2 columns): pitems_host[20 * n], rec_w_host[4 * 288 * n], rec_off_host[288 * n],
 * n <= mode_sphere_plan_polar_max_items(counts); *n_items = -1 when some column does not fit (use the pixel list instead). */
size_t mode_sphere_plan_polar_max_items(const int32_t* counts);

int mode_sphere_plan_polar(const float* pos_host, const int32_t* tiles_host, const int32_t* counts, int H, int W,
                           int32_t* pitems_host, float* rec_w_host, int32_t* rec_off_host, int32_t* n_items);

size_t mode_sphere_conv_bwd_weight_win_workspace_bytes(int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups,
                                                       int n_small, int n_rest_pixels, int n_polar_items);

/* mode_sphere_conv_bwd_weight_win_split: the same call with the compact-window tiles on the split-bf16 kernel (K = 16 pixels per
 * v_mfma_f32_32x32x16_bf16, fp32 operands split exactly into three bf16 pieces, fp32 accumulation). */
int mode_sphere_conv_bwd_weight_win_split(const float* gy, const float* pos, const float* x, float* gw, float* workspace,
                                          const int32_t* tiles, int n_small, int n_mid, int n_wrap, const float* rec_w,
                                          const int32_t* rec_off, const int32_t* rest_pixels, int n_rest_pixels,
                                          const int32_t* pitems, const float* prec_w, const int32_t* prec_off, int n_polar_items,
                                          int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups, const float* gy_t,
                                          const float* x_t, mode_stream_t stream);

/* ... and with those tiles on the two-piece fp16 arithmetic of mode_sphere_conv_fwd_win_split_f16: amax_g / amax_x = the maximum buffers
 * (MODE_BN_ABSMAX_FLOATS floats) of gy and of x.  The polar items keep three bf16 pieces. */
int mode_sphere_conv_bwd_weight_win_split_f16(const float* gy, const float* pos, const float* x, const float* amax_g, const float* amax_x,
                                              float* gw, float* workspace, const int32_t* tiles, int n_small, int n_mid, int n_wrap,
                                              const float* rec_w, const int32_t* rec_off, const int32_t* rest_pixels, int n_rest_pixels,
                                              const int32_t* pitems, const float* prec_w, const int32_t* prec_off, int n_polar_items,
                                              int B, int Ci, int H, int W, int Co, int Kh, int Kw, int groups, const float* gy_t,
                                              const float* x_t, mode_stream_t stream);

int mode_sphere_conv_bwd_weight_win(const float* gy, const float* pos, const float* x, float* gw, float* workspace,
                                    const int32_t* tiles, int n_small, int n_mid, int n_wrap, const float* rec_w,
                                    const int32_t* rec_off, const int32_t* rest_pixels, int n_rest_pixels, const int32_t* pitems,
                                    const float* prec_w, const int32_t* prec_off, int n_polar_items, int B, int Ci, int H, int W,
                                    int Co, int Kh, int Kw, int groups, const float* gy_t, const float* x_t, mode_stream_t stream);

/* Replaces the grad_weight half (sphere_conv_cuda.cpp:296-315: second im2col + addmm_(gO, col^T),
 * summed over the batch).  ACCUMULATES into gw (caller zero-fills, sphere_conv.py:63).  `workspace`
 * holds the deterministic split-K partial sums: >= mode_sphere_conv_bwd_weight_workspace_bytes(). */
size_t mode_sphere_conv_bwd_weight_workspace_bytes(int B, int Ci, int Co, int Kh, int Kw, int Ho, int Wo,
                                                   int groups);

int mode_sphere_conv_bwd_weight(const float* gy, const float* pos, const float* x, float* gw, float* workspace,
                                int B, int Ci, int H, int W, int Co, int Kh, int Kw, int sH, int sW,
                                int Ho, int Wo, int groups, mode_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Concatenation cost volume -- replaces the Python loop models/mode_disparity.py:104-113
 * (host-side zero tensor + H2D copy + 2*D4 strided slice copies + .contiguous()).
 *
 *   ref, tgt (B, C, H, W)  ->  cost (B, 2C, D4, H, W):
 *   cost[b, c, i, h, w] = ref[b, c, h, w], cost[b, C+c, i, h, w] = tgt[b, c, h, w-i] for w >= i, else 0.
 * Every output element is written exactly once (no memset).
 */
int mode_cost_volume_fwd(const float* ref, const float* tgt, float* cost, int B, int C, int D4, int H, int W,
                         mode_stream_t stream);

/* Autograd of the loop above: g_ref[b,c,h,w] = sum_{i<=w} g[b,c,i,h,w];
 * g_tgt[b,c,h,w] = sum_{i<W-w} g[b,C+c,i,h,w+i].  Writes (does not accumulate). */
int mode_cost_volume_bwd(const float* gcost, float* g_ref, float* g_tgt, int B, int C, int D4, int H, int W,
                         mode_stream_t stream);

/* Cost volume + first 3-D convolution without the volume (models/mode_disparity.py:104-116: the concatenation volume followed
 * by dres0[0][0] = Conv3d(2C -> Co, k3 s1 p1, no bias)).  The host forms the 18 partial products over (channel, kh)
 *   R[b, t*Co + o, h, w] = sum_{c,kh} W[o, c,   kd, kh, kw] * ref[b, c, h+kh-1, w],   t = kd*3 + kw,
 *   T[b, t*Co + o, h, u] = sum_{c,kh} W[o, C+c, kd, kh, kw] * tgt[b, c, h+kh-1, u]     (two GEMMs with K = 3C), and
 *   out[b,o,d,h,w] = sum_t [0 <= d' < D][d' <= w' < W] (R_t[b,o,h,w'] + T_t[b,o,h,w'-d']),  d' = d+kd-1, w' = w+kw-1
 * is the layer's output exactly (csrc/cost_conv.hip).  _bwd is the adjoint: gR, gT (B, 9*Co, H, W) from gout (B,Co,D,H,W);
 * both write (do not accumulate) and are deterministic. */
int mode_cost_conv_assemble_fwd(const float* R, const float* T, float* out, int B, int Co, int D, int H, int W, mode_stream_t stream);
int mode_cost_conv_assemble_bwd(const float* gout, float* gR, float* gT, int B, int Co, int D, int H, int W, mode_stream_t stream);

/* dres0[0] = convbn_3d(64, 32) + ReLU on the cost volume in eval mode: the assembly with the BatchNorm scale / shift (+ ReLU)
 * applied on the way out (bn->add must be NULL). */
int mode_cost_conv_assemble_fwd_bn(const float* R, const float* T, const mode_bn_epilogue* bn, float* out, int B, int Co, int D,
                                   int H, int W, mode_stream_t stream);
/* ... and (ABI 31) the maximum buffer of `out` filled on the way (MODE_BN_ABSMAX_FLOATS floats, zeroed by the call; NULL = the plain call):
 * the operand maximum of the next layer when it runs on the fp16 arithmetic in eval mode (mode_conv3d_fwd_split_f16_bn). */
int mode_cost_conv_assemble_fwd_bn_amax(const float* R, const float* T, const mode_bn_epilogue* bn, float* out, float* out_absmax, int B,
                                        int Co, int D, int H, int W, mode_stream_t stream);

/* Weight gradient of the regular 3x3 Conv2d layers of the extractor (nn.Conv2d inside convbn, models/submodule.py:15-17):
 * stride 1, padding = dilation in {1, 2}, no bias, groups 1.  gw (Co, Ci, 3, 3) (+)= sum_{b,h,w} gy[b,o,h,w] * x[b,c,h+(kh-1)*dil,
 * w+(kw-1)*dil]; gy (B,Co,H,W), x (B,Ci,H,W).  Deterministic.  `workspace` >= mode_conv2d_bwd_weight_workspace_bytes().
 * (Forward and input gradient of these layers stay on the vendor library's fp32 Winograd kernels.) */
/* Forward and input gradient of the same layers (used where they beat the vendor's Winograd: quarter resolution and dilation 2).
 * x (B,Ci,H,W), w (Co,Ci,3,3), y (B,Co,H,W); Co <= 128 (Ci <= 128 for _bwd_data).  Both overwrite their output.
 * `wpack` >= mode_conv2d_wpack_bytes(Ci, Co) bytes of scratch (fragment-ordered weights, rebuilt on every call). */
size_t mode_conv2d_wpack_bytes(int Ci, int Co);
int mode_conv2d_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int H, int W, int Co, int dilation,
                    mode_stream_t stream);
int mode_conv2d_bwd_data(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int H, int W, int Co, int dilation,
                         mode_stream_t stream);

/* convbn (models/submodule.py:15-17) in eval mode as one launch: mode_conv2d_fwd with the folded-BatchNorm epilogue. */
int mode_conv2d_fwd_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci, int H,
                       int W, int Co, int dilation, mode_stream_t stream);

/* out (planes, 2*Ho, 2*Wo): out[p][2h][2w] = in[p][h][w], zero elsewhere (Wo even).  The gradients of the extractor's one
 * stride-2 3x3 layer (models/submodule.py:158) are the stride-1 gradients of the zero-inserted output gradient. */
int mode_zero_insert2(const float* in, float* out, long long planes, int Ho, int Wo, mode_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 1x1 convolutions of the extractor (nn.Conv2d(kernel_size=1), models/submodule.py:162, 167-174: the `downsample` branches and
 * lastconv[0] / lastconv[4]; stride 1 or 2, no bias) as plain MFMA GEMMs over the NCHW planes -- csrc/conv1x1.hip.
 *   x (B, Ci, H, W)   w (Co, Ci[, 1, 1])   y (B, Co, Ho, Wo),  Ho = (H - 1) / stride + 1
 *   bwd_data WRITES gx (zeros where a stride-2 layer never read); bwd_weight writes gw (accumulate = 0) or adds to it, needs
 *   Wo % 4 == 0 (stride 2: W % 8 == 0) and `workspace` >= mode_conv1x1_bwd_weight_workspace_bytes().  Deterministic. */
size_t mode_conv1x1_wpack_bytes(int Ci, int Co);

int mode_conv1x1_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int H, int W, int Co, int stride,
                     mode_stream_t stream);

int mode_conv1x1_fwd_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci, int H,
                        int W, int Co, int stride, mode_stream_t stream);

int mode_conv1x1_bwd_data(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int H, int W, int Co, int stride,
                          mode_stream_t stream);

size_t mode_conv1x1_bwd_weight_workspace_bytes(int B, int Ci, int H, int W, int Co, int stride);

int mode_conv1x1_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int H, int W, int Co,
                            int stride, int accumulate, mode_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The stem of the extractor: Conv2d(3 -> 32, kernel 7, stride 2, padding 3, no bias) on the full-resolution image
 * (firstconv[0], models/submodule.py:155) -- csrc/conv_stem.hip.  The image carries no gradient: forward and weight gradient only.
 *   x (B, 3, H, W)   w (Co <= 32, 3, 7, 7)   y (B, Co, Ho, Wo),  Ho = (H - 1) / 2 + 1
 *   bwd_weight writes gw (accumulate = 0) or adds to it; `workspace` >= mode_conv_stem_bwd_weight_workspace_bytes().  Deterministic. */
size_t mode_conv_stem_wpack_bytes(int Ci, int Co);

int mode_conv_stem_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int H, int W, int Co, mode_stream_t stream);

int mode_conv_stem_fwd_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci, int H,
                          int W, int Co, mode_stream_t stream);

size_t mode_conv_stem_bwd_weight_workspace_bytes(int B, int Ci, int H, int W, int Co);

int mode_conv_stem_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int H, int W, int Co,
                              int accumulate, mode_stream_t stream);
size_t mode_conv2d_bwd_weight_workspace_bytes(int B, int Ci, int H, int W, int Co);
int mode_conv2d_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int H, int W, int Co,
                           int dilation, int accumulate, mode_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 3x3x3 convolution, padding 1, no bias -- replaces the cuDNN nn.Conv3d inside convbn_3d
 * (models/submodule.py:20-22) for dres0/dres1, the hourglass stride-1 layers and the classifier bodies
 * (models/mode_disparity.py:15-25, 66-80).  NCDHW fp32, implicit GEMM on fp32 MFMA.
 *
 *   x (B, Ci, D, H, W)   w (Co, Ci, 3, 3, 3)   y (B, Co, Do, Ho, Wo),  Xo = (X - 1) / stride + 1,  stride 1 or 2,
 *   Co and Ci <= 64 per call.  `wpack` >= mode_conv3d_wpack_bytes(Ci, Co) holds the fragment-ordered weights (rebuilt
 *   every call).  bwd_data writes gx (no accumulation; stride 2 needs even D, H, W); bwd_weight writes gw
 *   (accumulate = 0) or adds to it (accumulate = 1) using `workspace` >= mode_conv3d_bwd_weight_workspace_bytes() for
 *   deterministic split-K partial sums.
 *
 * mode_deconv3d_fwd: ConvTranspose3d k3 s2 p1 op1, no bias (hourglass conv5/conv6, models/mode_disparity.py:23, 25):
 *   x (B, Cin, D, H, W)   w (Cin, Cout, 3, 3, 3)   y (B, Cout, 2D, 2H, 2W).
 *   Its backward-data is mode_conv3d_fwd(gy, w, ..., Ci = Cout, Co = Cin, stride = 2) (same memory layout) and its
 *   backward-weight is mode_conv3d_bwd_weight(gy := x, x := gy_out, Ci = Cout, Co = Cin, stride = 2).
 */
size_t mode_conv3d_wpack_bytes(int Ci, int Co);

int mode_conv3d_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int D, int H, int W, int Co,
                    int stride, mode_stream_t stream);

int mode_conv3d_bwd_data(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int D, int H, int W,
                         int Co, int stride, mode_stream_t stream);

/* convbn_3d in eval mode as ONE launch (+ the weight packing): mode_conv3d_fwd / mode_deconv3d_fwd with the folded-BatchNorm
 * epilogue above (Co > 1). */
int mode_conv3d_fwd_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci, int D,
                       int H, int W, int Co, int stride, mode_stream_t stream);

int mode_deconv3d_fwd_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Cin, int D,
                         int H, int W, int Cout, mode_stream_t stream);

/* The same stride-1 layers on the bf16 matrix pipe with three-way split fp32 operands (csrc/conv3d_split.hip): six exact bf16
 * partial products per fp32 product, fp32 accumulation -- the rounding of an fp32 convolution at ~1.7 x the speed of the fp32 MFMA
 * kernels.  mode_conv3d_split_supported() says whether a layer can take this path (stride 1, <= 64 output channels of the GEMM,
 * reduction channels a multiple of 8; in the input-gradient GEMM the roles of Ci / Co are swapped; weight gradient: stride 1 and
 * more than one output channel).  mode_conv3d_fwd_split takes an optional folded-BatchNorm epilogue (NULL: plain convolution);
 * wpack as mode_conv3d_fwd; mode_conv3d_bwd_weight_split: arguments and workspace as mode_conv3d_bwd_weight with stride 1. */
int mode_conv3d_split_supported(int Ci, int Co, int stride, int which /* 0 forward, 1 input gradient, 2 weight gradient */);
/* Stride-2 forward (k3 p1) on the split-bf16 kernel (csrc/conv3d_split_s2.hip): hourglass conv1 / conv3 (mode_disparity.py:17-19) and
 * the input gradient of the transposed convolutions conv5 / conv6 (w = their (Cin, Cout, 27) weight read as (Co = Cin, Ci = Cout));
 * 33..64 output channels, input channels a multiple of 8: mode_conv3d_split_supported(Ci, Co, 2, 0) == 1.  bn: optional folded
 * eval-mode BatchNorm (+ residual) (+ ReLU) epilogue as in mode_conv3d_fwd_split. */
int mode_conv3d_fwd_s2_split(const float* x, const float* w, const mode_bn_epilogue* bn /* optional: eval-mode fold, NULL = plain */,
                             float* y, float* wpack, int B, int Ci, int D, int H, int W, int Co, mode_stream_t stream);
/* (ABI 31) the same with the maximum buffer of y out of the eval epilogue (out_absmax: MODE_BN_ABSMAX_FLOATS floats, zeroed by the call;
 * needs bn; NULL = the plain call) -- see mode_conv3d_fwd_split_f16_bn. */
int mode_conv3d_fwd_s2_split_amax(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* out_absmax, float* wpack, int B,
                                  int Ci, int D, int H, int W, int Co, mode_stream_t stream);

/* The transposed convolution (ConvTranspose3d k3 s2 p1 op1, hourglass conv5 / conv6, mode_disparity.py:23-25) and the input gradient
 * of the stride-2 convolution -- one operator -- on the split-bf16 kernel of csrc/conv3d_split_deconv.hip: input channels of the
 * transposed convolution a multiple of 8, 2..64 output channels (mode_deconv3d_split_supported(Cin, Cout) /
 * mode_conv3d_split_supported(Ci, Co, 2, 1)); even D, H, W for the gradient form. */
int mode_deconv3d_split_supported(int Cin, int Cout);
int mode_deconv3d_fwd_split(const float* x, const float* w, float* y, float* wpack, int B, int Cin, int D, int H, int W, int Cout,
                            mode_stream_t stream);
/* The same with the folded eval-mode BatchNorm (+ residual) (+ ReLU) epilogue of mode_deconv3d_fwd_bn (convbn_3d around the
 * ConvTranspose3d of hourglass conv5 / conv6 in eval mode, mode_disparity.py:23-25, 38-45): whole 32-channel output tiles
 * (mode_deconv3d_split_bn_supported(Cin, Cout) == 1); wpack >= mode_conv3d_wpack_bytes(Cin, Cout). */
/* The arithmetic of the stride-1 3x3x3 layers in a TRAINING step (nn.Conv3d of convbn_3d, models/submodule.py:20-22; the product
 * default since round 5, functional.CONV3D_S1_F16): fp32 operands split into TWO fp16 pieces, three v_mfma_f32_32x32x16_f16 per product
 * instead of six bf16 ones (2^-22 per product).  fp16's range is narrow, so each operand is scaled by a power of two that brings its
 * tensor's largest finite magnitude to [2^14, 2^15): amax_* = DEVICE buffers of MODE_BN_ABSMAX_FLOATS floats, one per operand (layout:
 * see MODE_BN_ABSMAX_FLOATS below), filled by mode_abs_max (an order-independent maximum over bit patterns; no host synchronisation,
 * graph-capturable) or by the `_amax` BatchNorm entries that write the tensor; a tensor's maximum can be computed once and passed to
 * every call that reads the tensor.  PRECISION CONTRACT: an element keeps 22 significant bits down to ~2^-17 of its tensor's maximum,
 * fewer below that, none below ~2^-39 of it (it contributes less than 2^-17 of the tensor's scale to any sum either way; the three-piece
 * bf16 entries keep 24 bits for every element) -- DESIGN.md 3u.
 * mode_conv3d_bwd_data_split_f16: acc may be NULL.  mode_conv3d_bwd_weight_split_f16: other arguments and workspace as
 * mode_conv3d_bwd_weight_split. */
int mode_abs_max(const float* x, long long n, float* out_device_buffer, mode_stream_t stream);
/* The same for n tensors in ONE launch (one workgroup per tensor: meant for the tens of small WEIGHT tensors of a model, whose maxima a
 * training step needs before its first convolution): device_ptrs / device_sizes = DEVICE arrays of n tensor addresses / element counts,
 * out = n consecutive maximum buffers (n * MODE_BN_ABSMAX_FLOATS floats), all written. */
int mode_abs_max_batch(const float* const* device_ptrs, const long long* device_sizes, int n, float* out, mode_stream_t stream);
int mode_conv3d_fwd_split_f16(const float* x, const float* w, const float* amax_x, const float* amax_w, float* y, float* wpack, int B, int Ci,
                              int D, int H, int W, int Co, mode_stream_t stream);
int mode_conv3d_bwd_data_split_f16(const float* gy, const float* w, const float* amax_g, const float* amax_w, const float* acc, float* gx,
                                   float* wpack, int B, int Ci, int D, int H, int W, int Co, mode_stream_t stream);
/* INFERENCE on the same arithmetic (ABI 31): y = relu?(bn(conv3d(x, w)) [+ bn->add]) as mode_conv3d_fwd_split with a non-NULL epilogue, on
 * two fp16 pieces.  The BatchNorm scale is folded into the weights before they are scaled and split: their maximum is the FOLDED weights',
 * taken inside the call where they are packed and kept in wpack (same size as for mode_conv3d_fwd_split; mode_pack_reuse applies to it as
 * to the packed weights).  In eval mode no BatchNorm pass writes the activations, so the kernel's epilogue leaves the maximum of what it
 * stored in amax_y (MODE_BN_ABSMAX_FLOATS floats, zeroed and filled by the call): the next layer's amax_x. */
int mode_conv3d_fwd_split_f16_bn(const float* x, const float* w, const float* amax_x, const mode_bn_epilogue* bn, float* y, float* amax_y,
                                 float* wpack, int B, int Ci, int D, int H, int W, int Co, mode_stream_t stream);
int mode_conv3d_bwd_weight_split_f16(const float* gy, const float* x, const float* amax_g, const float* amax_x, float* gw, float* workspace,
                                     int B, int Ci, int D, int H, int W, int Co, int accumulate, mode_stream_t stream);
/* The input gradient of a stride-1 / stride-2 convolution on the split kernels with a gradient that is already there added in the
 * store: gx = conv^T(gy) + acc, bit for bit the sum autograd would form with a separate pass (x has a second consumer whose gradient
 * came first: a residual skip, a classifier head).  acc has gx's shape and must not alias it; stride 2: whole 32-channel tiles of gx
 * (mode_conv3d_bwd_data_split_acc_supported(Ci, Co, stride) == 1), even D, H, W. */
int mode_conv3d_bwd_data_split_acc_supported(int Ci, int Co, int stride);
int mode_conv3d_bwd_data_split_acc(const float* gy, const float* w, const float* acc, float* gx, float* wpack, int B, int Ci, int D, int H,
                                   int W, int Co, int stride, mode_stream_t stream);
int mode_deconv3d_split_bn_supported(int Cin, int Cout);
int mode_deconv3d_fwd_split_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Cin, int D,
                               int H, int W, int Cout, mode_stream_t stream);
int mode_deconv3d_fwd_split_bn_amax(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* out_absmax /* as above */,
                                    float* wpack, int B, int Cin, int D, int H, int W, int Cout, mode_stream_t stream);
int mode_conv3d_bwd_data_s2_split(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int D, int H, int W, int Co,
                                  mode_stream_t stream);
/* The weight gradient of the stride-2 convolution on the split-bf16 kernel of csrc/conv3d_split_wgrad_s2.hip (the gradients cuDNN
 * computes for hourglass conv1 / conv3, mode_disparity.py:17-19; called with gy := the input of a ConvTranspose3d and x := the gradient
 * of its output it gives that layer's (Cin, Cout, 27) weight gradient: conv5 / conv6, :23-25).  Arguments and workspace as
 * mode_conv3d_bwd_weight with stride 2; x channels a multiple of 32, gy channels a multiple of 64
 * (mode_conv3d_split_supported(Ci, Co, 2, 2) == 1), D and H even, W a multiple of 8. */
int mode_conv3d_bwd_weight_s2_split(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int D, int H, int W,
                                    int Co, int accumulate, mode_stream_t stream);

int mode_conv3d_fwd_split(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci, int D,
                          int H, int W, int Co, mode_stream_t stream);
int mode_conv3d_bwd_data_split(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int D, int H, int W, int Co,
                               mode_stream_t stream);
int mode_conv3d_bwd_weight_split(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int D, int H, int W,
                                 int Co, int accumulate, mode_stream_t stream);

/* The regular 3x3 Conv2d layers (stride 1, dilation 1 / 2) on the same split-bf16 path (csrc/conv2d_split.hip): reduction channels
 * a multiple of 16, <= 512 output channels of the GEMM (one launch per block of 64); arguments and wpack as mode_conv2d_fwd / _bwd_data
 * (+ optional epilogue). */
int mode_conv2d_split_supported(int Ci, int Co, int dilation, int which /* 0 forward, 1 input gradient */);
int mode_conv2d_fwd_split(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci, int H, int W,
                          int Co, int dilation, mode_stream_t stream);
int mode_conv2d_bwd_data_split(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int H, int W, int Co, int dilation,
                               mode_stream_t stream);
/* The same with a gradient of the same tensor that is already there added in the store (gx = conv^T(gy) + acc, bit for bit autograd's
 * sum; acc has gx's shape and must not alias it): the input of a residual block of the extractor has two consumers, its first
 * convolution and the skip (models/submodule.py:36-46, 110-118).  Same predicate as mode_conv2d_bwd_data_split. */
int mode_conv2d_bwd_data_split_acc(const float* gy, const float* w, const float* acc, float* gx, float* wpack, int B, int Ci, int H, int W,
                                   int Co, int dilation, mode_stream_t stream);
/* mode_conv2d_fwd_split / mode_conv2d_bwd_data_split(_acc) of a TRAINING step on the two-piece fp16 arithmetic of the stride-1 3-D layers
 * (two fp16 pieces per value, three MFMAs per product, a power-of-two scale per operand): amax_* = the maximum buffers
 * (MODE_BN_ABSMAX_FLOATS floats: mode_abs_max, mode_bn_train_fwd_amax, mode_bn_train_bwd_amax) of the activation / gradient and of the
 * weight.  Same predicate (mode_conv2d_split_supported) and workspace; acc may be NULL. */
int mode_conv2d_fwd_split_f16(const float* x, const float* w, const float* amax_x, const float* amax_w, float* y, float* wpack, int B, int Ci,
                              int H, int W, int Co, int dilation, mode_stream_t stream);
int mode_conv2d_bwd_data_split_f16(const float* gy, const float* w, const float* amax_g, const float* amax_w, const float* acc, float* gx,
                                   float* wpack, int B, int Ci, int H, int W, int Co, int dilation, mode_stream_t stream);
/* INFERENCE on that arithmetic (ABI 31), the 2-D twin of mode_conv3d_fwd_split_f16_bn: mode_conv2d_fwd_split with a non-NULL epilogue on two
 * fp16 pieces; the folded weights' maximum is taken inside with the pack and kept in wpack (mode_conv2d_wpack_bytes has the room since
 * ABI 31), amax_x = the input's maximum buffer, amax_y = the stored output's (zeroed and filled by the call: the next layer's amax_x). */
int mode_conv2d_fwd_split_f16_bn(const float* x, const float* w, const float* amax_x, const mode_bn_epilogue* bn, float* y, float* amax_y,
                                 float* wpack, int B, int Ci, int H, int W, int Co, int dilation, mode_stream_t stream);
int mode_conv2d_bwd_weight_split(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int H, int W, int Co,
                                 int dilation, int accumulate, mode_stream_t stream);
/* ... on the two-piece fp16 arithmetic (mode_conv2d_fwd_split_f16): amax_g / amax_x = the maximum buffers of gy and of x. */
int mode_conv2d_bwd_weight_split_f16(const float* gy, const float* x, const float* amax_g, const float* amax_x, float* gw, float* workspace,
                                     int B, int Ci, int H, int W, int Co, int dilation, int accumulate, mode_stream_t stream); /* arguments / workspace: mode_conv2d_bwd_weight */

size_t mode_conv3d_bwd_weight_workspace_bytes(int B, int Ci, int D, int H, int W, int Co, int stride);

int mode_conv3d_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int D, int H,
                           int W, int Co, int stride, int accumulate, mode_stream_t stream);

int mode_deconv3d_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Cin, int D, int H, int W,
                      int Cout, mode_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused soft-argmin head -- replaces F.upsample(trilinear, align_corners=True) + F.softmax +
 * disparityregression (models/mode_disparity.py:131-152, models/submodule.py:50-57) and, when `conf` is non-NULL, the
 * confidence map of models/mode_disparity.py:157-183.  The (B,D,H,W) upsampled logits / probabilities are never
 * materialised.
 *
 *   logits (B, D4, H4, W4) [= the (B,1,D4,H4,W4) classifier output]  ->  pred (B, H, W) [, conf (B, H, W)]
 *   pred = sum_d d * softmax_d(upsample(logits))[d],   d = 0 .. D-1
 * mode_head_bwd: glogits (B,D4,H4,W4) = d(sum(gpred * pred)) / d logits, written (not accumulated); deterministic.
 * `workspace` >= mode_head_bwd_workspace_bytes(B, D4, H, W).
 */
int mode_head_fwd(const float* logits, float* pred, float* conf, int B, int D4, int H4, int W4, int D, int H, int W,
                  mode_stream_t stream);

size_t mode_head_bwd_workspace_bytes(int B, int D4, int H, int W);

int mode_head_bwd(const float* logits, const float* gpred, float* glogits, float* workspace, int B, int D4, int H4,
                  int W4, int D, int H, int W, mode_stream_t stream);

/* The loss of the training step next to the head (train_disparity.py:151-158: masked smooth-L1 of the three predictions, weights
 * 0.5 / 0.7 / 1.0, mean over the valid pixels), so that nothing elementwise is launched between the head and the optimizer:
 *   mode_smooth_l1_masked: out[0] = scale[0] * sum_i w_i * sum_pix [gt == gt] * smooth_l1(pred_i[pix] - gt[pix])   (beta = 1);
 *     pred1 / pred2 may be NULL; NaN ground truth = masked pixel (train_disparity.py:195); scale = a DEVICE scalar, normally
 *     1 / (number of valid pixels over all ranks); n = B * H * W; workspace >= mode_smooth_l1_workspace_bytes(n); deterministic.
 *   mode_head_bwd_loss: mode_head_bwd with gpred = weight * scale[0] * [gt == gt] * clamp(pred - gt, -1, 1) formed inside the kernel from
 *     the forward's own prediction `pred` (B, H, W) -- the gradient of the term weight * scale * sum smooth_l1(pred - gt) above; `scale`
 *     here also carries the upstream gradient of the loss.  workspace as mode_head_bwd.  Only where mode_head_loss_supported(...) == 1
 *     (D = 4 * D4 with D4 one of 4 / 8 / 12 / 16 / 48 / 64, W <= 512); otherwise form gpred and call mode_head_bwd.
 */
size_t mode_smooth_l1_workspace_bytes(long long n);

int mode_smooth_l1_masked(const float* pred0, const float* pred1, const float* pred2, const float* gt, float w0, float w1, float w2,
                          const float* scale, float* out, float* workspace, long long n, mode_stream_t stream);

int mode_head_loss_supported(int B, int D4, int H4, int W4, int D, int H, int W);

int mode_head_bwd_loss(const float* logits, const float* pred, const float* gt, float weight, const float* scale, float* glogits,
                       float* workspace, int B, int D4, int H4, int W4, int D, int H, int W, mode_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm (+ residual add) (+ ReLU) over (B, C, S) tensors, S = D*H*W or H*W, S % 4 == 0.
 * Replaces nn.BatchNorm3d/2d of convbn_3d / convbn (models/submodule.py:15-22) fused with the adds and ReLUs that
 * follow it in hourglass.forward / ModeDisparity.forward (models/mode_disparity.py:27-46, 115-129):
 *      out = relu?( gamma * (y - mean) / sqrt(var + eps) + beta  [+ add] )
 * train: batch statistics over (B, S) per channel (biased variance for normalisation); running_mean / running_var are
 *        updated in place with `momentum` (unbiased variance) and *num_batches_tracked is incremented (either may be
 *        NULL), as nn.BatchNorm does; save_mean / save_invstd feed the backward.  Two launches.
 * eval : running statistics.  One launch.
 * bwd  : g = relu ? gout * (out > 0) : gout;  gy = dL/dy written; gadd (optional, = g) written if non-NULL;
 *        ggamma / gbeta written (accumulate = 0) or added to (accumulate = 1: the caller's gradient buffer, which saves
 *        the separate accumulation kernels of autograd).  Two launches.
 *        The ReLU mask comes from `out`; when no residual was added, `out` may be NULL and the mask is rebuilt bit-exactly
 *        from y with the float32 save_scale / save_shift (C floats each, optional outputs of the forward): one tensor less
 *        to read in both backward passes.
 * groups: statistics over `groups` consecutive sub-batches of B / groups samples each (1 = the whole batch) -- what `groups`
 *        consecutive calls of the module on the sub-batches would compute, including the order of the running-statistics
 *        updates; save_* hold groups * C values.  Used to run the left and right images through the shared extractor as one batch.
 * `workspace` >= mode_bn_workspace_bytes(C * groups) for the training calls.
 * mode_bn_train_fwd is NOT in-place: `out` must not alias `y` (the normalisation pass re-reads one element per channel of y -- the
 *        pivot of the shifted sums -- while it writes `out`); out == y returns MODE_ERR_BAD_ARG.
 */
size_t mode_bn_workspace_bytes(int C);

int mode_bn_train_fwd(const float* y, const float* add, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu,
                      float* out, float* save_mean, float* save_invstd, float* save_scale, float* save_shift,
                      float* workspace, int B, int C, long long S, int groups, mode_stream_t stream);

/* A tensor's largest finite magnitude, as the fp16-arithmetic entries (mode_conv3d_*_split_f16, mode_sphere_conv_fwd_win_split_f16) take
 * it: a device buffer of MODE_BN_ABSMAX_FLOATS floats whose MAXIMUM is the value -- word 0 and 128 words of 128 different cache lines,
 * everything else zero (the blocks of the producing pass each add to one of them: one word for a whole launch serialises at the memory
 * side; the consumers' waves read all 129).  mode_abs_max fills such a buffer with a pass over the tensor; the `_amax` entries below fill
 * one on the way -- the operand maximum comes out of the pass that WRITES the tensor: mode_bn_train_fwd_amax (+ _prestats_amax) for `out`,
 * mode_bn_train_bwd_amax and mode_classif_train_bwd_amax for the gradient `gy` (read by both gradients of the convolution in front).
 * Same arguments as the plain entries plus the buffer (NULL = the plain call); the call zeroes the whole buffer first.  (ABI <= 29 had
 * one-shot thread-local setters, mode_bn_next_out_absmax / mode_bn_next_gy_absmax, instead: removed.) */
#define MODE_BN_ABSMAX_FLOATS 2064
int mode_bn_train_fwd_amax(const float* y, const float* add, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, float* out,
                           float* save_mean, float* save_invstd, float* save_scale, float* save_shift, float* workspace, int B, int C,
                           long long S, int groups, float* out_absmax, mode_stream_t stream);

int mode_bn_eval_fwd(const float* y, const float* add, const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float eps, int relu, float* out, int B, int C, long long S,
                     mode_stream_t stream);

int mode_bn_train_bwd(const float* gout, const float* y, const float* out, const float* gamma, const float* save_mean,
                      const float* save_invstd, const float* save_scale, const float* save_shift, int relu, float* gy,
                      float* gadd, float* ggamma, float* gbeta, int accumulate, float* workspace, int B, int C,
                      long long S, int groups, mode_stream_t stream);
int mode_bn_train_bwd_amax(const float* gout, const float* y, const float* out, const float* gamma, const float* save_mean,
                           const float* save_invstd, const float* save_scale, const float* save_shift, int relu, float* gy,
                           float* gadd, float* ggamma, float* gbeta, int accumulate, float* workspace, int B, int C,
                           long long S, int groups, float* gy_absmax, mode_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Classifier head of the 3-D regulariser in TRAINING (SURVEY a12 + a15):
 *     classifN = Sequential(convbn_3d(32, 32), ReLU, Conv3d(32, 1, k3 p1, bias=False))      models/mode_disparity.py:76-80
 *     cost1 = classif1(out1); cost2 = classif2(out2) + cost1; cost3 = classif3(out3) + cost2  models/mode_disparity.py:127-129
 * taken from the OUTPUT y (B, C, D, H, W) of the first convolution on: BatchNorm3d (batch statistics, torch defaults as in
 * mode_bn_train_fwd) + ReLU + the single-channel convolution [+ the residual `add` (B, 1, D, H, W), may be NULL], without the
 * activated tensor relu(bn(y)) ever being written (csrc/classif_head.hip: the normalisation is applied while the second convolution
 * stages its operand; in the backward ONE pass over y yields the weight gradient of the second convolution and both reductions of the
 * BatchNorm backward, a second one its input gradient with the BatchNorm backward's apply pass in the store).  C <= 32.
 *   fwd: cost (B, 1, D, H, W) written; running_mean / running_var / *num_batches_tracked updated (any may be NULL);
 *        save_mean / save_invstd / save_scale / save_shift (C floats each) feed the backward.
 *   bwd: gcost (B, 1, D, H, W) = dL/dcost -> gy (B, C, D, H, W) = dL/dy written; gw (C * 27) = gradient of the second convolution's
 *        weight (1, C, 3, 3, 3), ggamma / gbeta (C): written (accumulate = 0) or added to (accumulate = 1).  The gradient of `add` is
 *        gcost itself.  Deterministic (fixed-order split-K).
 * `workspace` >= mode_classif_workspace_bytes(B, C, D, H, W), 16-byte aligned, for both calls.
 */
size_t mode_classif_workspace_bytes(int B, int C, int D, int H, int W);

int mode_classif_train_fwd(const float* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                           long long* num_batches_tracked, float momentum, float eps, const float* w, const float* add, float* cost,
                           float* save_mean, float* save_invstd, float* save_scale, float* save_shift, float* workspace, int B, int C,
                           int D, int H, int W, mode_stream_t stream);

int mode_classif_train_bwd(const float* gcost, const float* y, const float* w, const float* gamma, const float* beta,
                           const float* save_mean, const float* save_invstd, const float* save_scale, const float* save_shift, float* gy,
                           float* gw, float* ggamma, float* gbeta, int accumulate, float* workspace, int B, int C, int D, int H, int W,
                           mode_stream_t stream);
int mode_classif_train_bwd_amax(const float* gcost, const float* y, const float* w, const float* gamma, const float* beta,
                                const float* save_mean, const float* save_invstd, const float* save_scale, const float* save_shift,
                                float* gy, float* gw, float* ggamma, float* gbeta, int accumulate, float* workspace, int B, int C, int D,
                                int H, int W, float* gy_absmax /* may be NULL */, mode_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Export-stage geometry: what happens between the disparity network and the fusion network.
 *
 * mode_disp2depth: sine-rule depth of save_output_disparity_stage.py:105-135 for a Cassini disparity map (H, W);
 *   disp == 0 -> 1000, clamp to [0, 1000]; `baseline` = the camera-pair baseline.
 * mode_grid_sample_border: F.grid_sample(bilinear, align_corners=True, padding_mode='border') as used by cassini2Equirec /
 *   rotateCassini / erp2rect_cassini (utils/geometry.py:38, 92, 188); src (N,C,Hs,Ws), grid (grids,Ho,Wo,2) with grids = 1
 *   (shared) or N, dst (N,C,Ho,Wo).
 * mode_depth_view_trans: depthViewTransWithConf (utils/geometry.py:99-145) including the sequential z-buffer of
 *   __iterPixels_with_conf (:148-156), reproduced bit-for-bit with a 64-bit atomic min.  view1/conf1/view2/conf2 (H, W);
 *   trig = the float32 values the reference computes with numpy, concatenated: sin phi[W], cos phi[W], sin theta[H],
 *   cos theta[H] (device array of 2W + 2H floats; the products r*sin(phi), (r*cos(phi))*sin(theta), (r*cos(phi))*cos(theta) are
 *   formed in that order like numpy's); R (9 doubles, row-major) and t (3 doubles) on the HOST;
 *   workspace >= mode_depth_view_trans_workspace_bytes.
 */
int mode_disp2depth(const float* disp, float* depth, int H, int W, float baseline, mode_stream_t stream);

int mode_grid_sample_border(const float* src, const float* grid, float* dst, int N, int C, int Hs, int Ws, int Ho, int Wo,
                            int grids, mode_stream_t stream);

size_t mode_depth_view_trans_workspace_bytes(int H, int W);

int mode_depth_view_trans(const float* view1, const float* conf1, const float* trig, const double* R, const double* t,
                          float* view2, float* conf2, void* workspace, int H, int W, mode_stream_t stream);

/* The two halves of mode_depth_view_trans on their own: the projection of every source pixel (r2 in float64, target index
 * i*W + j or -1), and the z-buffer over n given (r2, target) pairs (workspace >= 8 n bytes).  The z-buffer is exact; the
 * projection rounds angles to pixel indices, and the reference's maps put whole families of points exactly on the rounding
 * boundary, where the last bit of atan2 / asin decides -- parity of the projection is therefore tested off those boundaries. */
int mode_depth_view_project(const float* view1, const float* trig, const double* R, const double* t, double* r2, int32_t* target,
                            int H, int W, mode_stream_t stream);

int mode_zbuffer(const double* r2, const int32_t* target, const float* conf1, float* view2, float* conf2, void* workspace,
                 long long n, mode_stream_t stream);

/* Training forward of convbn_3d (models/submodule.py:20-22) without the statistics pass: the stride-1 split-bf16 convolution kernel
 * takes the BatchNorm batch statistics of its output from the accumulators (per channel: sum(y - K), sum((y - K)^2), K = the layer's own
 * first output value) and leaves them in `stats` = the BatchNorm workspace (>= mode_bn_workspace_bytes(Co) bytes) as
 * mode_conv3d_fwd_split_stats_partials() pairs per channel + the pivots; mode_bn_train_fwd_prestats(..., nsplit = that number, ...) then
 * is mode_bn_train_fwd minus its statistics kernel (one statistics group).  Needs mode_conv3d_split_supported(Ci, Co, 1, 0) == 1. */
int mode_conv3d_fwd_split_stats_partials(void);
int mode_conv3d_fwd_split_stats(const float* x, const float* w, float* y, float* wpack, float* stats, int B, int Ci, int D, int H, int W,
                                int Co, mode_stream_t stream);
int mode_bn_train_fwd_prestats(const float* y, const float* add, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, float* out,
                               float* save_mean, float* save_invstd, float* save_scale, float* save_shift, float* workspace, int nsplit,
                               int B, int C, long long S, mode_stream_t stream);
int mode_bn_train_fwd_prestats_amax(const float* y, const float* add, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, float* out,
                                    float* save_mean, float* save_invstd, float* save_scale, float* save_shift, float* workspace,
                                    int nsplit, int B, int C, long long S, float* out_absmax /* as mode_bn_train_fwd_amax */,
                                    mode_stream_t stream);

/* out = a + b [+ c [+ d]] (c, d may be NULL; fixed association): the gradient of a tensor with several consumers in ONE pass.
 * No reference counterpart as code: autograd's pairwise accumulation of the gradients of cost0 / pre1
 * (models/mode_disparity.py:119-125) is what it replaces.  n elements, 16-byte aligned buffers. */
int mode_sum_n(const float* a, const float* b, const float* c, const float* d, float* out, long long n, mode_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fusion network (models/mode_fusion.py:91-307, built as train_fusion.py:64): the layers that are neither a 3x3
 * convolution (mode_conv2d_*) nor a BatchNorm (mode_bn_*).  csrc/fusion_ops.hip; all deterministic, fp32.
 *
 * mode_maxpool2x2_fwd / _bwd: nn.MaxPool2d(2, stride=2) (mode_fusion.py:146, :161, :190) over N = B * C planes (H, W) -> (H / 2, W / 2),
 *   floor; picks like torch (scan order, a later value replaces the current one when it is greater or NaN); the backward recomputes
 *   the choice from x (no index tensor) and writes all of gx (zeros where nothing was picked).
 * mode_depth_to_space2 / mode_space_to_depth2: the rearrangement half of nn.ConvTranspose2d(Ci, Co, 2, 2) (mode_fusion.py:195, :212) --
 *   kernel 2, stride 2: no overlapping taps, so the layer is ONE 1x1 convolution with 4 Co output channels (row 4 o + 2 i + j of the
 *   weight's own (Ci, 4 Co) storage: mode_conv1x1_bwd_data on x with the weight as it lies) followed by
 *       y[b, o, 2h+i, 2w+j] = relu?(scale[o] * y4[b, 4 o + 2 i + j, h, w] + shift[o])        (scale NULL = 1, shift NULL = 0)
 *   with the bias (training) or the bias and the folded eval-mode BatchNorm (inference) as the per-channel affine.  The inverse
 *   rearranges the output gradient for mode_conv1x1_fwd (input gradient) / mode_conv1x1_bwd_weight (weight gradient).
 * mode_conv1x1_sigmoid_fwd / _bwd: nn.Conv2d(C, 1, 1, bias=True) + nn.Sigmoid (mode_fusion.py:228-229) over (B, C, S) planes, S % 4 == 0,
 *   C <= 64: s = sigmoid(bias + sum_c w[c] x[b, c, :]).  bwd: gs = dL/ds -> gx (B, C, S) written (may be NULL), gw (C) and gbias (1, may be
 *   NULL) written (accumulate = 0) or added to (accumulate = 1); workspace >= mode_conv1x1_sigmoid_bwd_workspace_bytes(B, S). */
int mode_maxpool2x2_fwd(const float* x, float* y, long long N, int H, int W, mode_stream_t stream);
int mode_maxpool2x2_bwd(const float* x, const float* gy, float* gx, long long N, int H, int W, mode_stream_t stream);
int mode_depth_to_space2(const float* y4, const float* scale, const float* shift, float* y, int B, int Co, int H, int W, int relu,
                         mode_stream_t stream);
int mode_space_to_depth2(const float* gy, float* g4, int B, int Co, int H, int W, mode_stream_t stream);
int mode_conv1x1_sigmoid_fwd(const float* x, const float* w, const float* bias, float* s, int B, int C, long long S, mode_stream_t stream);
size_t mode_conv1x1_sigmoid_bwd_workspace_bytes(int B, long long S);
int mode_conv1x1_sigmoid_bwd(const float* x, const float* w, const float* s, const float* gs, float* gx, float* gw, float* gbias,
                             int accumulate, float* workspace, int B, int C, long long S, mode_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MODE_HIP_H_ */
