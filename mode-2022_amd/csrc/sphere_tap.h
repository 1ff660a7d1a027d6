// Sampling record of one (output pixel, tap) of the spherical convolution, shared by host planning code and device kernels.
//
// Reference arithmetic: sphere_conv_cuda_kernel.cu:83-113 (bilinear read with per-corner zero padding) and :246 (the whole
// tap contributes nothing unless -1 < h < H and -1 < w < W).
#pragma once
#include <hip/hip_runtime.h>

namespace mode {

// "Fixed-step" form used by the windowed kernels: the four corners are always (r0, c0), (r0, c0+1), (r0+1, c0), (r0+1, c0+1)
// with 0 <= r0 <= H-1 and 0 <= c0 <= W-1, and wt = their weights in that order.  Corners outside the image get weight 0
// (so whatever finite value is read for them does not matter); when the LOW corner is the one outside (h or w in (-1, 0)),
// the record is shifted so that the valid corner sits at r0 / c0.  Returns false (and zero weights) for a dead tap.
__host__ __device__ inline bool tap_record_fixed(float h, float w, int H, int W, int& r0, int& c0, float4& wt) {
  r0 = 0;
  c0 = 0;
  wt.x = wt.y = wt.z = wt.w = 0.f;
  if (!(h > -1.f && w > -1.f && h < (float)H && w < (float)W)) return false;
  const float hf = floorf(h), wf = floorf(w);
  const int hl = (int)hf, wl = (int)wf;
  const int hh = hl + 1, wh = wl + 1;
  const float lh = h - hf, lw = w - wf;
  const float uh = 1.f - lh, uw = 1.f - lw;
  float tl = (hl >= 0 && wl >= 0) ? uh * uw : 0.f;
  float tr = (hl >= 0 && wh <= W - 1) ? uh * lw : 0.f;
  float bl = (hh <= H - 1 && wl >= 0) ? lh * uw : 0.f;
  float br = (hh <= H - 1 && wh <= W - 1) ? lh * lw : 0.f;
  r0 = hl;
  c0 = wl;
  if (hl < 0) {  // rows -1 | 0: only the lower row exists; move it to the first slot
    r0 = 0;
    tl = bl;
    tr = br;
    bl = br = 0.f;
  }
  if (wl < 0) {
    c0 = 0;
    tl = tr;
    bl = br;
    tr = br = 0.f;
  }
  wt.x = tl;
  wt.y = tr;
  wt.z = bl;
  wt.w = br;
  return true;
}

}  // namespace mode
