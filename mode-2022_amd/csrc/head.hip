// Fused soft-argmin head: trilinear x4 upsample (align_corners=True) + softmax over disparity + expectation, and the
// optional confidence map.  Reference: models/mode_disparity.py:131-152 (F.upsample + F.softmax), models/submodule.py:50-57
// (disparityregression), models/mode_disparity.py:157-183 (confidence).
//
// The reference materialises the upsampled logits (B,192,1024,512) = 402.7 MB and the same-size softmax per head (three
// heads in training).  Here neither exists: one thread owns one output pixel, interpolates the D/4 low-resolution logits of
// its column bilinearly in (h,w) into LDS, then walks the D disparities (linear interpolation along d) with a running
// exp-sum.  Algorithmic traffic: read B*D4*H4*W4*4 (6.3 MB) + write B*H*W*4 (2.1 MB) per head and sample.
//
// Backward (d pred / d logit): gv_d = g * p_d * (d - pred); its transpose-interpolation along d is accumulated per pixel
// (kernel 1, writes G = (B,D4,H,W)), and the (h,w) transpose is a deterministic gather over the <= ~(2*scale)^2 pixels that
// touch a low-resolution node (kernel 2).  No atomics.
#include "common.h"

namespace {

constexpr int NT = 256;

struct HDims {
  int B, D4, H4, W4, D, H, W;
  float sd, sh, sw;  // align_corners scales (in-1)/(out-1)
};

__device__ __forceinline__ void src_index(int o, float scale, int in, int& i0, int& i1, float& l1) {
  // area_pixel_compute_source_index(align_corners=True) + the index/lambda computation of upsample_*linear
  const float s = scale * (float)o;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l1 = s - (float)i0;
}

// u[d4] for this thread's pixel -> LDS column; returns max over d4
__device__ __forceinline__ float fill_column(const float* __restrict__ Lb, const HDims& d, int h, int w, float* ucol) {
  int h0, h1, w0, w1;
  float lh, lw;
  src_index(h, d.sh, d.H4, h0, h1, lh);
  src_index(w, d.sw, d.W4, w0, w1, lw);
  const float uh = 1.f - lh, uw = 1.f - lw;
  const int plane = d.H4 * d.W4;
  const int o00 = h0 * d.W4 + w0, o01 = h0 * d.W4 + w1, o10 = h1 * d.W4 + w0, o11 = h1 * d.W4 + w1;
  float m = -INFINITY;
  for (int k = 0; k < d.D4; ++k) {
    const float* p = Lb + (long long)k * plane;
    const float u = uh * (uw * p[o00] + lw * p[o01]) + lh * (uw * p[o10] + lw * p[o11]);
    ucol[k * NT] = u;
    m = fmaxf(m, u);
  }
  return m;
}

__device__ __forceinline__ float logit_at(const float* ucol, const HDims& d, int dd) {
  int d0, d1;
  float ld;
  src_index(dd, d.sd, d.D4, d0, d1, ld);
  return (1.f - ld) * ucol[d0 * NT] + ld * ucol[d1 * NT];
}

__global__ __launch_bounds__(NT) void head_fwd_kernel(const float* __restrict__ L, float* __restrict__ pred,
                                                      float* __restrict__ conf, HDims d) {
  extern __shared__ __attribute__((aligned(16))) float u[];  // [D4][NT]
  const long long npix = (long long)d.B * d.H * d.W;
  const long long pix = (long long)blockIdx.x * NT + threadIdx.x;
  if (pix >= npix) return;  // no barriers in this kernel
  const int w = (int)(pix % d.W);
  const int h = (int)((pix / d.W) % d.H);
  const int b = (int)(pix / ((long long)d.W * d.H));
  float* ucol = u + threadIdx.x;
  const float m = fill_column(L + (long long)b * d.D4 * d.H4 * d.W4, d, h, w, ucol);
  float s0 = 0.f, s1 = 0.f;
  for (int dd = 0; dd < d.D; ++dd) {
    const float e = __expf(logit_at(ucol, d, dd) - m);
    s0 += e;
    s1 += e * (float)dd;
  }
  const float p = s1 / s0;
  pred[pix] = p;
  if (conf) {
    // P(round(p)-1) + P(round(p)) + P(round(p)+1), indices clamped to the border (mode_disparity.py:159-180)
    const float r = rintf(p);
    float c = 0.f;
#pragma unroll
    for (int off = -1; off <= 1; ++off) {
      const int idx = (int)fminf(fmaxf(r + (float)off, 0.f), (float)(d.D - 1));
      c += __expf(logit_at(ucol, d, idx) - m);
    }
    conf[pix] = c / s0;
  }
}

// Backward kernel 1: G[b][d4][h][w] = sum_d (lerp weight of d4 at d) * gpred * p_d * (d - pred)
__global__ __launch_bounds__(NT) void head_bwd_pix_kernel(const float* __restrict__ L, const float* __restrict__ gpred,
                                                          float* __restrict__ G, HDims d) {
  extern __shared__ __attribute__((aligned(16))) float u[];  // [D4][NT]: logits column
  const long long npix = (long long)d.B * d.H * d.W;
  const long long pix = (long long)blockIdx.x * NT + threadIdx.x;
  if (pix >= npix) return;
  const int w = (int)(pix % d.W);
  const int h = (int)((pix / d.W) % d.H);
  const int b = (int)(pix / ((long long)d.W * d.H));
  float* ucol = u + threadIdx.x;
  const float m = fill_column(L + (long long)b * d.D4 * d.H4 * d.W4, d, h, w, ucol);
  float s0 = 0.f, s1 = 0.f;
  for (int dd = 0; dd < d.D; ++dd) {
    const float e = __expf(logit_at(ucol, d, dd) - m);
    s0 += e;
    s1 += e * (float)dd;
  }
  const float p = s1 / s0;
  const float g = gpred[pix] / s0;
  // The source node d0 of disparity dd grows monotonically with dd (d1 = d0 + 1, or d0 at the last node): the gradient of a
  // node is complete once d0 has moved past it, so two running sums replace the read-modify-write column in LDS (half the LDS:
  // three workgroups per CU instead of one) and every node is written exactly once, straight to G, in a fixed order.
  const long long hw = (long long)d.H * d.W;
  float* Gb = G + (long long)b * d.D4 * hw + (long long)h * d.W + w;
  int cur = 0;            // node held in a0; a1 holds node cur + 1
  float a0 = 0.f, a1 = 0.f;
  for (int dd = 0; dd < d.D; ++dd) {
    int d0, d1;
    float ld;
    src_index(dd, d.sd, d.D4, d0, d1, ld);
    while (cur < d0) {  // flush finished nodes (nodes skipped by a coarse disparity axis get their zero)
      Gb[(long long)cur * hw] = a0;
      a0 = a1;
      a1 = 0.f;
      ++cur;
    }
    const float v = (1.f - ld) * ucol[d0 * NT] + ld * ucol[d1 * NT];
    const float gv = g * __expf(v - m) * ((float)dd - p);
    a0 += (1.f - ld) * gv;
    if (d1 != d0)
      a1 += ld * gv;
    else
      a0 += ld * gv;
  }
  for (; cur < d.D4; ++cur) {
    Gb[(long long)cur * hw] = a0;
    a0 = a1;
    a1 = 0.f;
  }
}

// Backward kernel 2: gL[b][d4][h4][w4] = sum_{h,w} wh(h,h4) * ww(w,w4) * G[b][d4][h][w]
__global__ __launch_bounds__(NT) void head_bwd_gather_kernel(const float* __restrict__ G, float* __restrict__ gL, HDims d) {
  const long long total = (long long)d.B * d.D4 * d.H4 * d.W4;
  const long long idx = (long long)blockIdx.x * NT + threadIdx.x;
  if (idx >= total) return;
  const int w4 = (int)(idx % d.W4);
  const int h4 = (int)((idx / d.W4) % d.H4);
  const long long bd = idx / ((long long)d.W4 * d.H4);  // b*D4 + d4
  // output rows/cols whose source index can fall in (h4-1, h4+1)
  // (one extra row/column of margin against float rounding; membership is decided by src_index below)
  const int hlo = d.sh > 0.f ? max(0, (int)floorf((float)(h4 - 1) / d.sh) - 1) : 0;
  const int hhi = d.sh > 0.f ? min(d.H - 1, (int)ceilf((float)(h4 + 1) / d.sh) + 1) : d.H - 1;
  const int wlo = d.sw > 0.f ? max(0, (int)floorf((float)(w4 - 1) / d.sw) - 1) : 0;
  const int whi = d.sw > 0.f ? min(d.W - 1, (int)ceilf((float)(w4 + 1) / d.sw) + 1) : d.W - 1;
  const float* Gp = G + bd * (long long)d.H * d.W;
  float sum = 0.f;
  constexpr int MAXW = 16;
  const bool small = whi - wlo + 1 <= MAXW;  // x4 upsampling: 10-12 columns; weights of the columns computed once, not per row
  float wwv[MAXW];
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int w = wlo + i;
    int w0, w1;
    float lw;
    src_index(min(w, d.W - 1), d.sw, d.W4, w0, w1, lw);
    wwv[i] = (small && w <= whi) ? (w0 == w4 ? 1.f - lw : 0.f) + (w1 == w4 ? lw : 0.f) : 0.f;
  }
  for (int h = hlo; h <= hhi; ++h) {
    int h0, h1;
    float lh;
    src_index(h, d.sh, d.H4, h0, h1, lh);
    const float wh = (h0 == h4 ? 1.f - lh : 0.f) + (h1 == h4 ? lh : 0.f);
    if (wh == 0.f) continue;
    float rs = 0.f;
    if (small) {
      const float* row = Gp + (long long)h * d.W + wlo;
#pragma unroll
      for (int i = 0; i < MAXW; ++i)
        if (wwv[i] != 0.f) rs += wwv[i] * row[i];  // (unconditional loads of all 16 columns measured slower: 0.63 vs 0.50 ms)
    } else {
      for (int w = wlo; w <= whi; ++w) {
        int w0, w1;
        float lw;
        src_index(w, d.sw, d.W4, w0, w1, lw);
        const float ww = (w0 == w4 ? 1.f - lw : 0.f) + (w1 == w4 ? lw : 0.f);
        if (ww != 0.f) rs += ww * Gp[(long long)h * d.W + w];
      }
    }
    sum += wh * rs;
  }
  gL[idx] = sum;
}

int make_hdims(HDims& d, int B, int D4, int H4, int W4, int D, int H, int W, const char* who) {
  MODE_REQUIRE(B >= 0 && D4 > 0 && H4 > 0 && W4 > 0 && D > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE((size_t)D4 * NT * sizeof(float) <= 160 * 1024, MODE_ERR_UNSUPPORTED, "%s: D4 = %d too large for LDS", who, D4);
  d.B = B; d.D4 = D4; d.H4 = H4; d.W4 = W4; d.D = D; d.H = H; d.W = W;
  d.sd = D > 1 ? (float)(D4 - 1) / (float)(D - 1) : 0.f;
  d.sh = H > 1 ? (float)(H4 - 1) / (float)(H - 1) : 0.f;
  d.sw = W > 1 ? (float)(W4 - 1) / (float)(W - 1) : 0.f;
  return MODE_OK;
}

}  // namespace

extern "C" int mode_head_fwd(const float* logits, float* pred, float* conf, int B, int D4, int H4, int W4, int D, int H, int W,
                             mode_stream_t stream) {
  HDims d;
  int rc = make_hdims(d, B, D4, H4, W4, D, H, W, "mode_head_fwd");
  if (rc != MODE_OK) return rc;
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(logits && pred, MODE_ERR_BAD_ARG, "mode_head_fwd: null pointer");
  const size_t lds = (size_t)D4 * NT * sizeof(float);
  rc = mode::allow_lds(head_fwd_kernel, lds, "mode_head_fwd");
  if (rc != MODE_OK) return rc;
  const long long npix = (long long)B * H * W;
  hipLaunchKernelGGL(head_fwd_kernel, dim3(mode::cdiv(npix, NT)), dim3(NT), lds, mode::as_stream(stream), logits, pred, conf, d);
  return mode::check_launch("mode_head_fwd");
}

extern "C" size_t mode_head_bwd_workspace_bytes(int B, int D4, int H, int W) {
  if (B <= 0 || D4 <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)B * D4 * H * W * sizeof(float);
}

extern "C" int mode_head_bwd(const float* logits, const float* gpred, float* glogits, float* workspace, int B, int D4, int H4,
                             int W4, int D, int H, int W, mode_stream_t stream) {
  HDims d;
  int rc = make_hdims(d, B, D4, H4, W4, D, H, W, "mode_head_bwd");
  if (rc != MODE_OK) return rc;
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(logits && gpred && glogits, MODE_ERR_BAD_ARG, "mode_head_bwd: null pointer");
  MODE_REQUIRE(workspace, MODE_ERR_WORKSPACE, "mode_head_bwd: workspace required");
  const size_t lds = (size_t)D4 * NT * sizeof(float);
  rc = mode::allow_lds(head_bwd_pix_kernel, lds, "mode_head_bwd");
  if (rc != MODE_OK) return rc;
  hipStream_t st = mode::as_stream(stream);
  const long long npix = (long long)B * H * W;
  hipLaunchKernelGGL(head_bwd_pix_kernel, dim3(mode::cdiv(npix, NT)), dim3(NT), lds, st, logits, gpred, workspace, d);
  rc = mode::check_launch("mode_head_bwd(pixels)");
  if (rc != MODE_OK) return rc;
  const long long n = (long long)B * D4 * H4 * W4;
  hipLaunchKernelGGL(head_bwd_gather_kernel, dim3(mode::cdiv(n, NT)), dim3(NT), 0, st, workspace, glogits, d);
  return mode::check_launch("mode_head_bwd(gather)");
}
