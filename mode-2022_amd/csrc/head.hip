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

#include <algorithm>

namespace {

constexpr int NT = 256;

struct HDims {
  int B, D4, H4, W4, D, H, W;
  float sd, sh, sw;  // align_corners scales (in-1)/(out-1)
};

__host__ __device__ __forceinline__ void src_index(int o, float scale, int in, int& i0, int& i1, float& l1) {
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

// ---------------------------------------------------------------------------------------------------------------------
// Fast forms for D = 4 * D4 with D4 a compile-time constant (the network: D4 = maxdisp / 4; 48 at the benchmark).  The generic
// kernels above spend ~25 instructions per (pixel, disparity): the source-index arithmetic of the align_corners up-sampling, two
// LDS reads, the lerp, the exponential.  With D4 fixed the disparity loop unrolls completely: node index and lerp weight of every
// disparity are CONSTANTS folded by the compiler (same float expressions as src_index, evaluated in IEEE fp32 at compile time),
// the D4 column values stay in registers (no LDS), and a (pixel, disparity) costs lerp (2) + v_exp_f32 + 2 accumulations.
// Measured at B = 2, 1024 x 512 x 192: forward 0.165 -> see DESIGN.md; the 201 M exponentials are ~0.02 ms of it.
constexpr float kLog2e = 1.4426950408889634f;

template <int D4>
struct HeadConst {
  static constexpr int D = 4 * D4;
  static constexpr float sd = D > 1 ? (float)(D4 - 1) / (float)(D - 1) : 0.f;
  // area_pixel_compute_source_index(align_corners=True) for output index dd, as src_index() computes it at run time
  static constexpr int d0(int dd) {
    int i = (int)(sd * (float)dd);
    return i > D4 - 1 ? D4 - 1 : i;
  }
  static constexpr int d1(int dd) { return d0(dd) + (d0(dd) < D4 - 1 ? 1 : 0); }
  static constexpr float ld(int dd) { return sd * (float)dd - (float)d0(dd); }
};

// a[k] = (bilinear column value - max over the column) * log2(e), in registers
template <int D4>
__device__ __forceinline__ void fill_column_regs(const float* __restrict__ Lb, const HDims& d, int h, int w, float (&a)[D4]) {
  int h0, h1, w0, w1;
  float lh, lw;
  src_index(h, d.sh, d.H4, h0, h1, lh);
  src_index(w, d.sw, d.W4, w0, w1, lw);
  const float uh = 1.f - lh, uw = 1.f - lw;
  const unsigned plane = (unsigned)(d.H4 * d.W4);
  const unsigned o00 = h0 * d.W4 + w0, o01 = h0 * d.W4 + w1, o10 = h1 * d.W4 + w0, o11 = h1 * d.W4 + w1;
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < D4; ++k) {
    const float* p = Lb + k * plane;
    const float u = uh * (uw * p[o00] + lw * p[o01]) + lh * (uw * p[o10] + lw * p[o11]);
    a[k] = u;
    m = fmaxf(m, u);
  }
#pragma unroll
  for (int k = 0; k < D4; ++k) a[k] = (a[k] - m) * kLog2e;
}

template <int D4, bool CONF>
__global__ __launch_bounds__(NT) void head_fwd_fast_kernel(const float* __restrict__ L, float* __restrict__ pred, float* __restrict__ conf,
                                                           HDims d) {
  using C = HeadConst<D4>;
  extern __shared__ __attribute__((aligned(16))) float u[];  // CONF only: [D4][NT] column for the three run-time indexed reads
  const long long npix = (long long)d.B * d.H * d.W;
  const long long pix = (long long)blockIdx.x * NT + threadIdx.x;
  if (pix >= npix) return;  // no barriers in this kernel
  const int w = (int)(pix % d.W);
  const int h = (int)((pix / d.W) % d.H);
  const int b = (int)(pix / ((long long)d.W * d.H));
  float a[D4];
  fill_column_regs<D4>(L + (long long)b * D4 * d.H4 * d.W4, d, h, w, a);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int dd = 0; dd < C::D; ++dd) {
    constexpr int dummy = 0;
    (void)dummy;
    const float l = C::ld(dd);
    const float v = (1.f - l) * a[C::d0(dd)] + l * a[C::d1(dd)];
    const float e = __builtin_amdgcn_exp2f(v);
    s0 += e;
    s1 = fmaf(e, (float)dd, s1);
  }
  const float p = s1 / s0;
  pred[pix] = p;
  if (CONF) {
    float* ucol = u + threadIdx.x;
#pragma unroll
    for (int k = 0; k < D4; ++k) ucol[k * NT] = a[k];
    // P(round(p)-1) + P(round(p)) + P(round(p)+1), indices clamped to the border (mode_disparity.py:159-180)
    const float r = rintf(p);
    float c = 0.f;
#pragma unroll
    for (int off = -1; off <= 1; ++off) {
      const int idx = (int)fminf(fmaxf(r + (float)off, 0.f), (float)(C::D - 1));
      int i0, i1;
      float l;
      src_index(idx, d.sd, D4, i0, i1, l);
      c += __builtin_amdgcn_exp2f((1.f - l) * ucol[i0 * NT] + l * ucol[i1 * NT]);
    }
    conf[pix] = c / s0;
  }
}

// Backward, one pass: with e_dd = exp(v_dd - m), A_k = sum_dd w_k(dd) e_dd and B_k = sum_dd w_k(dd) e_dd (dd - c_k) (w_k = the lerp
// weight of node k at disparity dd, c_k = a constant near the node: keeps the products small), the gradient of node k is
//   G_k = g / s0 * (B_k - (p - c_k) A_k),   s0 = sum_k A_k,   p = sum_k (B_k + c_k A_k) / s0
// -- the same sums as gv_dd = g p_dd (dd - pred) scattered to the two nodes of dd, without a second walk over the disparities.
template <int D4>
__global__ __launch_bounds__(NT) void head_bwd_pix_fast_kernel(const float* __restrict__ L, const float* __restrict__ gpred,
                                                               float* __restrict__ G, HDims d) {
  using C = HeadConst<D4>;
  const long long npix = (long long)d.B * d.H * d.W;
  const long long pix = (long long)blockIdx.x * NT + threadIdx.x;
  if (pix >= npix) return;
  const int w = (int)(pix % d.W);
  const int h = (int)((pix / d.W) % d.H);
  const int b = (int)(pix / ((long long)d.W * d.H));
  float a[D4];
  fill_column_regs<D4>(L + (long long)b * D4 * d.H4 * d.W4, d, h, w, a);
  float A[D4], Bc[D4];
#pragma unroll
  for (int k = 0; k < D4; ++k) A[k] = Bc[k] = 0.f;
#pragma unroll
  for (int dd = 0; dd < C::D; ++dd) {
    const float l = C::ld(dd);
    const int k0 = C::d0(dd), k1 = C::d1(dd);
    const float e = __builtin_amdgcn_exp2f((1.f - l) * a[k0] + l * a[k1]);
    // (all four coefficients are compile-time constants: four FMAs per disparity)
    if (k1 != k0) {
      A[k0] = fmaf(1.f - l, e, A[k0]);
      Bc[k0] = fmaf((1.f - l) * (float)(dd - 4 * k0), e, Bc[k0]);
      A[k1] = fmaf(l, e, A[k1]);
      Bc[k1] = fmaf(l * (float)(dd - 4 * k1), e, Bc[k1]);
    } else {  // the last node: both lerp weights land on it
      A[k0] += e;
      Bc[k0] = fmaf((float)(dd - 4 * k0), e, Bc[k0]);
    }
  }
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int k = 0; k < D4; ++k) {
    s0 += A[k];
    s1 += fmaf((float)(4 * k), A[k], Bc[k]);
  }
  const float p = s1 / s0;
  const float g = gpred[pix] / s0;
  const long long hw = (long long)d.H * d.W;
  float* Gb = G + (long long)b * D4 * hw + (long long)h * d.W + w;
#pragma unroll
  for (int k = 0; k < D4; ++k) Gb[(long long)k * hw] = g * (Bc[k] - (p - (float)(4 * k)) * A[k]);
}

// Backward, separable (h, w) transpose of the bilinear up-sampling: rows first (R[bd][h][w4] = sum_w ww(w, w4) G[bd][h][w]), then
// columns (gL[bd][h4][w4] = sum_h wh(h, h4) R[bd][h][w4]): ~10 + ~10 terms per node instead of ~100, both passes coalesced along w4.
__global__ __launch_bounds__(NT) void head_bwd_rows_kernel(const float* __restrict__ G, float* __restrict__ R, HDims d, long long rows) {
  const long long total = rows * d.W4;  // rows = B * D4 * H
  const long long idx = (long long)blockIdx.x * NT + threadIdx.x;
  if (idx >= total) return;
  const int w4 = (int)(idx % d.W4);
  const long long row = idx / d.W4;
  const int wlo = d.sw > 0.f ? max(0, (int)floorf((float)(w4 - 1) / d.sw) - 1) : 0;
  const int whi = d.sw > 0.f ? min(d.W - 1, (int)ceilf((float)(w4 + 1) / d.sw) + 1) : d.W - 1;
  const float* g = G + row * d.W;
  float sum = 0.f;
  for (int w = wlo; w <= whi; ++w) {
    int w0, w1;
    float lw;
    src_index(w, d.sw, d.W4, w0, w1, lw);
    const float ww = (w0 == w4 ? 1.f - lw : 0.f) + (w1 == w4 ? lw : 0.f);
    sum = fmaf(ww, g[w], sum);  // (ww is exactly 0 outside the support: membership is decided by src_index, as in the one-pass kernel)
  }
  R[idx] = sum;
}

__global__ __launch_bounds__(NT) void head_bwd_cols_kernel(const float* __restrict__ R, float* __restrict__ gL, HDims d) {
  const long long total = (long long)d.B * d.D4 * d.H4 * d.W4;
  const long long idx = (long long)blockIdx.x * NT + threadIdx.x;
  if (idx >= total) return;
  const int w4 = (int)(idx % d.W4);
  const int h4 = (int)((idx / d.W4) % d.H4);
  const long long bd = idx / ((long long)d.W4 * d.H4);
  const int hlo = d.sh > 0.f ? max(0, (int)floorf((float)(h4 - 1) / d.sh) - 1) : 0;
  const int hhi = d.sh > 0.f ? min(d.H - 1, (int)ceilf((float)(h4 + 1) / d.sh) + 1) : d.H - 1;
  const float* r = R + bd * (long long)d.H * d.W4 + w4;
  float sum = 0.f;
  for (int h = hlo; h <= hhi; ++h) {
    int h0, h1;
    float lh;
    src_index(h, d.sh, d.H4, h0, h1, lh);
    const float wh = (h0 == h4 ? 1.f - lh : 0.f) + (h1 == h4 ? lh : 0.f);
    sum = fmaf(wh, r[(long long)h * d.W4], sum);
  }
  gL[idx] = sum;
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward with the per-pixel pass and the ROW half of the separable (h, w) transpose in ONE kernel (round 5).  The two-kernel form wrote
// G = (B, D4, H, W) -- 201 MB per head at the benchmark shape -- and read it back to form the row sums R = (B, D4, H, W4): 0.157 ms of a
// 0.26 ms head backward were that round trip.  Here a block owns ONE image row (W <= 512 threads, one per pixel), keeps the 48 node
// gradients of every pixel in registers (a block is at most 512 threads: two waves per SIMD), and passes them through LDS 16 nodes at a time so that the threads can re-read them along w:
// R[k][w4] = sum_w ww(w, w4) G[k][w] over the <= 12 pixels of a node's support.  G never exists; the column half (head_bwd_cols_kernel)
// is unchanged.  LOSS: the upstream gradient is not read from memory but formed from the forward's own prediction and the ground truth --
// the masked smooth-L1 of train_disparity.py:151-158: g = weight * [gt == gt] * clamp(pred - gt, -1, 1) * (*scale) -- so that the
// loss chain between the head and the optimizer needs no elementwise launches.
constexpr int HR_KC = 16;   // nodes per LDS pass
constexpr int HR_MAXS = 12;  // pixels in the support of a low-resolution node (x4 up-sampling: 8-9)

template <int D4, bool LOSS>
__global__ __launch_bounds__(512) void head_bwd_pixrows_kernel(const float* __restrict__ L, const float* __restrict__ gpred,
                                                                const float* __restrict__ pred, const float* __restrict__ gt, float weight,
                                                                const float* __restrict__ scale, float* __restrict__ R, HDims d) {
  using C = HeadConst<D4>;
  extern __shared__ __attribute__((aligned(16))) float gl[];  // [HR_KC][pitch], element w of a row at w + (w >> 2): the stride-4 reads of
  const int pitch = d.W + (d.W >> 2) + 1;                      // neighbouring nodes then fall on different banks
  const int w = threadIdx.x;
  const int h = blockIdx.x % d.H, b = blockIdx.x / d.H;
  const bool live = w < d.W;
  float A[D4], Bc[D4];
  float gscale = 0.f, p = 0.f;
  if (live) {
    float a[D4];
    fill_column_regs<D4>(L + (long long)b * D4 * d.H4 * d.W4, d, h, w, a);
#pragma unroll
    for (int k = 0; k < D4; ++k) A[k] = Bc[k] = 0.f;
#pragma unroll
    for (int dd = 0; dd < C::D; ++dd) {
      const float l = C::ld(dd);
      const int k0 = C::d0(dd), k1 = C::d1(dd);
      const float e = __builtin_amdgcn_exp2f((1.f - l) * a[k0] + l * a[k1]);
      if (k1 != k0) {
        A[k0] = fmaf(1.f - l, e, A[k0]);
        Bc[k0] = fmaf((1.f - l) * (float)(dd - 4 * k0), e, Bc[k0]);
        A[k1] = fmaf(l, e, A[k1]);
        Bc[k1] = fmaf(l * (float)(dd - 4 * k1), e, Bc[k1]);
      } else {
        A[k0] += e;
        Bc[k0] = fmaf((float)(dd - 4 * k0), e, Bc[k0]);
      }
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < D4; ++k) {
      s0 += A[k];
      s1 += fmaf((float)(4 * k), A[k], Bc[k]);
    }
    p = s1 / s0;
    const long long pix = ((long long)b * d.H + h) * d.W + w;
    float g;
    if (LOSS) {
      const float t = gt[pix];
      const float df = pred[pix] - t;  // the forward's own value: the gradient of the loss that was reported
      g = (t == t) ? weight * fminf(fmaxf(df, -1.f), 1.f) * scale[0] : 0.f;  // NaN ground truth = masked out (train_disparity.py:195)
    } else {
      g = gpred[pix];
    }
    gscale = g / s0;
  }
  // this thread's output nodes: w4 = threadIdx.x % W4 for the node rows kq, kq + nkq, ... of a pass (host: blockDim.x % W4 == 0)
  const int w4 = threadIdx.x % d.W4, kq = threadIdx.x / d.W4, nkq = blockDim.x / d.W4;
  const int wlo = d.sw > 0.f ? max(0, (int)floorf((float)(w4 - 1) / d.sw) - 1) : 0;
  const int whi = d.sw > 0.f ? min(d.W - 1, (int)ceilf((float)(w4 + 1) / d.sw) + 1) : d.W - 1;
  float ww[HR_MAXS];
#pragma unroll
  for (int i = 0; i < HR_MAXS; ++i) {
    const int wq = min(wlo + i, d.W - 1);
    int w0, w1;
    float lw;
    src_index(wq, d.sw, d.W4, w0, w1, lw);
    ww[i] = (wlo + i <= whi) ? (w0 == w4 ? 1.f - lw : 0.f) + (w1 == w4 ? lw : 0.f) : 0.f;  // exactly 0 outside the support
  }
  float* Rb = R + (((long long)b * D4) * d.H + h) * d.W4 + w4;
  const long long rplane = (long long)d.H * d.W4;
#pragma unroll
  for (int k0 = 0; k0 < D4; k0 += HR_KC) {
    if (live) {
#pragma unroll
      for (int kk = 0; kk < HR_KC; ++kk)
        if (k0 + kk < D4) gl[kk * pitch + w + (w >> 2)] = gscale * (Bc[k0 + kk] - (p - (float)(4 * (k0 + kk))) * A[k0 + kk]);
    }
    __syncthreads();
    for (int kk = kq; kk < HR_KC && k0 + kk < D4; kk += nkq) {
      const float* row = gl + kk * pitch;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < HR_MAXS; ++i) {
        const int wq = min(wlo + i, d.W - 1);
        sum = fmaf(ww[i], row[wq + (wq >> 2)], sum);
      }
      Rb[(long long)(k0 + kk) * rplane] = sum;
    }
    __syncthreads();
  }
}

// Masked smooth-L1 of up to three predictions against one ground truth (train_disparity.py:151-158):
//   out[0] = (*scale) * sum_i weight_i * sum_pix [gt == gt] * smooth_l1(pred_i - gt),   smooth_l1(x) = 0.5 x^2 (|x| < 1), |x| - 0.5 otherwise
// two launches with a fixed summation order (per-thread runs, a block tree, then one block over the block sums in double).
__global__ __launch_bounds__(NT) void smooth_l1_partial_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                               const float* __restrict__ p2, const float* __restrict__ gt, float w0, float w1,
                                                               float w2, long long n, float* __restrict__ partial) {
  __shared__ float sh[NT / 64];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
    const float t = gt[i];
    if (t == t) {
      auto sl1 = [&](float v) {
        const float ax = fabsf(v - t);
        return ax < 1.f ? 0.5f * ax * ax : ax - 0.5f;
      };
      float v = w0 * sl1(p0[i]);
      if (p1) v += w1 * sl1(p1[i]);
      if (p2) v += w2 * sl1(p2[i]);
      s += v;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = sh[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) v += sh[i];
    partial[blockIdx.x] = v;
  }
}

__global__ __launch_bounds__(NT) void smooth_l1_final_kernel(const float* __restrict__ partial, int nblocks, const float* __restrict__ scale,
                                                             float* __restrict__ out) {
  __shared__ double sh[NT / 64];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += NT) s += (double)partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = sh[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) v += sh[i];
    out[0] = (float)(v * (double)scale[0]);
  }
}

template <int D4>
int launch_fwd_fast(const float* logits, float* pred, float* conf, const HDims& d, hipStream_t st) {
  const long long npix = (long long)d.B * d.H * d.W;
  if (conf) {
    const size_t lds = (size_t)D4 * NT * sizeof(float);
    int rc = mode::allow_lds(head_fwd_fast_kernel<D4, true>, lds, "mode_head_fwd");
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL((head_fwd_fast_kernel<D4, true>), dim3(mode::cdiv(npix, NT)), dim3(NT), lds, st, logits, pred, conf, d);
  } else {
    hipLaunchKernelGGL((head_fwd_fast_kernel<D4, false>), dim3(mode::cdiv(npix, NT)), dim3(NT), 0, st, logits, pred, conf, d);
  }
  return mode::check_launch("mode_head_fwd");
}

template <int D4>
int launch_bwd_fast(const float* logits, const float* gpred, float* ws, const HDims& d, hipStream_t st) {
  const long long npix = (long long)d.B * d.H * d.W;
  hipLaunchKernelGGL(head_bwd_pix_fast_kernel<D4>, dim3(mode::cdiv(npix, NT)), dim3(NT), 0, st, logits, gpred, ws, d);
  return mode::check_launch("mode_head_bwd(pixels)");
}

// one block per image row: W <= 512 pixels (8 waves: the 48-node kernel needs 2 waves per SIMD = 256 registers; wider rows take the
// two-kernel form), a whole number of node rows per pass of the block's threads
// ... and every pixel that contributes to a low-resolution node along w lies within the HR_MAXS pixels from the node's `wlo` that the
// kernel sums (the same wlo and the same src_index, in the same float arithmetic).  At the x4 up-sampling of the model a node's support
// is 8-9 pixels; at a wider ratio (W = 512 over W4 = 64: ~18) the kernel would silently drop contributions -- such shapes
// take the two-kernel form, whose row kernel loops over the whole support.
bool pixrows_support_fits(const HDims& d) {
  if (!(d.sw > 0.f)) return d.W <= HR_MAXS;
  for (int w = 0; w < d.W; ++w) {
    int w0, w1;
    float lw;
    src_index(w, d.sw, d.W4, w0, w1, lw);
    const int nodes[2] = {w0, w1};
    const float wt[2] = {1.f - lw, lw};
    for (int j = 0; j < 2; ++j) {
      if (wt[j] == 0.f) continue;
      const int wlo = std::max(0, (int)floorf((float)(nodes[j] - 1) / d.sw) - 1);
      if (w < wlo || w - wlo >= HR_MAXS) return false;
    }
  }
  return true;
}
bool pixrows_fits(const HDims& d) {
  const int nt = ((d.W + 63) / 64) * 64;
  return d.W <= 512 && d.W4 <= d.W && nt % d.W4 == 0 && (long long)d.B * d.H < (1ll << 31) &&
         (size_t)HR_KC * (d.W + (d.W >> 2) + 1) * sizeof(float) <= 64 * 1024 && pixrows_support_fits(d);
}

template <int D4, bool LOSS>
int launch_bwd_pixrows(const float* logits, const float* gpred, const float* pred, const float* gt, float weight, const float* scale,
                       float* R, const HDims& d, hipStream_t st) {
  const int nt = ((d.W + 63) / 64) * 64;
  const size_t lds = (size_t)HR_KC * (d.W + (d.W >> 2) + 1) * sizeof(float);
  hipLaunchKernelGGL((head_bwd_pixrows_kernel<D4, LOSS>), dim3(d.B * d.H), dim3(nt), lds, st, logits, gpred, pred, gt, weight, scale, R, d);
  return mode::check_launch("mode_head_bwd(pixels + rows)");
}

// the compile-time instantiations: D4 = maxdisp / 4 of the configurations in use (16 ... 256 disparities)
#define MODE_HEAD_FAST_D4(X) X(4) X(8) X(12) X(16) X(48) X(64)
bool head_fast(const HDims& d) {
  if (d.D != 4 * d.D4) return false;
#define X(N) if (d.D4 == N) return true;
  MODE_HEAD_FAST_D4(X)
#undef X
  return false;
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
  if (head_fast(d)) {
#define X(N) if (D4 == N) return launch_fwd_fast<N>(logits, pred, conf, d, mode::as_stream(stream));
    MODE_HEAD_FAST_D4(X)
#undef X
  }
  const size_t lds = (size_t)D4 * NT * sizeof(float);
  rc = mode::allow_lds(head_fwd_kernel, lds, "mode_head_fwd");
  if (rc != MODE_OK) return rc;
  const long long npix = (long long)B * H * W;
  hipLaunchKernelGGL(head_fwd_kernel, dim3(mode::cdiv(npix, NT)), dim3(NT), lds, mode::as_stream(stream), logits, pred, conf, d);
  return mode::check_launch("mode_head_fwd");
}

// G (B, D4, H, W) and, behind it, the row sums R (B, D4, H, W4 <= W) of the separable gather
extern "C" size_t mode_head_bwd_workspace_bytes(int B, int D4, int H, int W) {
  if (B <= 0 || D4 <= 0 || H <= 0 || W <= 0) return 0;
  return 2 * (size_t)B * D4 * H * W * sizeof(float);
}

extern "C" int mode_head_bwd(const float* logits, const float* gpred, float* glogits, float* workspace, int B, int D4, int H4,
                             int W4, int D, int H, int W, mode_stream_t stream) {
  HDims d;
  int rc = make_hdims(d, B, D4, H4, W4, D, H, W, "mode_head_bwd");
  if (rc != MODE_OK) return rc;
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(logits && gpred && glogits, MODE_ERR_BAD_ARG, "mode_head_bwd: null pointer");
  MODE_REQUIRE(workspace, MODE_ERR_WORKSPACE, "mode_head_bwd: workspace required");
  hipStream_t st = mode::as_stream(stream);
  const long long npix = (long long)B * H * W;
  bool fast = false;
  if (head_fast(d) && pixrows_fits(d)) {  // per-pixel pass and row sums in one kernel, then the column sums
    float* R = workspace;
#define X(N) if (D4 == N) rc = launch_bwd_pixrows<N, false>(logits, gpred, nullptr, nullptr, 0.f, nullptr, R, d, st);
    MODE_HEAD_FAST_D4(X)
#undef X
    if (rc != MODE_OK) return rc;
    const long long n = (long long)B * D4 * H4 * W4;
    hipLaunchKernelGGL(head_bwd_cols_kernel, dim3(mode::cdiv(n, NT)), dim3(NT), 0, st, R, glogits, d);
    return mode::check_launch("mode_head_bwd(columns)");
  }
  if (head_fast(d)) {
#define X(N) if (D4 == N) { rc = launch_bwd_fast<N>(logits, gpred, workspace, d, st); fast = true; }
    MODE_HEAD_FAST_D4(X)
#undef X
  }
  if (!fast) {
    const size_t lds = (size_t)D4 * NT * sizeof(float);
    rc = mode::allow_lds(head_bwd_pix_kernel, lds, "mode_head_bwd");
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL(head_bwd_pix_kernel, dim3(mode::cdiv(npix, NT)), dim3(NT), lds, st, logits, gpred, workspace, d);
    rc = mode::check_launch("mode_head_bwd(pixels)");
  }
  if (rc != MODE_OK) return rc;
  if (W4 <= W) {  // separable gather: rows, then columns
    float* R = workspace + (size_t)B * D4 * H * W;
    const long long rows = (long long)B * D4 * H;
    hipLaunchKernelGGL(head_bwd_rows_kernel, dim3(mode::cdiv(rows * W4, NT)), dim3(NT), 0, st, workspace, R, d, rows);
    const long long n = (long long)B * D4 * H4 * W4;
    hipLaunchKernelGGL(head_bwd_cols_kernel, dim3(mode::cdiv(n, NT)), dim3(NT), 0, st, R, glogits, d);
    return mode::check_launch("mode_head_bwd(gather)");
  }
  const long long n = (long long)B * D4 * H4 * W4;
  hipLaunchKernelGGL(head_bwd_gather_kernel, dim3(mode::cdiv(n, NT)), dim3(NT), 0, st, workspace, glogits, d);
  return mode::check_launch("mode_head_bwd(gather)");
}

// ---------------------------------------------------------------------------------------------------------------------
// The loss of the training step next to the head (train_disparity.py:151-158): see include/mode_hip.h.
extern "C" size_t mode_smooth_l1_workspace_bytes(long long n) {
  if (n <= 0) return 0;
  return (size_t)std::min<long long>((n + NT - 1) / NT, 4LL * kNumCU) * sizeof(float);
}

extern "C" int mode_smooth_l1_masked(const float* pred0, const float* pred1, const float* pred2, const float* gt, float w0, float w1,
                                     float w2, const float* scale, float* out, float* workspace, long long n, mode_stream_t stream) {
  MODE_REQUIRE(n > 0, MODE_ERR_BAD_ARG, "mode_smooth_l1_masked: non-positive size");
  MODE_REQUIRE(pred0 && gt && scale && out && workspace && (pred1 || !pred2), MODE_ERR_BAD_ARG, "mode_smooth_l1_masked: null pointer");
  const int blocks = (int)(mode_smooth_l1_workspace_bytes(n) / sizeof(float));
  hipStream_t st = mode::as_stream(stream);
  hipLaunchKernelGGL(smooth_l1_partial_kernel, dim3(blocks), dim3(NT), 0, st, pred0, pred1, pred2, gt, w0, w1, w2, n, workspace);
  hipLaunchKernelGGL(smooth_l1_final_kernel, dim3(1), dim3(NT), 0, st, workspace, blocks, scale, out);
  return mode::check_launch("mode_smooth_l1_masked");
}

extern "C" int mode_head_bwd_loss(const float* logits, const float* pred, const float* gt, float weight, const float* scale, float* glogits,
                                  float* workspace, int B, int D4, int H4, int W4, int D, int H, int W, mode_stream_t stream) {
  HDims d;
  int rc = make_hdims(d, B, D4, H4, W4, D, H, W, "mode_head_bwd_loss");
  if (rc != MODE_OK) return rc;
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(logits && pred && gt && scale && glogits, MODE_ERR_BAD_ARG, "mode_head_bwd_loss: null pointer");
  MODE_REQUIRE(workspace, MODE_ERR_WORKSPACE, "mode_head_bwd_loss: workspace required");
  MODE_REQUIRE(head_fast(d) && pixrows_fits(d), MODE_ERR_UNSUPPORTED,
               "mode_head_bwd_loss: needs D = 4 * D4 with D4 in {4, 8, 12, 16, 48, 64}, W <= 512 and W4 dividing the row's thread count "
               "(mode_head_loss_supported); form the gradient of the loss and call mode_head_bwd otherwise");
  hipStream_t st = mode::as_stream(stream);
  float* R = workspace;
#define X(N) if (D4 == N) rc = launch_bwd_pixrows<N, true>(logits, nullptr, pred, gt, weight, scale, R, d, st);
  MODE_HEAD_FAST_D4(X)
#undef X
  if (rc != MODE_OK) return rc;
  const long long n = (long long)B * D4 * H4 * W4;
  hipLaunchKernelGGL(head_bwd_cols_kernel, dim3(mode::cdiv(n, NT)), dim3(NT), 0, st, R, glogits, d);
  return mode::check_launch("mode_head_bwd_loss(columns)");
}

extern "C" int mode_head_loss_supported(int B, int D4, int H4, int W4, int D, int H, int W) {
  HDims d;
  if (B <= 0 || make_hdims(d, B, D4, H4, W4, D, H, W, "mode_head_loss_supported") != MODE_OK) return 0;
  return head_fast(d) && pixrows_fits(d) ? 1 : 0;
}
