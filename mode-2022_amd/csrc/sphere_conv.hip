// Spherical convolution for gfx950 (MI355X): fused bilinear gather + fp32-MFMA contraction.
//
// Reference being replaced (paths relative to the upstream repo):
//   models/basic/spherical_conv/src/sphere_conv_cuda_kernel.cu:195-262  im2col gather (K1)
//   models/basic/spherical_conv/src/sphere_conv_cuda_kernel.cu:293-356  col2im atomic scatter (K2)
//   models/basic/spherical_conv/src/sphere_conv_cuda.cpp:129-336        per-sample loops + cuBLAS addmm_ (K3-K5)
// The reference materialises a (Ci*Kh*Kw) x (Ho*Wo) column buffer in HBM per sample and per layer (151 MB at
// 128 channels, 256x128) and runs SGEMM on it.  Here the column tile only ever exists in LDS:
//
//   forward      y[o, p]      = sum_kk W[o, kk] * col[kk, p]            D[i=o ][j=p ]   A = packed W (global/L2)
//   bwd-data     gcol[kk, p]  = sum_o  W[o, kk] * gy[o, p]  -> scatter   D[i=kk][j=p ]   A = packed W^T, B = gy tile (LDS)
//   bwd-weight   gW[o, kk]    = sum_p  gy[o, p] * col[kk, p]             D[i=o ][j=kk]   A = gy tile, B = col tile (LDS)
//
// with kk = c*Kh*Kw + tap, p = linear output pixel, all on v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s peak).
// Per output pixel and tap the sampling record (clamped corner offset + 4 bilinear weights with the per-corner
// zero padding of cu:96-107 folded in) is computed once per tile into LDS and shared by every channel.
//
// MFMA operand maps (32x32x2 f32): lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; it receives
// D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31] in accumulator register r (r = 0..15).
#include <cstring>

#include "common.h"
#include "sphere_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int P = 64;          // output pixels per tile (2 MFMA column tiles)
constexpr int CCH = 8;         // input channels per forward K-chunk
constexpr int NTHREADS = 256;  // 4 waves: one 32-row MFMA tile each

struct Dims {
  int B, Ci, H, W, Co, KK, sH, sW, Ho, Wo, G;
  int Cig, Cog;   // channels per group
  int npix;       // Ho*Wo
  int tps;        // pixel tiles per sample
  int MT;         // 32-row tiles over Cog
  int NCHUNK;     // forward K-chunks (CCH channels each)
  int CB;         // channels per 128-row block (bwd): floor(128 / KK)
  int NB;         // such blocks over Cig
  int KSQ;        // k-step quads over Cog (bwd-data): ceil(Cog / 8)
  int accumulate; // gather-form bwd-data: add to gx (1, the reference op's contract) or overwrite it (0)
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------
// Sampling records.  tap_off: bits 0-29 offset of the (clamped) top-left corner, bit 30 = step to the right
// corner (0/1), bit 31 = step to the lower row (0/1).  tap_w = weights of (top-left, top-right, bottom-left,
// bottom-right) with invalid corners zeroed (cu:96-107) and the whole tap zeroed outside (-1,H)x(-1,W) (cu:246).
// With a pixel selection (pixmap, nsel) the tile holds the listed output pixels pixmap[pix0 .. pix0+P) instead of the
// consecutive ones.
__device__ __forceinline__ void compute_tapinfo(const float* __restrict__ pos, const Dims& d, int pix0,
                                                unsigned* tap_off, float4* tap_w, const int* __restrict__ pixmap = nullptr,
                                                int nsel = 0) {
  const int HW = d.H * d.W;
  for (int item = threadIdx.x; item < d.KK * P; item += NTHREADS) {
    const int k = item / P;
    const int p = item % P;
    int pix = pix0 + p;
    if (pixmap) pix = pix < nsel ? pixmap[pix] : d.npix;
    unsigned off = 0;
    float4 wt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pix < d.npix) {
      const int ho = pix / d.Wo;
      const int wo = pix - ho * d.Wo;
      const int idx = (ho * d.sH) * d.W + wo * d.sW;
      const float h = pos[(2 * k) * HW + idx];
      const float w = pos[(2 * k + 1) * HW + idx];
      if (h > -1.f && w > -1.f && h < (float)d.H && w < (float)d.W) {
        const float hf = floorf(h), wf = floorf(w);
        const int hl = (int)hf, wl = (int)wf;
        const int hh = hl + 1, wh = wl + 1;
        const float lh = h - hf, lw = w - wf;
        const float uh = 1.f - lh, uw = 1.f - lw;
        wt.x = (hl >= 0 && wl >= 0) ? uh * uw : 0.f;
        wt.y = (hl >= 0 && wh <= d.W - 1) ? uh * lw : 0.f;
        wt.z = (hh <= d.H - 1 && wl >= 0) ? lh * uw : 0.f;
        wt.w = (hh <= d.H - 1 && wh <= d.W - 1) ? lh * lw : 0.f;
        const int hlc = max(hl, 0), hhc = min(hh, d.H - 1);
        const int wlc = max(wl, 0), whc = min(wh, d.W - 1);
        off = (unsigned)(hlc * d.W + wlc) | ((unsigned)(whc - wlc) << 30) | ((unsigned)(hhc - hlc) << 31);
      }
    }
    tap_off[item] = off;
    tap_w[item] = wt;
  }
}

__device__ __forceinline__ float sample(const float* __restrict__ xc, unsigned po, const float4& tw, int W) {
  const int off = (int)(po & 0x3fffffffu);
  const int dw = (int)((po >> 30) & 1u);
  const int dh = (po >> 31) ? W : 0;
  // same operand order as cu:111
  return tw.x * xc[off] + tw.y * xc[off + dw] + tw.z * xc[off + dh] + tw.w * xc[off + dh + dw];
}

// ---------------------------------------------------------------------------------------------------------
// Weight packing into MFMA A-fragment order (one float4 = the lane's operands of 4 consecutive k-steps).
//
// forward:  wp[((g*MT + mt)*NCHUNK + ch)*KK + quad][lane][j] = W[g*Cog + mt*32 + (lane&31)][ch*8 + kl/KK][kl%KK],
//           kl = 2*(quad*4 + j) + (lane>>5), zero outside the real weight.
//           fold != 0: output channel co is scaled by the folded BatchNorm scale of `bn`; block 0 writes the shifts to
//           wp[total + channel] (common.h).
__global__ void pack_w_fwd(const float* __restrict__ w, float* __restrict__ wp, Dims d, int fold, mode_bn_epilogue bn) {
  const long long total = (long long)d.G * d.MT * d.NCHUNK * d.KK * 64 * 4;
  if (fold && blockIdx.x == 0)
    for (int o = threadIdx.x; o < d.Co; o += blockDim.x) wp[total + o] = fold_shift(bn, o);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int quad = (int)(r % d.KK);
    r /= d.KK;
    const int ch = (int)(r % d.NCHUNK);
    r /= d.NCHUNK;
    const int mt = (int)(r % d.MT);
    const int g = (int)(r / d.MT);
    const int kl = 2 * (quad * 4 + j) + (lane >> 5);
    const int c = ch * CCH + kl / d.KK;
    const int tap = kl % d.KK;
    const int co = mt * 32 + (lane & 31);
    float v = 0.f;
    if (co < d.Cog && c < d.Cig) {
      v = w[((long long)(g * d.Cog + co) * d.Cig + c) * d.KK + tap];
      if (fold) v *= fold_scale(bn, g * d.Cog + co);
    }
    wp[idx] = v;
  }
}

// bwd-data: wp[(((g*NB + nb)*4 + mtl)*KSQ + ks4)][lane][j] = W[g*Cog + co][nb*CB + rr/KK][rr%KK],
//           rr = mtl*32 + (lane&31) (< CB*KK), co = 2*(ks4*4 + j) + (lane>>5).
__global__ void pack_w_bwd(const float* __restrict__ w, float* __restrict__ wp, Dims d) {
  const long long total = (long long)d.G * d.NB * 4 * d.KSQ * 64 * 4;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int ks4 = (int)(r % d.KSQ);
    r /= d.KSQ;
    const int mtl = (int)(r & 3);
    r >>= 2;
    const int nb = (int)(r % d.NB);
    const int g = (int)(r / d.NB);
    const int rr = mtl * 32 + (lane & 31);
    const int c = nb * d.CB + rr / d.KK;
    const int tap = rr % d.KK;
    const int co = 2 * (ks4 * 4 + j) + (lane >> 5);
    float v = 0.f;
    if (rr < d.CB * d.KK && c < d.Cig && co < d.Cog) v = w[((long long)(g * d.Cog + co) * d.Cig + c) * d.KK + tap];
    wp[idx] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Forward.  grid = (B*tps, ceil(MT/4), G); LDS = KK*P*(16+4) + 2 * CCH*KK*P*4 bytes.
// KT = compile-time tap count (9 for the 3x3 kernels of the network) enables the software pipeline: the 4*2*KT corner
// loads of the NEXT chunk are issued into registers before the MFMA phase of the current chunk and combined / written to
// the other LDS buffer after it, so the gather latency hides under the MFMAs.  KT = 0: generic tap count, no pipelining.
template <int KT, bool EPI>
__global__ __launch_bounds__(NTHREADS) void sphere_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                               const float4* __restrict__ wp, float* __restrict__ y,
                                                               Dims dd, Epi epi) {
  Dims d = dd;
  if (KT > 0) d.KK = KT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* tap_w = reinterpret_cast<float4*>(smem);
  unsigned* tap_off = reinterpret_cast<unsigned*>(smem + (size_t)d.KK * P * 16);
  float* colbuf = reinterpret_cast<float*>(smem + (size_t)d.KK * P * 20);
  const int rows = CCH * d.KK;  // column-tile rows per chunk

  const int tile = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring pixel tiles (overlapping gather footprints) on one XCD
  const int b = tile / d.tps;
  const int pix0 = (tile - b * d.tps) * P;
  const int g = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int mt = blockIdx.y * 4 + wave;
  const bool active = mt < d.MT;

  compute_tapinfo(pos, d, pix0, tap_off, tap_w);
  __syncthreads();

  const int p = tid & (P - 1);
  const int q = tid / P;  // 0..3 -> channels 2q, 2q+1 of the chunk
  const long long HW = (long long)d.H * d.W;
  const float* xg = x + ((long long)b * d.Ci + (long long)g * d.Cig) * HW;

  auto produce = [&](int ch, float* buf) {
    for (int k = 0; k < d.KK; ++k) {
      const unsigned po = tap_off[k * P + p];
      const float4 tw = tap_w[k * P + p];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int cl = q * 2 + cc;
        const int c = ch * CCH + cl;
        float v = 0.f;
        if (c < d.Cig) v = sample(xg + c * HW, po, tw, d.W);
        buf[(cl * d.KK + k) * P + p] = v;
      }
    }
  };

  // pipelined form (KT > 0): registers for the 2 channels x KT taps x 4 corners of the next chunk
  constexpr int NLV = KT > 0 ? KT * 8 : 1;
  float lv[NLV];
  auto issue = [&](int ch) {
#pragma unroll
    for (int k = 0; k < (KT > 0 ? KT : 0); ++k) {
      const unsigned po = tap_off[k * P + p];
      const int off = (int)(po & 0x3fffffffu);
      const int dw = (int)((po >> 30) & 1u);
      const int dh = (po >> 31) ? d.W : 0;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int c = ch * CCH + q * 2 + cc;
        const float* xc = xg + (c < d.Cig ? c : 0) * HW;
        lv[(k * 2 + cc) * 4 + 0] = xc[off];
        lv[(k * 2 + cc) * 4 + 1] = xc[off + dw];
        lv[(k * 2 + cc) * 4 + 2] = xc[off + dh];
        lv[(k * 2 + cc) * 4 + 3] = xc[off + dh + dw];
      }
    }
  };
  auto finish = [&](int ch, float* buf) {
#pragma unroll
    for (int k = 0; k < (KT > 0 ? KT : 0); ++k) {
      const float4 tw = tap_w[k * P + p];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int cl = q * 2 + cc;
        const float* l4 = lv + (k * 2 + cc) * 4;
        const float v = tw.x * l4[0] + tw.y * l4[1] + tw.z * l4[2] + tw.w * l4[3];
        buf[(cl * KT + k) * P + p] = (ch * CCH + cl < d.Cig) ? v : 0.f;
      }
    }
  };

  f32x16 acc0 = {0}, acc1 = {0};
  const float4* wpa = wp + ((long long)(g * d.MT + (active ? mt : 0)) * d.NCHUNK) * d.KK * 64 + lane;

  if (KT > 0) {
    issue(0);
    finish(0, colbuf);
  } else {
    produce(0, colbuf);
  }
  __syncthreads();
  for (int ch = 0; ch < d.NCHUNK; ++ch) {
    float* cur = colbuf + (ch & 1) * rows * P;
    float* nxt = colbuf + ((ch + 1) & 1) * rows * P;
    if (ch + 1 < d.NCHUNK) {
      if (KT > 0)
        issue(ch + 1);  // loads fly during the MFMA phase below
      else
        produce(ch + 1, nxt);
    }
    if (active) {
      const float4* wq = wpa + (long long)ch * d.KK * 64;
      const float* bp = cur + (lane >> 5) * P + (lane & 31);
      for (int quad = 0; quad < d.KK; ++quad) {
        const float4 a4 = wq[quad * 64];
        const float* bq = bp + quad * 8 * P;
        acc0 = mfma32(a4.x, bq[0], acc0);
        acc1 = mfma32(a4.x, bq[32], acc1);
        acc0 = mfma32(a4.y, bq[2 * P], acc0);
        acc1 = mfma32(a4.y, bq[2 * P + 32], acc1);
        acc0 = mfma32(a4.z, bq[4 * P], acc0);
        acc1 = mfma32(a4.z, bq[4 * P + 32], acc1);
        acc0 = mfma32(a4.w, bq[6 * P], acc0);
        acc1 = mfma32(a4.w, bq[6 * P + 32], acc1);
      }
    }
    if (KT > 0 && ch + 1 < d.NCHUNK) finish(ch + 1, nxt);  // the other buffer: no barrier needed before writing it
    __syncthreads();
  }

  if (active) {
    float* yb = y + ((long long)b * d.Co + (long long)g * d.Cog) * d.npix;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (co < d.Cog) {
        const int px = pix0 + (lane & 31);
        const long long i0 = (long long)co * d.npix + px;
        if (EPI) {  // eval mode: folded BatchNorm shift (+ residual) (+ ReLU) on the way out
          const int ch = g * d.Cog + co;
          if (px < d.npix) yb[i0] = apply_epi(epi, acc0[r], ch, (yb - y) + i0);
          if (px + 32 < d.npix) yb[i0 + 32] = apply_epi(epi, acc1[r], ch, (yb - y) + i0 + 32);
        } else {
          if (px < d.npix) yb[i0] = acc0[r];
          if (px + 32 < d.npix) yb[i0 + 32] = acc1[r];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Backward w.r.t. the input, gather form (deterministic, no atomics).
//   gx[c, q] = sum_{k, o} W[o, c, k] * colT[(o,k), q],     colT[(o,k), q] = sum_{e in L(k,q)} wt_e * gy[o, p_e]
// where L(k, q) lists the output pixels p whose tap k touches input pixel q with bilinear weight wt (the transpose of the
// sampling table, built once per table by mode_sphere_adjoint_build).  Same structure as the forward kernel: the producer
// fills a column tile in LDS (8 output channels x KK taps x 64 input pixels), the contraction runs on MFMA with
// D[i = c][j = q].  grid = (B*tiles(H*W), ceil(MTc/4), G).  Adds to (d.accumulate) or overwrites gx; each element is owned by one lane.
//
// wp[((g*MTc + mt)*NCHo + ch)*KK + quad][lane][j] = W[g*Cog + ch*8 + kl/KK][mt*32 + (lane&31)][kl%KK], kl = 2*(quad*4+j) + (lane>>5)
__global__ void pack_w_adj(const float* __restrict__ w, float* __restrict__ wp, Dims d, int MTc, int NCHo) {
  const long long total = (long long)d.G * MTc * NCHo * d.KK * 64 * 4;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int quad = (int)(r % d.KK);
    r /= d.KK;
    const int ch = (int)(r % NCHo);
    r /= NCHo;
    const int mt = (int)(r % MTc);
    const int g = (int)(r / MTc);
    const int kl = 2 * (quad * 4 + j) + (lane >> 5);
    const int o = ch * CCH + kl / d.KK;
    const int tap = kl % d.KK;
    const int c = mt * 32 + (lane & 31);
    float v = 0.f;
    if (o < d.Cog && c < d.Cig) v = w[((long long)(g * d.Cog + o) * d.Cig + c) * d.KK + tap];
    wp[idx] = v;
  }
}

__global__ __launch_bounds__(NTHREADS) void sphere_bwd_data_adj_kernel(const float* __restrict__ gy, const int* __restrict__ rowptr,
                                                                        const int2* __restrict__ entries,
                                                                        const float4* __restrict__ wp, float* __restrict__ gx, Dims d,
                                                                        int MTc, int NCHo, int qtiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int2* span = reinterpret_cast<int2*>(smem);  // [KK][P]: (first entry, count) of L(k, q)
  float* colbuf = reinterpret_cast<float*>(smem + (size_t)d.KK * P * 8);
  const int rows = CCH * d.KK;
  const int HWin = d.H * d.W;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring pixel tiles (overlapping gather footprints) on one XCD
  const int b = tile / qtiles;
  const int q0 = (tile - b * qtiles) * P;
  const int g = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int mt = blockIdx.y * 4 + wave;
  const bool active = mt < MTc;

  for (int item = tid; item < d.KK * P; item += NTHREADS) {
    const int k = item / P, pp = item % P;
    const int qq = q0 + pp;
    int2 s = make_int2(0, 0);
    if (qq < HWin) {
      const int beg = rowptr[(long long)k * HWin + qq];
      s = make_int2(beg, rowptr[(long long)k * HWin + qq + 1] - beg);
    }
    span[item] = s;
  }
  __syncthreads();

  const int p = tid & (P - 1);
  const int qd = tid / P;  // 0..3 -> output channels 2qd, 2qd+1 of the chunk
  const float* gyg = gy + ((long long)b * d.Co + (long long)g * d.Cog) * d.npix;

  auto produce = [&](int ch, float* buf) {
    const int o0 = ch * CCH + qd * 2;
    const bool ok0 = o0 < d.Cog, ok1 = o0 + 1 < d.Cog;
    const float* g0 = gyg + (long long)(ok0 ? o0 : 0) * d.npix;
    const float* g1 = gyg + (long long)(ok1 ? o0 + 1 : 0) * d.npix;
    for (int k = 0; k < d.KK; ++k) {
      const int2 s = span[k * P + p];
      float v0 = 0.f, v1 = 0.f;
      for (int e = 0; e < s.y; ++e) {
        const int2 ent = entries[s.x + e];
        const float wt = __int_as_float(ent.y);
        v0 += wt * g0[ent.x];
        v1 += wt * g1[ent.x];
      }
      buf[((qd * 2) * d.KK + k) * P + p] = ok0 ? v0 : 0.f;
      buf[((qd * 2 + 1) * d.KK + k) * P + p] = ok1 ? v1 : 0.f;
    }
  };

  f32x16 acc0 = {0}, acc1 = {0};
  const float4* wpa = wp + ((long long)(g * MTc + (active ? mt : 0)) * NCHo) * d.KK * 64 + lane;

  produce(0, colbuf);
  __syncthreads();
  for (int ch = 0; ch < NCHo; ++ch) {
    float* cur = colbuf + (ch & 1) * rows * P;
    if (ch + 1 < NCHo) produce(ch + 1, colbuf + ((ch + 1) & 1) * rows * P);
    if (active) {
      const float4* wq = wpa + (long long)ch * d.KK * 64;
      const float* bp = cur + (lane >> 5) * P + (lane & 31);
      for (int quad = 0; quad < d.KK; ++quad) {
        const float4 a4 = wq[quad * 64];
        const float* bq = bp + quad * 8 * P;
        acc0 = mfma32(a4.x, bq[0], acc0);
        acc1 = mfma32(a4.x, bq[32], acc1);
        acc0 = mfma32(a4.y, bq[2 * P], acc0);
        acc1 = mfma32(a4.y, bq[2 * P + 32], acc1);
        acc0 = mfma32(a4.z, bq[4 * P], acc0);
        acc1 = mfma32(a4.z, bq[4 * P + 32], acc1);
        acc0 = mfma32(a4.w, bq[6 * P], acc0);
        acc1 = mfma32(a4.w, bq[6 * P + 32], acc1);
      }
    }
    __syncthreads();
  }

  if (active) {
    float* gxb = gx + ((long long)b * d.Ci + (long long)g * d.Cig) * HWin;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (c < d.Cig) {
        const int qq = q0 + (lane & 31);
        float* o0 = gxb + (long long)c * HWin + qq;
        if (qq < HWin) o0[0] = d.accumulate ? o0[0] + acc0[r] : acc0[r];
        if (qq + 32 < HWin) o0[32] = d.accumulate ? o0[32] + acc1[r] : acc1[r];
      }
    }
  }
}

// Software-pipelined variant of the kernel above for 3x3 kernels (KK = 9).  The first EC = 4 adjoint entries of every
// (tap, input pixel) of the tile -- all of them for the tables of the network away from the poles -- are cached in LDS once
// per tile; the gy loads of the next chunk (2 channels x 9 taps x 4 entries per thread) are issued before the MFMA phase of
// the current chunk and combined after it.  Entries beyond the fourth are added by a (rare) synchronous tail loop.
__global__ __launch_bounds__(NTHREADS) void sphere_bwd_data_adj9_kernel(const float* __restrict__ gy, const int* __restrict__ rowptr,
                                                                         const int2* __restrict__ entries,
                                                                         const float4* __restrict__ wp, float* __restrict__ gx, Dims d,
                                                                         int MTc, int NCHo, int qtiles, const int* __restrict__ tile_list) {
  constexpr int KT = 9, EC = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int4* ent_p = reinterpret_cast<int4*>(smem);                          // [KT][P] output pixels of the first 4 entries
  float4* ent_w = reinterpret_cast<float4*>(smem + KT * P * 16);        // [KT][P] their weights (0 beyond the list)
  int2* span = reinterpret_cast<int2*>(smem + KT * P * 32);             // [KT][P] (first entry, count)
  float* colbuf = reinterpret_cast<float*>(smem + KT * P * 40);         // [2][CCH*KT][P]
  constexpr int rows = CCH * KT;
  const int HWin = d.H * d.W;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring pixel tiles (overlapping gather footprints) on one XCD
  const int b = tile / qtiles;
  // tile_list: only the listed 64-pixel tiles (qtiles of them per sample) -- the rest of the image belongs to the windowed kernel
  const int q0 = (tile_list ? tile_list[tile - b * qtiles] : tile - b * qtiles) * P;
  const int g = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int mt = blockIdx.y * 4 + wave;
  const bool active = mt < MTc;

  for (int item = tid; item < KT * P; item += NTHREADS) {
    const int k = item / P, pp = item % P;
    const int qq = q0 + pp;
    int beg = 0, cnt = 0;
    if (qq < HWin) {
      beg = rowptr[(long long)k * HWin + qq];
      cnt = rowptr[(long long)k * HWin + qq + 1] - beg;
    }
    int pe[EC];
    float we[EC];
#pragma unroll
    for (int e = 0; e < EC; ++e) {
      // always a valid element of this row's list (or element 0): the load stays unconditional, the selects come after
      const int2 ent = entries[beg + min(e, max(cnt - 1, 0))];
      pe[e] = e < cnt ? ent.x * 4 : 0;  // byte offset within a channel plane (32-bit: the gathers use a scalar base)
      we[e] = e < cnt ? __int_as_float(ent.y) : 0.f;
    }
    ent_p[item] = make_int4(pe[0], pe[1], pe[2], pe[3]);
    ent_w[item] = make_float4(we[0], we[1], we[2], we[3]);
    span[item] = make_int2(beg, cnt);
  }
  __syncthreads();

  const int p = tid & (P - 1);
  static_assert(P == 64, "one wave per output-channel pair of the chunk");
  const int qd = __builtin_amdgcn_readfirstlane(tid / P);  // output channels 2qd, 2qd+1 of the chunk (wave-uniform -> scalar bases)
  const float* gyg = gy + ((long long)b * d.Co + (long long)g * d.Cog) * d.npix;
  float lv[KT * 2 * EC];
  bool has_tail = false;
#pragma unroll
  for (int k = 0; k < KT; ++k) has_tail |= span[k * P + p].y > EC;

  auto issue = [&](int ch) {
    const int o0 = ch * CCH + qd * 2;
    const float* g0 = gyg + (long long)(o0 < d.Cog ? o0 : 0) * d.npix;
    const float* g1 = gyg + (long long)(o0 + 1 < d.Cog ? o0 + 1 : 0) * d.npix;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int4 pe = ent_p[k * P + p];
      const char* c0 = reinterpret_cast<const char*>(g0);
      const char* c1 = reinterpret_cast<const char*>(g1);
#define MODE_LD(base, off) (*reinterpret_cast<const float*>((base) + (unsigned)(off)))
      lv[k * 8 + 0] = MODE_LD(c0, pe.x); lv[k * 8 + 1] = MODE_LD(c0, pe.y); lv[k * 8 + 2] = MODE_LD(c0, pe.z); lv[k * 8 + 3] = MODE_LD(c0, pe.w);
      lv[k * 8 + 4] = MODE_LD(c1, pe.x); lv[k * 8 + 5] = MODE_LD(c1, pe.y); lv[k * 8 + 6] = MODE_LD(c1, pe.z); lv[k * 8 + 7] = MODE_LD(c1, pe.w);
#undef MODE_LD
    }
  };
  auto finish = [&](int ch, float* buf) {
    const int o0 = ch * CCH + qd * 2;
    const bool ok0 = o0 < d.Cog, ok1 = o0 + 1 < d.Cog;
    const float* g0 = gyg + (long long)(ok0 ? o0 : 0) * d.npix;
    const float* g1 = gyg + (long long)(ok1 ? o0 + 1 : 0) * d.npix;
    float v0[KT], v1[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const float4 we = ent_w[k * P + p];
      // Two SCALAR FMA chains, kept apart by opaque asm.  Written as plain expressions the two chains are SLP-packed into
      // v_pk_mul_f32 / v_pk_fma_f32 on register pairs whose halves are the results of two different gathers, and that form was not
      // repeatable: with a second process on the GPU (loads slow enough for the wave to sit in its s_waitcnt) one (channel, tap) row of
      // the column tile came out wrong in lanes 48..63 -- about one call in 300; in-kernel checks showed the gathered registers equal
      // to memory afterwards and the combined value not (round 4; tools/determinism_hunt.py: 22-46 differing steps in 232
      // with the packed form, 0 in 232 with this one).
      float a = we.x * lv[k * 8 + 0];
      float b = we.x * lv[k * 8 + 4];
      asm volatile("" : "+v"(a), "+v"(b));
      a = fmaf(we.y, lv[k * 8 + 1], a);
      b = fmaf(we.y, lv[k * 8 + 5], b);
      asm volatile("" : "+v"(a), "+v"(b));
      a = fmaf(we.z, lv[k * 8 + 2], a);
      b = fmaf(we.z, lv[k * 8 + 6], b);
      asm volatile("" : "+v"(a), "+v"(b));
      a = fmaf(we.w, lv[k * 8 + 3], a);
      b = fmaf(we.w, lv[k * 8 + 7], b);
      asm volatile("" : "+v"(a), "+v"(b));
      v0[k] = a;
      v1[k] = b;
    }
    if (has_tail) {  // this pixel has a list longer than EC entries for some tap (near the poles): one branch per chunk
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int2 s = span[k * P + p];
        for (int e = EC; e < s.y; ++e) {
          const int2 ent = entries[s.x + e];
          const float wt = __int_as_float(ent.y);
          float a = g0[ent.x], b = g1[ent.x];
          asm volatile("" : "+v"(a), "+v"(b));  // (no v_pk_fma_f32 on the pair of freshly loaded values: see above)
          v0[k] = fmaf(wt, a, v0[k]);
          asm volatile("" : "+v"(v0[k]));
          v1[k] = fmaf(wt, b, v1[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      buf[((qd * 2) * KT + k) * P + p] = ok0 ? v0[k] : 0.f;
      buf[((qd * 2 + 1) * KT + k) * P + p] = ok1 ? v1[k] : 0.f;
    }
  };

  f32x16 acc0 = {0}, acc1 = {0};
  const float4* wpa = wp + ((long long)(g * MTc + (active ? mt : 0)) * NCHo) * KT * 64 + lane;

  // Weight fragments of chunk ch+1 (9 x float4) are requested BEFORE the gathers of chunk ch+1, so that by the time the MFMAs
  // of chunk ch+1 run they are long complete and nothing in the MFMA phase waits on the (in-order) vector-memory counter;
  // the 8 LDS operands of quad q+1 are read before the 16 MFMAs of quad q are issued.  (The straightforward loop waited for an
  // LDS round trip every two MFMAs and for an L1/L2 round trip -- and with it for all 72 gathers in flight -- every quad.)
  float4 wcur[KT], wnxt[KT];
#pragma unroll
  for (int quad = 0; quad < KT; ++quad) wcur[quad] = wpa[quad * 64];
  issue(0);
  finish(0, colbuf);
  __syncthreads();
  for (int ch = 0; ch < NCHo; ++ch) {
    float* cur = colbuf + (ch & 1) * rows * P;
    float* nxt = colbuf + ((ch + 1) & 1) * rows * P;
    if (ch + 1 < NCHo) {
      const float4* wq = wpa + (long long)(ch + 1) * KT * 64;
#pragma unroll
      for (int quad = 0; quad < KT; ++quad) wnxt[quad] = wq[quad * 64];
      issue(ch + 1);
    }
    if (active) {
      const float* bp = cur + (lane >> 5) * P + (lane & 31);
      float bn[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) bn[i] = bp[(i >> 1) * 2 * P + (i & 1) * 32];
#pragma unroll
      for (int quad = 0; quad < KT; ++quad) {
        const float4 a4 = wcur[quad];
        float bc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bc[i] = bn[i];
        if (quad + 1 < KT) {
          const float* bq = bp + (quad + 1) * 8 * P;
#pragma unroll
          for (int i = 0; i < 8; ++i) bn[i] = bq[(i >> 1) * 2 * P + (i & 1) * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = mfma32(a4.x, bc[0], acc0);
        acc1 = mfma32(a4.x, bc[1], acc1);
        acc0 = mfma32(a4.y, bc[2], acc0);
        acc1 = mfma32(a4.y, bc[3], acc1);
        acc0 = mfma32(a4.z, bc[4], acc0);
        acc1 = mfma32(a4.z, bc[5], acc1);
        acc0 = mfma32(a4.w, bc[6], acc0);
        acc1 = mfma32(a4.w, bc[7], acc1);
      }
    }
    if (ch + 1 < NCHo) {
      finish(ch + 1, nxt);
#pragma unroll
      for (int quad = 0; quad < KT; ++quad) wcur[quad] = wnxt[quad];
    }
    __syncthreads();
  }

  if (active) {
    float* gxb = gx + ((long long)b * d.Ci + (long long)g * d.Cig) * HWin;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (c < d.Cig) {
        const int qq = q0 + (lane & 31);
        float* o0 = gxb + (long long)c * HWin + qq;
        if (qq < HWin) o0[0] = d.accumulate ? o0[0] + acc0[r] : acc0[r];
        if (qq + 32 < HWin) o0[32] = d.accumulate ? o0[32] + acc1[r] : acc1[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Backward w.r.t. the input, scatter form (needs no adjoint table).  grid = (B*tps, 1, G); LDS = KK*P*20 + KSQ*8*P*4 bytes.
// gcol rows come out of the MFMA in 128-row blocks (CB channels x KK taps) and are scattered with the same
// bilinear weights as the forward gather (the transpose of cu:83-113, i.e. what cu:293-356 computes).
__global__ __launch_bounds__(NTHREADS) void sphere_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ pos,
                                                                    const float4* __restrict__ wp, float* __restrict__ gx,
                                                                    Dims d) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* tap_w = reinterpret_cast<float4*>(smem);
  unsigned* tap_off = reinterpret_cast<unsigned*>(smem + (size_t)d.KK * P * 16);
  float* gyl = reinterpret_cast<float*>(smem + (size_t)d.KK * P * 20);  // [KSQ*8][P]

  const int tile = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring pixel tiles (overlapping gather footprints) on one XCD
  const int b = tile / d.tps;
  const int pix0 = (tile - b * d.tps) * P;
  const int g = blockIdx.z;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;

  compute_tapinfo(pos, d, pix0, tap_off, tap_w);
  const float* gyb = gy + ((long long)b * d.Co + (long long)g * d.Cog) * d.npix;
  for (int idx = tid; idx < d.KSQ * 8 * P; idx += NTHREADS) {
    const int co = idx / P, p = idx % P;
    float v = 0.f;
    if (co < d.Cog && pix0 + p < d.npix) v = gyb[(long long)co * d.npix + pix0 + p];
    gyl[idx] = v;
  }
  __syncthreads();

  const long long HW = (long long)d.H * d.W;
  float* gxg = gx + ((long long)b * d.Ci + (long long)g * d.Cig) * HW;
  const float* bp = gyl + (lane >> 5) * P + (lane & 31);
  const int nrows = d.CB * d.KK;

  for (int nb = 0; nb < d.NB; ++nb) {
    f32x16 acc0 = {0}, acc1 = {0};
    const float4* wq = wp + ((long long)((g * d.NB + nb) * 4 + wave) * d.KSQ) * 64 + lane;
    for (int ks4 = 0; ks4 < d.KSQ; ++ks4) {
      const float4 a4 = wq[ks4 * 64];
      const float* bq = bp + ks4 * 8 * P;
      acc0 = mfma32(a4.x, bq[0], acc0);
      acc1 = mfma32(a4.x, bq[32], acc1);
      acc0 = mfma32(a4.y, bq[2 * P], acc0);
      acc1 = mfma32(a4.y, bq[2 * P + 32], acc1);
      acc0 = mfma32(a4.z, bq[4 * P], acc0);
      acc1 = mfma32(a4.z, bq[4 * P + 32], acc1);
      acc0 = mfma32(a4.w, bq[6 * P], acc0);
      acc1 = mfma32(a4.w, bq[6 * P + 32], acc1);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int cl = rr / d.KK;
      const int k = rr - cl * d.KK;
      const int c = nb * d.CB + cl;
      if (rr < nrows && c < d.Cig) {
        float* gxc = gxg + c * HW;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const int pp = nt * 32 + (lane & 31);
          const float v = nt ? acc1[r] : acc0[r];
          const unsigned po = tap_off[k * P + pp];
          const float4 tw = tap_w[k * P + pp];
          const int off = (int)(po & 0x3fffffffu);
          const int dw = (int)((po >> 30) & 1u);
          const int dh = (po >> 31) ? d.W : 0;
          if (tw.x != 0.f) atomicAdd(gxc + off, tw.x * v);
          if (tw.y != 0.f) atomicAdd(gxc + off + dw, tw.y * v);
          if (tw.z != 0.f) atomicAdd(gxc + off + dh, tw.z * v);
          if (tw.w != 0.f) atomicAdd(gxc + off + dh + dw, tw.w * v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Backward w.r.t. the weight.  grid = (S, NB, G*MG); LDS = KK*P*20 + 2 * 128*(P+1)*4 bytes.
// Split-K over pixel tiles: slice s owns tiles s, s+S, ...; partial[s][g][mg][nb][128][128] is summed by
// reduce_gw in a fixed order (deterministic, unlike the reference's cuBLAS/atomic path).
constexpr int PS = P + 1;  // padded row stride: bank = (row*65 + k) % 32 -> conflict-free fragment reads

__global__ __launch_bounds__(NTHREADS) void sphere_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ pos,
                                                                      const float* __restrict__ x, float* __restrict__ part,
                                                                      Dims d, int S, int MG, const int* __restrict__ pixmap,
                                                                      int nsel) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* tap_w = reinterpret_cast<float4*>(smem);
  unsigned* tap_off = reinterpret_cast<unsigned*>(smem + (size_t)d.KK * P * 16);
  float* gyl = reinterpret_cast<float*>(smem + (size_t)d.KK * P * 20);  // [128][PS]
  float* col = gyl + 128 * PS;                                           // [128][PS]

  const int s = blockIdx.x, nb = blockIdx.y;
  const int g = blockIdx.z / MG, mg = blockIdx.z % MG;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int p = tid & (P - 1), q = tid / P;
  const long long HW = (long long)d.H * d.W;
  const int nrows = d.CB * d.KK;
  const int T = d.B * d.tps;

  f32x16 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x16){0};

  for (int t = s; t < T; t += S) {
    const int b = t / d.tps;
    const int pix0 = (t - b * d.tps) * P;
    compute_tapinfo(pos, d, pix0, tap_off, tap_w, pixmap, nsel);
    const float* gyb = gy + ((long long)b * d.Co + (long long)g * d.Cog) * d.npix;
    int mypix = pix0 + p;  // this thread's output pixel of the tile (d.npix = none)
    if (pixmap) mypix = mypix < nsel ? pixmap[mypix] : d.npix;
    // gy tile: a wave loads 64 consecutive pixels of one output-channel row per instruction; 8 loads in flight per thread
#pragma unroll 1
    for (int r0 = 0; r0 < 128; r0 += 32) {
      float t8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = r0 + j * 4 + q;
        const bool ok = mg * 128 + row < d.Cog && mypix < d.npix;
        const float v = gyb[ok ? (long long)(mg * 128 + row) * d.npix + mypix : 0];
        t8[j] = ok ? v : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) gyl[(r0 + j * 4 + q) * PS + p] = t8[j];
    }
    __syncthreads();
    // column tile: 32 samples per thread, issued 8 at a time (32 corner loads in flight), branch-free
    const float* xg = x + ((long long)b * d.Ci + (long long)g * d.Cig) * HW;
#pragma unroll 1
    for (int i0 = 0; i0 < 32; i0 += 8) {
      float l[8][4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rr = q * 32 + i0 + j;
        const int cl = rr / d.KK;
        const int k = rr - cl * d.KK;
        const int c = nb * d.CB + cl;
        const bool ok = rr < nrows && c < d.Cig;
        const unsigned po = tap_off[(ok ? k : 0) * P + p];
        const int off = (int)(po & 0x3fffffffu);
        const int dw = (int)((po >> 30) & 1u);
        const int dh = (po >> 31) ? d.W : 0;
        const float* xc = xg + (ok ? c : 0) * HW;
        l[j][0] = xc[off];
        l[j][1] = xc[off + dw];
        l[j][2] = xc[off + dh];
        l[j][3] = xc[off + dh + dw];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rr = q * 32 + i0 + j;
        const int cl = rr / d.KK;
        const int k = rr - cl * d.KK;
        const bool ok = rr < nrows && nb * d.CB + cl < d.Cig;
        const float4 tw = tap_w[(ok ? k : 0) * P + p];
        const float v = tw.x * l[j][0] + tw.y * l[j][1] + tw.z * l[j][2] + tw.w * l[j][3];
        col[rr * PS + p] = ok ? v : 0.f;
      }
    }
    __syncthreads();
    const float* ap = gyl + (wave * 32 + (lane & 31)) * PS + (lane >> 5);
    const float* bp = col + (lane & 31) * PS + (lane >> 5);
#pragma unroll 4
    for (int ks = 0; ks < P / 2; ++ks) {
      const float a = ap[2 * ks];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = mfma32(a, bp[nt * 32 * PS + 2 * ks], acc[nt]);
    }
    __syncthreads();
  }

  float* pb = part + ((((long long)s * d.G + g) * MG + mg) * d.NB + nb) * (128 * 128);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      pb[i * 128 + nt * 32 + (lane & 31)] = acc[nt][r];
    }
}

__global__ void reduce_gw(const float* __restrict__ part, float* __restrict__ gw, Dims d, int S, int MG) {
  const long long total = (long long)d.Co * d.Cig * d.KK;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % d.KK);
    long long r = idx / d.KK;
    const int c = (int)(r % d.Cig);
    const int o = (int)(r / d.Cig);
    const int g = o / d.Cog, ol = o % d.Cog;
    const int mg = ol / 128, row = ol % 128;
    const int nb = c / d.CB, colr = (c % d.CB) * d.KK + k;
    const long long stride = (long long)d.G * MG * d.NB * (128 * 128);
    const float* pp = part + (((long long)g * MG + mg) * d.NB + nb) * (128 * 128) + row * 128 + colr;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // fixed association, 4 independent loads in flight
    int s = 0;
    for (; s + 3 < S; s += 4) {
      a0 += pp[s * stride];
      a1 += pp[(s + 1) * stride];
      a2 += pp[(s + 2) * stride];
      a3 += pp[(s + 3) * stride];
    }
    for (; s < S; ++s) a0 += pp[s * stride];
    gw[idx] += (a0 + a1) + (a2 + a3);
  }
}

// ---------------------------------------------------------------------------------------------------------
int make_dims(Dims& d, int B, int Ci, int H, int W, int Co, int Kh, int Kw, int sH, int sW, int Ho, int Wo, int groups,
              const char* who) {
  MODE_REQUIRE(B >= 0 && Ci > 0 && H > 0 && W > 0 && Co > 0 && Kh > 0 && Kw > 0 && sH > 0 && sW > 0 && Ho > 0 && Wo > 0 &&
                   groups > 0,
               MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(Ci % groups == 0 && Co % groups == 0, MODE_ERR_BAD_ARG, "%s: channels (%d,%d) not divisible by groups %d", who,
               Ci, Co, groups);
  // the table is indexed at (h_out*sH, w_out*sW) and must stay inside (H, W)  (sphere_conv_cuda.cpp:107-109)
  MODE_REQUIRE((long long)(Ho - 1) * sH < H && (long long)(Wo - 1) * sW < W, MODE_ERR_BAD_ARG,
               "%s: output %dx%d with stride %dx%d reads the position table outside %dx%d", who, Ho, Wo, sH, sW, H, W);
  MODE_REQUIRE(Kh * Kw <= 32, MODE_ERR_UNSUPPORTED, "%s: kernel %dx%d has more than 32 taps", who, Kh, Kw);
  MODE_REQUIRE((long long)H * W < (1ll << 30), MODE_ERR_UNSUPPORTED, "%s: image larger than 2^30 pixels", who);
  d.B = B; d.Ci = Ci; d.H = H; d.W = W; d.Co = Co; d.KK = Kh * Kw; d.sH = sH; d.sW = sW; d.Ho = Ho; d.Wo = Wo; d.G = groups;
  d.Cig = Ci / groups; d.Cog = Co / groups;
  d.npix = Ho * Wo;
  d.tps = mode::cdiv(d.npix, P);
  d.MT = mode::cdiv(d.Cog, 32);
  d.NCHUNK = mode::cdiv(d.Cig, CCH);
  d.CB = 128 / d.KK;
  d.NB = mode::cdiv(d.Cig, d.CB);
  d.KSQ = mode::cdiv(d.Cog, 8);
  return MODE_OK;
}

size_t wpack_floats(const Dims& d) {
  const size_t f = (size_t)d.G * d.MT * d.NCHUNK * d.KK * 256;
  const size_t bw = (size_t)d.G * d.NB * 4 * d.KSQ * 256;
  const size_t adj = (size_t)d.G * mode::cdiv(d.Cig, 32) * mode::cdiv(d.Cog, CCH) * d.KK * 256;
  return std::max(f, std::max(bw, adj)) + (size_t)d.Co;  // + the folded BatchNorm shifts of the *_bn entry point
}

int bww_splits(const Dims& d, int MG) {
  const int T = d.B * d.tps;
  int S = mode::cdiv(768, d.NB * d.G * MG);
  if (S > T) S = T;
  if (S < 1) S = 1;
  return S;
}

}  // namespace

extern "C" size_t mode_sphere_conv_wpack_bytes(int Ci, int Co, int Kh, int Kw, int groups) {
  Dims d;
  if (make_dims(d, 1, Ci, 1 << 14, 1 << 14, Co, Kh, Kw, 1, 1, 1, 1, groups, "mode_sphere_conv_wpack_bytes") != MODE_OK) return 0;
  return wpack_floats(d) * sizeof(float);
}

namespace {
template <int KT, bool EPI>
int launch_fwd(const float* x, const float* pos, const float* wpack, float* y, const Dims& d, dim3 grid, size_t lds, hipStream_t st,
               const Epi& epi) {
  int rc = mode::allow_lds(sphere_fwd_kernel<KT, EPI>, lds, "mode_sphere_conv_fwd");
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL((sphere_fwd_kernel<KT, EPI>), grid, dim3(NTHREADS), lds, st, x, pos, reinterpret_cast<const float4*>(wpack), y, d, epi);
  return mode::check_launch("mode_sphere_conv_fwd");
}
}  // namespace

static int sphere_conv_fwd_impl(const float* x, const float* pos, const float* w, float* y, float* wpack, int B, int Ci, int H, int W,
                                int Co, int Kh, int Kw, int sH, int sW, int Ho, int Wo, int groups, mode_stream_t stream,
                                const mode_bn_epilogue* bn) {
  MODE_REQUIRE(x && pos && w && y && wpack, MODE_ERR_BAD_ARG, "mode_sphere_conv_fwd: null pointer");
  Dims d;
  int rc = make_dims(d, B, Ci, H, W, Co, Kh, Kw, sH, sW, Ho, Wo, groups, "mode_sphere_conv_fwd");
  if (rc != MODE_OK) return rc;
  if (B == 0) return MODE_OK;
  hipStream_t st = mode::as_stream(stream);
  const long long npack = (long long)d.G * d.MT * d.NCHUNK * d.KK * 256;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w_fwd, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, d, bn ? 1 : 0, bn ? *bn : mode_bn_epilogue());
  const Epi epi = make_epi(bn, wpack + npack);
  const size_t lds = (size_t)d.KK * P * 20 + 2 * (size_t)CCH * d.KK * P * 4;
  const dim3 grid(B * d.tps, mode::cdiv(d.MT, 4), d.G);
  if (d.KK == 9)  // the 3x3 kernels of the network: software-pipelined gather
    return bn ? launch_fwd<9, true>(x, pos, wpack, y, d, grid, lds, st, epi) : launch_fwd<9, false>(x, pos, wpack, y, d, grid, lds, st, epi);
  return bn ? launch_fwd<0, true>(x, pos, wpack, y, d, grid, lds, st, epi) : launch_fwd<0, false>(x, pos, wpack, y, d, grid, lds, st, epi);
}

extern "C" int mode_sphere_conv_fwd(const float* x, const float* pos, const float* w, float* y, float* wpack, int B, int Ci,
                                    int H, int W, int Co, int Kh, int Kw, int sH, int sW, int Ho, int Wo, int groups,
                                    mode_stream_t stream) {
  return sphere_conv_fwd_impl(x, pos, w, y, wpack, B, Ci, H, W, Co, Kh, Kw, sH, sW, Ho, Wo, groups, stream, nullptr);
}

extern "C" int mode_sphere_conv_fwd_bn(const float* x, const float* pos, const float* w, const mode_bn_epilogue* bn, float* y,
                                       float* wpack, int B, int Ci, int H, int W, int Co, int Kh, int Kw, int sH, int sW, int Ho,
                                       int Wo, int groups, mode_stream_t stream) {
  int rc = mode::check_bn(bn, "mode_sphere_conv_fwd_bn");
  if (rc != MODE_OK) return rc;
  return sphere_conv_fwd_impl(x, pos, w, y, wpack, B, Ci, H, W, Co, Kh, Kw, sH, sW, Ho, Wo, groups, stream, bn);
}

extern "C" int mode_sphere_conv_bwd_data(const float* gy, const float* pos, const float* w, float* gx, float* wpack, int B,
                                         int Ci, int H, int W, int Co, int Kh, int Kw, int sH, int sW, int Ho, int Wo,
                                         int groups, mode_stream_t stream) {
  MODE_REQUIRE(gy && pos && w && gx && wpack, MODE_ERR_BAD_ARG, "mode_sphere_conv_bwd_data: null pointer");
  Dims d;
  int rc = make_dims(d, B, Ci, H, W, Co, Kh, Kw, sH, sW, Ho, Wo, groups, "mode_sphere_conv_bwd_data");
  if (rc != MODE_OK) return rc;
  if (B == 0) return MODE_OK;
  hipStream_t st = mode::as_stream(stream);
  const long long npack = (long long)d.G * d.NB * 4 * d.KSQ * 256;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w_bwd, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, d);
  const size_t lds = (size_t)d.KK * P * 20 + (size_t)d.KSQ * 8 * P * 4;
  rc = mode::allow_lds(sphere_bwd_data_kernel, lds, "mode_sphere_conv_bwd_data (Co/groups too large)");
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(sphere_bwd_data_kernel, dim3(B * d.tps, 1, d.G), dim3(NTHREADS), lds, st, gy, pos,
                     reinterpret_cast<const float4*>(wpack), gx, d);
  return mode::check_launch("mode_sphere_conv_bwd_data");
}

extern "C" size_t mode_sphere_conv_bwd_weight_workspace_bytes(int B, int Ci, int Co, int Kh, int Kw, int Ho, int Wo,
                                                              int groups) {
  Dims d;
  if (make_dims(d, B, Ci, 1 << 14, 1 << 14, Co, Kh, Kw, 1, 1, Ho, Wo, groups, "mode_sphere_conv_bwd_weight_workspace_bytes") !=
      MODE_OK)
    return 0;
  const int MG = mode::cdiv(d.Cog, 128);
  return (size_t)bww_splits(d, MG) * d.G * MG * d.NB * 128 * 128 * sizeof(float);
}

namespace mode {
// General (gather) weight-gradient kernels over all output pixels (pixmap == nullptr) or over the listed ones only; ADDS the
// result to gw.  Used by mode_sphere_conv_bwd_weight and, for the tiles the windowed kernel leaves out, by
// mode_sphere_conv_bwd_weight_win (sphere_conv_win.hip).
size_t sphere_bwd_weight_general_workspace(int B, int Ci, int Co, int Kh, int Kw, int Ho, int Wo, int groups, int nsel) {
  Dims d;
  if (make_dims(d, B, Ci, 1 << 14, 1 << 14, Co, Kh, Kw, 1, 1, Ho, Wo, groups, "mode_sphere_conv_bwd_weight_workspace_bytes") != MODE_OK)
    return 0;
  if (nsel > 0) d.tps = mode::cdiv(nsel, P);
  const int MG = mode::cdiv(d.Cog, 128);
  return (size_t)bww_splits(d, MG) * d.G * MG * d.NB * 128 * 128 * sizeof(float);
}

int sphere_bwd_weight_general(const float* gy, const float* pos, const float* x, float* gw, float* workspace, int B, int Ci, int H,
                              int W, int Co, int Kh, int Kw, int sH, int sW, int Ho, int Wo, int groups, const int* pixmap, int nsel,
                              hipStream_t st, const char* who) {
  Dims d;
  int rc = make_dims(d, B, Ci, H, W, Co, Kh, Kw, sH, sW, Ho, Wo, groups, who);
  if (rc != MODE_OK) return rc;
  if (B == 0 || (pixmap && nsel == 0)) return MODE_OK;
  if (pixmap) d.tps = mode::cdiv(nsel, P);
  const int MG = mode::cdiv(d.Cog, 128);
  const int S = bww_splits(d, MG);
  const size_t lds = (size_t)d.KK * P * 20 + 2 * (size_t)128 * PS * 4;
  rc = mode::allow_lds(sphere_bwd_weight_kernel, lds, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(sphere_bwd_weight_kernel, dim3(S, d.NB, d.G * MG), dim3(NTHREADS), lds, st, gy, pos, x, workspace, d, S, MG,
                     pixmap, nsel);
  rc = mode::check_launch(who);
  if (rc != MODE_OK) return rc;
  const long long n = (long long)d.Co * d.Cig * d.KK;
  hipLaunchKernelGGL(reduce_gw, dim3(mode::cdiv(n, 256)), dim3(256), 0, st, workspace, gw, d, S, MG);
  return mode::check_launch(who);
}
}  // namespace mode

extern "C" int mode_sphere_conv_bwd_weight(const float* gy, const float* pos, const float* x, float* gw, float* workspace,
                                           int B, int Ci, int H, int W, int Co, int Kh, int Kw, int sH, int sW, int Ho, int Wo,
                                           int groups, mode_stream_t stream) {
  MODE_REQUIRE(gy && pos && x && gw, MODE_ERR_BAD_ARG, "mode_sphere_conv_bwd_weight: null pointer");
  MODE_REQUIRE(workspace, MODE_ERR_WORKSPACE, "mode_sphere_conv_bwd_weight: workspace required");
  return mode::sphere_bwd_weight_general(gy, pos, x, gw, workspace, B, Ci, H, W, Co, Kh, Kw, sH, sW, Ho, Wo, groups, nullptr, 0,
                                         mode::as_stream(stream), "mode_sphere_conv_bwd_weight");
}

// ---------------------------------------------------------------------------------------------------------
// Adjoint (transposed) sampling table, built on the HOST once per position table (the table is a constant of the
// module: sphere_conv.py:150, 156-157).  For tap k and input pixel q it lists every (output pixel p, weight) with
// S_k[p, q] != 0, using exactly the arithmetic of compute_tapinfo above.  CSR layout:
//   rowptr[k*H*W + q] .. rowptr[k*H*W + q + 1]  index into entries[] = (p, float bits of the weight) pairs.
extern "C" size_t mode_sphere_adjoint_max_entries(int Kh, int Kw, int Ho, int Wo) {
  if (Kh <= 0 || Kw <= 0 || Ho <= 0 || Wo <= 0) return 0;
  return (size_t)Kh * Kw * Ho * Wo * 4;
}

extern "C" int mode_sphere_adjoint_build(const float* pos_host, int H, int W, int Kh, int Kw, int sH, int sW, int Ho, int Wo,
                                         int32_t* rowptr_host, int32_t* entries_host, int64_t* n_entries) {
  MODE_REQUIRE(pos_host && rowptr_host && entries_host && n_entries, MODE_ERR_BAD_ARG, "mode_sphere_adjoint_build: null pointer");
  MODE_REQUIRE(H > 0 && W > 0 && Kh > 0 && Kw > 0 && sH > 0 && sW > 0 && Ho > 0 && Wo > 0, MODE_ERR_BAD_ARG,
               "mode_sphere_adjoint_build: non-positive size");
  MODE_REQUIRE((long long)(Ho - 1) * sH < H && (long long)(Wo - 1) * sW < W, MODE_ERR_BAD_ARG,
               "mode_sphere_adjoint_build: output %dx%d with stride %dx%d reads the position table outside %dx%d", Ho, Wo, sH, sW, H,
               W);
  const int KK = Kh * Kw;
  const long long HW = (long long)H * W;
  MODE_REQUIRE((long long)KK * HW < (1ll << 31) && (long long)KK * Ho * Wo * 4 < (1ll << 31), MODE_ERR_UNSUPPORTED,
               "mode_sphere_adjoint_build: table too large for 32-bit indices");
  const long long nrows = (long long)KK * HW;
  // corner records of one (k, p): up to 4 (q, weight)
  auto corners = [&](int k, int p, int qs[4], float ws[4]) -> int {
    const int ho = p / Wo, wo = p - ho * Wo;
    const long long idx = (long long)(ho * sH) * W + wo * sW;
    const float h = pos_host[(long long)(2 * k) * HW + idx];
    const float w = pos_host[(long long)(2 * k + 1) * HW + idx];
    if (!(h > -1.f && w > -1.f && h < (float)H && w < (float)W)) return 0;
    const float hf = floorf(h), wf = floorf(w);
    const int hl = (int)hf, wl = (int)wf, hh = hl + 1, wh = wl + 1;
    const float lh = h - hf, lw = w - wf, uh = 1.f - lh, uw = 1.f - lw;
    const float wt[4] = {uh * uw, uh * lw, lh * uw, lh * lw};
    const int hc[4] = {hl, hl, hh, hh}, wc[4] = {wl, wh, wl, wh};
    int n = 0;
    for (int i = 0; i < 4; ++i)
      if (hc[i] >= 0 && hc[i] <= H - 1 && wc[i] >= 0 && wc[i] <= W - 1 && wt[i] != 0.f) {
        qs[n] = hc[i] * W + wc[i];
        ws[n] = wt[i];
        ++n;
      }
    return n;
  };
  for (long long i = 0; i <= nrows; ++i) rowptr_host[i] = 0;
  const int npix = Ho * Wo;
  int qs[4];
  float ws[4];
  for (int k = 0; k < KK; ++k)
    for (int p = 0; p < npix; ++p) {
      const int n = corners(k, p, qs, ws);
      for (int i = 0; i < n; ++i) rowptr_host[(long long)k * HW + qs[i] + 1]++;
    }
  for (long long i = 0; i < nrows; ++i) rowptr_host[i + 1] += rowptr_host[i];
  *n_entries = rowptr_host[nrows];
  // fill (rows are filled in ascending p: deterministic summation order); use the row starts as moving cursors
  for (int k = 0; k < KK; ++k)
    for (int p = 0; p < npix; ++p) {
      const int n = corners(k, p, qs, ws);
      for (int i = 0; i < n; ++i) {
        const int32_t at = rowptr_host[(long long)k * HW + qs[i]]++;
        entries_host[2 * (long long)at] = p;
        int32_t bits;
        memcpy(&bits, &ws[i], 4);
        entries_host[2 * (long long)at + 1] = bits;
      }
    }
  for (long long i = nrows; i > 0; --i) rowptr_host[i] = rowptr_host[i - 1];  // undo the cursor shift
  rowptr_host[0] = 0;
  return MODE_OK;
}

static int bwd_data_adj_impl(const float* gy, const float* w, float* gx, float* wpack, const int32_t* adj_rowptr,
                             const int32_t* adj_entries, int B, int Ci, int H, int W, int Co, int Kh, int Kw, int Ho, int Wo, int groups,
                             int accumulate, const int32_t* tile_list, int n_list, mode_stream_t stream);

extern "C" int mode_sphere_conv_bwd_data_adj(const float* gy, const float* w, float* gx, float* wpack, const int32_t* adj_rowptr,
                                             const int32_t* adj_entries, int B, int Ci, int H, int W, int Co, int Kh, int Kw,
                                             int Ho, int Wo, int groups, int accumulate, mode_stream_t stream) {
  return bwd_data_adj_impl(gy, w, gx, wpack, adj_rowptr, adj_entries, B, Ci, H, W, Co, Kh, Kw, Ho, Wo, groups, accumulate, nullptr, 0, stream);
}

// The same on a LIST of 64-pixel tiles (tile t = linear input pixels 64 t .. 64 t + 63 of the H x W image as stored): the input-gradient
// tiles the windowed split kernel does not take (mode_sphere_adjplan_build).  3x3 kernels only.
extern "C" int mode_sphere_conv_bwd_data_adj_list(const float* gy, const float* w, float* gx, float* wpack, const int32_t* adj_rowptr,
                                                  const int32_t* adj_entries, int B, int Ci, int H, int W, int Co, int Kh, int Kw,
                                                  int Ho, int Wo, int groups, int accumulate, const int32_t* tile_list, int n_list,
                                                  mode_stream_t stream) {
  MODE_REQUIRE(Kh * Kw == 9, MODE_ERR_UNSUPPORTED, "mode_sphere_conv_bwd_data_adj_list: 3x3 kernels only");
  MODE_REQUIRE(n_list >= 0 && (n_list == 0 || tile_list), MODE_ERR_BAD_ARG, "mode_sphere_conv_bwd_data_adj_list: bad tile list");
  if (n_list == 0) return MODE_OK;
  return bwd_data_adj_impl(gy, w, gx, wpack, adj_rowptr, adj_entries, B, Ci, H, W, Co, Kh, Kw, Ho, Wo, groups, accumulate, tile_list, n_list,
                           stream);
}

static int bwd_data_adj_impl(const float* gy, const float* w, float* gx, float* wpack, const int32_t* adj_rowptr,
                             const int32_t* adj_entries, int B, int Ci, int H, int W, int Co, int Kh, int Kw, int Ho, int Wo, int groups,
                             int accumulate, const int32_t* tile_list, int n_list, mode_stream_t stream) {
  Dims d;
  int rc = make_dims(d, B, Ci, H, W, Co, Kh, Kw, 1, 1, Ho, Wo, groups, "mode_sphere_conv_bwd_data_adj");
  if (rc != MODE_OK) return rc;
  d.accumulate = accumulate ? 1 : 0;
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(gy && w && gx && wpack && adj_rowptr && adj_entries, MODE_ERR_BAD_ARG, "mode_sphere_conv_bwd_data_adj: null pointer");
  hipStream_t st = mode::as_stream(stream);
  const int MTc = mode::cdiv(d.Cig, 32);
  const int NCHo = mode::cdiv(d.Cog, CCH);
  const long long npack = (long long)d.G * MTc * NCHo * d.KK * 256;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w_adj, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, d, MTc, NCHo);
  const int qtiles = tile_list ? n_list : mode::cdiv((long long)H * W, P);
  const dim3 grid(B * qtiles, mode::cdiv(MTc, 4), d.G);
  if (d.KK == 9) {
    const size_t lds = (size_t)9 * P * 40 + 2 * (size_t)CCH * 9 * P * 4;
    rc = mode::allow_lds(sphere_bwd_data_adj9_kernel, lds, "mode_sphere_conv_bwd_data_adj");
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL(sphere_bwd_data_adj9_kernel, grid, dim3(NTHREADS), lds, st, gy, adj_rowptr,
                       reinterpret_cast<const int2*>(adj_entries), reinterpret_cast<const float4*>(wpack), gx, d, MTc, NCHo, qtiles, tile_list);
  } else {
    const size_t lds = (size_t)d.KK * P * 8 + 2 * (size_t)CCH * d.KK * P * 4;
    rc = mode::allow_lds(sphere_bwd_data_adj_kernel, lds, "mode_sphere_conv_bwd_data_adj");
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL(sphere_bwd_data_adj_kernel, grid, dim3(NTHREADS), lds, st, gy, adj_rowptr,
                       reinterpret_cast<const int2*>(adj_entries), reinterpret_cast<const float4*>(wpack), gx, d, MTc, NCHo, qtiles);
  }
  return mode::check_launch("mode_sphere_conv_bwd_data_adj");
}
