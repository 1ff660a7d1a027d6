// Weight gradient of the regular 3x3 convolutions of the 2-D feature extractor (reference: nn.Conv2d inside convbn,
// models/submodule.py:15-17 -- stride 1, padding = dilation in {1, 2}, no bias; 31 of the extractor's 39 Conv2d layers).
//
// The vendor library runs this gradient as an NHWC implicit GEMM between two layout transposes (64 TFLOP/s effective at the
// benchmark shapes, 8 ms of a step); forward and input gradient stay on its fp32 Winograd kernels, which beat any direct
// convolution.  Here:   gW[o][c][kh][kw] = sum_{b,h,w} gy[b,o,h,w] * x[b,c,h+(kh-1)*dil, w+(kw-1)*dil]
//   D[i = o][j = c] per tap on v_mfma_f32_32x32x2_f32;  A[i = o][k = pixel] = gy tile, B[k = pixel][j = c] = x tile shifted by
//   the tap, both in LDS with odd channel strides (conflict-free fragments).  A workgroup owns a 32 x 32 (o, c) block and a
//   slice of the 4 x 32 pixel tiles; wave v takes row v of the tile (16 k-steps) for ALL nine taps (9 accumulators), so one A
//   fragment feeds 9 MFMAs.  The next tile travels global -> registers under the MFMAs (same pipeline as conv3d.hip: loads
//   unconditional from clamped addresses, masks at the LDS store, operands of k-step n+1 read before the MFMAs of step n).
//   The four waves' sums are combined through LDS at the end; split-K partials are reduced in a fixed order: deterministic.
#include "common.h"
#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 256;
constexpr int TH = 4;  // tile rows (one per wave)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct W2Dims {
  int B, Ci, Co, H, W;
  int nHt, nWt, T, S, MTo, MTc;
};

template <int DIL>
struct Geo {
  static constexpr int XR = TH + 2 * DIL, XW = 32 + 2 * DIL;
  static constexpr int XPLANE = (XR * XW) | 1;
  static constexpr int GPLANE = TH * 32 + 1;
  static constexpr size_t LDS = (size_t)(32 * XPLANE + 32 * GPLANE) * sizeof(float);
};

template <int DIL>
__global__ __launch_bounds__(NT, 2) void conv2d_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                  float* __restrict__ part, W2Dims d) {
  using G = Geo<DIL>;
  constexpr int XR = G::XR, XW = G::XW, XPLANE = G::XPLANE, GPLANE = G::GPLANE;
  constexpr int NXM = 4 * XR;                              // interior items per thread: channel hwv + 8 q, row r
  constexpr int NHALO = 32 * XR * 2 * DIL;                 // halo-column elements of the tile
  constexpr int NXH = (NHALO + NT - 1) / NT;               // per thread
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xl = lds;                 // [32][XPLANE]
  float* gl = lds + 32 * XPLANE;   // [32][GPLANE]
  const int s = blockIdx.x, ob = blockIdx.y, cb = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int hwv = tid >> 5, l32 = tid & 31;
  const int HW = d.H * d.W;  // (host guarantees max(Ci, Co) * H * W < 2^29)

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x16){0};

  float px[NXM], ph[NXH], pg[16];
  unsigned chan_ok = 0, gchan_ok = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    chan_ok |= (cb * 32 + hwv + 8 * q < d.Ci ? 1u : 0u) << q;
    gchan_ok |= (ob * 32 + hwv + 8 * q < d.Co ? 1u : 0u) << q;
  }

  // halo element e of a thread: (channel, row, j) with j = 0..2*DIL-1 -> tile column j (left) or 32 + j (right)
  auto halo_geom = [&](int k, int& c, int& r, int& col) {
    const int e = k * NT + tid;
    const int cr = e / (2 * DIL), j = e - cr * (2 * DIL);
    c = cr / XR;
    r = cr - c * XR;
    col = j < DIL ? j : 32 + j;
    return e < NHALO;
  };

  auto prefetch = [&](int tt, int& h0, int& w0) {
    int t = tt;
    const int wt = t % d.nWt;
    t /= d.nWt;
    const int ht = t % d.nHt;
    const int b = t / d.nHt;
    w0 = wt * 32;
    h0 = ht * TH;
    const float* xb = x + ((long long)b * d.Ci + cb * 32) * (long long)HW;
    const float* gb = gy + ((long long)b * d.Co + ob * 32) * (long long)HW;
    const int gw = w0 + l32;
    const bool wok = gw < d.W;
#pragma unroll
    for (int jj = 0; jj < NXM; ++jj) {
      const int q = jj / XR, r = jj % XR;
      const int gh = h0 + r - DIL;
      const bool ok = ((chan_ok >> q) & 1) && wok && gh >= 0 && gh < d.H;
      px[jj] = xb[(unsigned)(ok ? (hwv + 8 * q) * HW + gh * d.W + gw : 0)];
    }
#pragma unroll
    for (int k = 0; k < NXH; ++k) {
      int c, r, col;
      const bool in = halo_geom(k, c, r, col);
      const int gh = h0 + r - DIL, gwh = w0 + col - DIL;
      const bool ok = in && cb * 32 + c < d.Ci && gh >= 0 && gh < d.H && gwh >= 0 && gwh < d.W;
      ph[k] = xb[(unsigned)(ok ? c * HW + gh * d.W + gwh : 0)];
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int q = j >> 2, r = j & 3;  // output channel hwv + 8 q, tile row r
      const int gh = h0 + r;
      const bool ok = ((gchan_ok >> q) & 1) && wok && gh < d.H;
      pg[j] = gb[(unsigned)(ok ? (hwv + 8 * q) * HW + gh * d.W + gw : 0)];
    }
  };
  auto store = [&](int h0, int w0) {
    const bool wok = w0 + l32 < d.W;
    float* xrow = xl + hwv * XPLANE + DIL + l32;
#pragma unroll
    for (int jj = 0; jj < NXM; ++jj) {
      const int q = jj / XR, r = jj % XR;
      const int gh = h0 + r - DIL;
      const bool ok = ((chan_ok >> q) & 1) && wok && gh >= 0 && gh < d.H;
      xrow[q * 8 * XPLANE + r * XW] = ok ? px[jj] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NXH; ++k) {
      int c, r, col;
      const bool in = halo_geom(k, c, r, col);
      const int gh = h0 + r - DIL, gwh = w0 + col - DIL;
      const bool ok = cb * 32 + c < d.Ci && gh >= 0 && gh < d.H && gwh >= 0 && gwh < d.W;
      if (in) xl[c * XPLANE + r * XW + col] = ok ? ph[k] : 0.f;
    }
    float* grow = gl + hwv * GPLANE + l32;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int q = j >> 2, r = j & 3;
      const bool ok = ((gchan_ok >> q) & 1) && wok && h0 + r < d.H;
      grow[q * 8 * GPLANE + r * 32] = ok ? pg[j] : 0.f;
    }
  };

  // XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs; a contiguous range of tiles per XCD lets
  // neighbouring tiles find their shared halo rows in that XCD's L2
  const int s_x = xcd_remap(s, d.S);
  int h0 = 0, w0 = 0, nh0 = 0, nw0 = 0;
  if (s_x < d.T) prefetch(s_x, nh0, nw0);
  const float* ap = gl + (lane & 31) * GPLANE + wave * 32 + (lane >> 5);
  const float* bp = xl + (lane & 31) * XPLANE + wave * XW + (lane >> 5);
  for (int tt = s_x; tt < d.T; tt += d.S) {
    h0 = nh0;
    w0 = nw0;
    store(h0, w0);
    __syncthreads();
    if (tt + d.S < d.T) prefetch(tt + d.S, nh0, nw0);
    __builtin_amdgcn_sched_barrier(0);
    float a_n = ap[0], b_n[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) b_n[t] = bp[(t / 3) * DIL * XW + (t % 3) * DIL];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float a = a_n;
      float bb[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) bb[t] = b_n[t];
      if (ks + 1 < 16) {
        a_n = ap[2 * (ks + 1)];
#pragma unroll
        for (int t = 0; t < 9; ++t) b_n[t] = bp[2 * (ks + 1) + (t / 3) * DIL * XW + (t % 3) * DIL];
      }
      __builtin_amdgcn_sched_barrier(0);  // (the compiler would sink these reads to their first use)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = mfma32(a, bb[t], acc[t]);
    }
    __syncthreads();
  }

  // combine the four waves (rows of the tile) through LDS, three taps per round (48 KB), then one partial per workgroup
  float* red = lds;  // [3 taps][4 waves][1024]
  float* pb = part + (((long long)s * d.MTo + ob) * d.MTc + cb) * (9 * 1024);
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3) {
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        red[(u * 4 + wave) * 1024 + i * 32 + (lane & 31)] = acc[t3 * 3 + u][q];
      }
    __syncthreads();
    for (int idx = tid; idx < 3 * 1024; idx += NT) {
      const int u = idx >> 10, e = idx & 1023;
      const float* r4 = red + u * 4096 + e;
      pb[(t3 * 3 + u) * 1024 + e] = (r4[0] + r4[1024]) + (r4[2048] + r4[3072]);
    }
    __syncthreads();
  }
}

// gw[o][c][tap] (+)= sum_s part[s][o/32][c/32][tap][o%32][c%32].  A block of 256 threads handles 32 consecutive elements of the
// partial layout: thread (e, g) sums the slices s = g, g+8, ... in order (coalesced 128-byte reads per slice), the 8 group sums
// are combined through LDS in a fixed association -- deterministic, and 8x the parallelism of one thread per element (the split
// count is 128-512 and an element per thread walked them as one dependent chain).
__global__ __launch_bounds__(256) void reduce_gw2d(const float* __restrict__ part, float* __restrict__ gw, W2Dims d, int accumulate) {
  __shared__ float sh[8][33];
  const long long stride = (long long)d.MTo * d.MTc * 9 * 1024;
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long long e = (long long)blockIdx.x * 32 + el;
  float a0 = 0.f, a1 = 0.f;
  if (e < stride) {
    const float* p = part + e;
    int s = g;
    for (; s + 8 < d.S; s += 16) {
      a0 += p[(long long)s * stride];
      a1 += p[(long long)(s + 8) * stride];
    }
    if (s < d.S) a0 += p[(long long)s * stride];
  }
  sh[g][el] = a0 + a1;
  __syncthreads();
  if (g == 0 && e < stride) {
    const float sum = ((sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el])) + ((sh[4][el] + sh[5][el]) + (sh[6][el] + sh[7][el]));
    const int j = (int)(e & 31), i = (int)((e >> 5) & 31);
    long long r = e >> 10;
    const int tap = (int)(r % 9);
    r /= 9;
    const int cb = (int)(r % d.MTc), ob = (int)(r / d.MTc);
    const int o = ob * 32 + i, c = cb * 32 + j;
    if (o < d.Co && c < d.Ci) {
      float* q = gw + ((long long)o * d.Ci + c) * 9 + tap;
      *q = accumulate ? *q + sum : sum;
    }
  }
}

void make_dims(W2Dims& d, int B, int Ci, int H, int W, int Co) {
  d.B = B; d.Ci = Ci; d.Co = Co; d.H = H; d.W = W;
  d.nHt = mode::cdiv(H, TH);
  d.nWt = mode::cdiv(W, 32);
  d.T = B * d.nHt * d.nWt;
  d.MTo = mode::cdiv(Co, 32);
  d.MTc = mode::cdiv(Ci, 32);
  int S = mode::cdiv(2 * kNumCU, d.MTo * d.MTc);
  if (S > d.T) S = d.T;
  d.S = S < 1 ? 1 : S;
}

template <int DIL>
int launch(const float* gy, const float* x, float* workspace, const W2Dims& d, hipStream_t st, const char* who) {
  const size_t lds = std::max(Geo<DIL>::LDS, (size_t)3 * 4 * 1024 * sizeof(float));  // tile buffers, reused by the final cross-wave sum
  int rc = mode::allow_lds(conv2d_bwd_weight_kernel<DIL>, lds, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(conv2d_bwd_weight_kernel<DIL>, dim3(d.S, d.MTo, d.MTc), dim3(NT), lds, st, gy, x, workspace, d);
  return mode::check_launch(who);
}

}  // namespace

extern "C" size_t mode_conv2d_bwd_weight_workspace_bytes(int B, int Ci, int H, int W, int Co) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || H <= 0 || W <= 0) return 0;
  W2Dims d;
  make_dims(d, B, Ci, H, W, Co);
  return (size_t)d.S * d.MTo * d.MTc * 9 * 1024 * sizeof(float);
}

extern "C" int mode_conv2d_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int H, int W, int Co,
                                      int dilation, int accumulate, mode_stream_t stream) {
  const char* who = "mode_conv2d_bwd_weight";
  MODE_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(dilation == 1 || dilation == 2, MODE_ERR_UNSUPPORTED, "%s: dilation %d not implemented (1 or 2)", who, dilation);
  MODE_REQUIRE((long long)std::max(Ci, Co) * H * W < (1ll << 29), MODE_ERR_UNSUPPORTED, "%s: a sample larger than 2^29 elements", who);
  hipStream_t st = mode::as_stream(stream);
  if (B == 0) {
    if (!accumulate) return mode::zero_floats(gw, (size_t)Co * Ci * 9, st, "mode_conv2d_bwd_weight");
    return MODE_OK;
  }
  MODE_REQUIRE(gy && x && gw && workspace, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  W2Dims d;
  make_dims(d, B, Ci, H, W, Co);
  int rc = dilation == 1 ? launch<1>(gy, x, workspace, d, st, who) : launch<2>(gy, x, workspace, d, st, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(reduce_gw2d, dim3(mode::cdiv((long long)d.MTo * d.MTc * 9 * 1024, 32)), dim3(256), 0, st, workspace, gw, d, accumulate);
  return mode::check_launch("mode_conv2d_bwd_weight(reduce)");
}

static int conv2d_bwd_weight_split(const char* who, const float* gy, const float* x, const float* amax_g, const float* amax_x, float* gw,
                                   float* workspace, int B, int Ci, int H, int W, int Co, int dilation, int accumulate, mode_stream_t stream);

extern "C" int mode_conv2d_bwd_weight_split(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int H, int W, int Co,
                                            int dilation, int accumulate, mode_stream_t stream) {
  return conv2d_bwd_weight_split("mode_conv2d_bwd_weight_split", gy, x, nullptr, nullptr, gw, workspace, B, Ci, H, W, Co, dilation, accumulate,
                                 stream);
}

// The same on the two-piece fp16 arithmetic (mode_conv2d_fwd_split_f16): amax_g / amax_x = the maximum buffers of gy and of x.
extern "C" int mode_conv2d_bwd_weight_split_f16(const float* gy, const float* x, const float* amax_g, const float* amax_x, float* gw,
                                                float* workspace, int B, int Ci, int H, int W, int Co, int dilation, int accumulate,
                                                mode_stream_t stream) {
  MODE_REQUIRE(amax_g && amax_x, MODE_ERR_BAD_ARG, "mode_conv2d_bwd_weight_split_f16: null maximum");
  return conv2d_bwd_weight_split("mode_conv2d_bwd_weight_split_f16", gy, x, amax_g, amax_x, gw, workspace, B, Ci, H, W, Co, dilation, accumulate,
                                 stream);
}

static int conv2d_bwd_weight_split(const char* who, const float* gy, const float* x, const float* amax_g, const float* amax_x, float* gw,
                                   float* workspace, int B, int Ci, int H, int W, int Co, int dilation, int accumulate, mode_stream_t stream) {
  MODE_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(dilation == 1 || dilation == 2, MODE_ERR_UNSUPPORTED, "%s: dilation %d not implemented (1 or 2)", who, dilation);
  MODE_REQUIRE((long long)std::max(Ci, Co) * H * W < (1ll << 29), MODE_ERR_UNSUPPORTED, "%s: a sample larger than 2^29 elements", who);
  hipStream_t st = mode::as_stream(stream);
  if (B == 0) {
    if (!accumulate) return mode::zero_floats(gw, (size_t)Co * Ci * 9, st, "mode_conv2d_bwd_weight");
    return MODE_OK;
  }
  MODE_REQUIRE(gy && x && gw && workspace, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  W2Dims d;
  make_dims(d, B, Ci, H, W, Co);
  mode::Wgrad2SplitDims q;
  q.Ci = Ci; q.Co = Co; q.H = H; q.W = W;
  q.nWt = d.nWt; q.MTo = d.MTo; q.MTc = d.MTc;
  q.nGroups = mode::cdiv(H, 4);
  // one workgroup per CU and (o, c) block pair; a unit starts with an un-pipelined prologue, so runs as long as the image allows
  // while every workgroup still gets a unit
  int S = mode::cdiv(kNumCU, d.MTo * d.MTc);
  int run = q.nGroups;
  while (run > 4 && (long long)B * d.nWt * mode::cdiv(q.nGroups, run) < S) run = mode::cdiv(run, 2);
  q.run_groups = run;
  q.nRun = mode::cdiv(q.nGroups, run);
  q.units = B * d.nWt * q.nRun;
  if (S > q.units) S = q.units;
  if (S > d.S) S = d.S;  // never more slices than the workspace query assumed
  q.S = d.S = S;
  int rc = mode::conv2d_bww_split_launch(gy, x, workspace, q, dilation, st, who, amax_g, amax_x);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(reduce_gw2d, dim3(mode::cdiv((long long)d.MTo * d.MTc * 9 * 1024, 32)), dim3(256), 0, st, workspace, gw, d, accumulate);
  return mode::check_launch("mode_conv2d_bwd_weight_split(reduce)");
}
