// Regular 3x3 convolution (stride 1, padding = dilation in {1, 2}, no bias) of the 2-D feature extractor (reference: nn.Conv2d
// inside convbn, models/submodule.py:15-17: firstconv[1..2], layer1-3, lastconv[1] -- 31 of its 39 Conv2d layers).  Measured at
// the step's 4 images: 90-130 TFLOP/s here against 80-112 for the vendor library (Winograd / NHWC implicit GEMM between layout
// transposes for the dilated layers); the host picks per layer (functional._conv2d_own).
//
// Same structure as conv3d.hip, one dimension down: implicit GEMM on v_mfma_f32_32x32x2_f32, D[i = o][j = 32 pixels along w];
// haloed input tile [8 channels][TH + 2 dil][32 + 2 dil] in LDS, input channels streamed in chunks of 8 with the next chunk
// travelling global -> registers under the MFMAs (half-wave = channel, item = tile row; unconditional loads from clamped
// addresses, masks at the LDS store); weights pre-packed in fragment order and requested one tap ahead.  The input gradient is the
// same kernel on gy with the weights transposed and flipped by the packing kernel.
#include "common.h"
#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 256;
constexpr int CCH = 8;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct C2Dims {
  int B, Ci, Co, H, W;
  int nHt, nWt, ntiles, NCHUNK;
};

// wp[((mt*NCHUNK + ch)*9 + tap)*64 + lane][cp] = Wsrc(o = mt*32 + (lane&31), c = ch*8 + 2*cp + (lane>>5), tap)
//   flip == 0: Wsrc(o,c,tap) = w[o][c][tap]      (forward; w is (Co,Ci,3,3), rows = Co, K = Ci)
//   flip == 1: Wsrc(o,c,tap) = w[c][o][8 - tap]  (input gradient: rows = Ci of the convolution, K = Co)
//   fold != 0: row o is scaled by the folded BatchNorm scale of `bn` and block 0 writes the shifts to wp[total + o] (common.h).
__global__ void pack_w2d(const float* __restrict__ w, float* __restrict__ wp, int rows, int K, int MT, int NCHUNK, int flip, int fold,
                         mode_bn_epilogue bn) {
  const long long total = (long long)MT * NCHUNK * 9 * 64 * 4;
  if (fold && blockIdx.x == 0)
    for (int o = threadIdx.x; o < rows; o += blockDim.x) wp[total + o] = fold_shift(bn, o);
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int cp = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int tap = (int)(r % 9);
    r /= 9;
    const int ch = (int)(r % NCHUNK);
    const int mt = (int)(r / NCHUNK);
    const int o = mt * 32 + (lane & 31);
    const int c = ch * CCH + 2 * cp + (lane >> 5);
    float v = 0.f;
    if (o < rows && c < K) v = flip ? w[((long long)c * rows + o) * 9 + 8 - tap] : w[((long long)o * K + c) * 9 + tap];
    if (fold && o < rows) v *= fold_scale(bn, o);
    wp[idx] = v;
  }
}

// Waves per SIMD the eval variant (folded-BatchNorm epilogue) is held to: the occupancy of the plain kernel.  Left alone the compiler
// spends 20-60 more registers on the epilogue's address arithmetic and drops a wave (144 vs 85 VGPRs for <4, 8>: one wave per SIMD instead of two; <2, 8> stays at three instead of four -- four spills).
__host__ __device__ constexpr int epi_waves(int MT, int TH) { return TH == 8 ? (MT == 1 ? 5 : MT == 2 ? 3 : 2) : 1; }

template <int MT, int TH, int DIL, bool EPI>
__global__ __launch_bounds__(NT, EPI ? epi_waves(MT, TH) : 1) void conv2d_kernel(const float* __restrict__ x, const float4* __restrict__ wp, float* __restrict__ y,
                                                    C2Dims d, Epi epi) {
  constexpr int R = TH / 4;  // output rows per wave
  constexpr int IH = TH + 2 * DIL, IW = 32 + 2 * DIL;
  constexpr int PLANE = (IH * IW) | 1;
  constexpr int NHALO = CCH * IH * 2 * DIL, NPH = (NHALO + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [CCH][PLANE]

  int t = xcd_remap(blockIdx.x, d.ntiles);  // neighbouring tiles (shared halos) on one XCD
  const int wt = t % d.nWt;
  t /= d.nWt;
  const int ht = t % d.nHt;
  const int b = t / d.nHt;
  const int w0 = wt * 32, h0 = ht * TH;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int hwv = tid >> 5, l32 = tid & 31;
  const int HW = d.H * d.W;  // (host guarantees max(Ci, Co) * H * W < 2^29)

  f32x16 acc[MT][R];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[m][r] = (f32x16){0};

  const float* xb = x + (long long)b * d.Ci * HW;
  // row offsets / validity of the tile rows (chunk-invariant); column validity of the interior group
  int rowoff[IH];
  unsigned rowok = 0;
#pragma unroll
  for (int r = 0; r < IH; ++r) {
    const int gh = h0 + r - DIL;
    const bool ok = gh >= 0 && gh < d.H && w0 + l32 < d.W;
    rowoff[r] = ok ? gh * d.W + w0 + l32 : 0;
    rowok |= (ok ? 1u : 0u) << r;
  }
  float vm[IH], vh[NPH];
  auto halo_geom = [&](int k, int& c, int& r, int& col) {
    const int e = k * NT + tid;
    const int cr = e / (2 * DIL), j = e - cr * (2 * DIL);
    c = cr / IH;
    r = cr - c * IH;
    col = j < DIL ? j : 32 + j;
    return e < NHALO;
  };
  auto issue = [&](int ch) {
    const float* xc = xb + (long long)(ch * CCH) * HW;
    const bool cok = ch * CCH + hwv < d.Ci;
#pragma unroll
    for (int r = 0; r < IH; ++r) vm[r] = xc[(unsigned)(cok ? hwv * HW + rowoff[r] : 0)];
#pragma unroll
    for (int k = 0; k < NPH; ++k) {
      int c, r, col;
      const bool in = halo_geom(k, c, r, col);
      const int gh = h0 + r - DIL, gw = w0 + col - DIL;
      const bool ok = in && ch * CCH + c < d.Ci && gh >= 0 && gh < d.H && gw >= 0 && gw < d.W;
      vh[k] = xc[(unsigned)(ok ? c * HW + gh * d.W + gw : 0)];
    }
  };
  auto commit = [&](int ch) {
    const bool cok = ch * CCH + hwv < d.Ci;
    float* dst = tile + hwv * PLANE + DIL + l32;
#pragma unroll
    for (int r = 0; r < IH; ++r) dst[r * IW] = (cok && ((rowok >> r) & 1)) ? vm[r] : 0.f;
#pragma unroll
    for (int k = 0; k < NPH; ++k) {
      int c, r, col;
      const bool in = halo_geom(k, c, r, col);
      const int gh = h0 + r - DIL, gw = w0 + col - DIL;
      const bool ok = ch * CCH + c < d.Ci && gh >= 0 && gh < d.H && gw >= 0 && gw < d.W;
      if (in) tile[c * PLANE + r * IW + col] = ok ? vh[k] : 0.f;
    }
  };

  const float* bbase = tile + (lane >> 5) * PLANE + (wave * R) * IW + (lane & 31);
  issue(0);
  commit(0);
  __syncthreads();
  for (int ch = 0; ch < d.NCHUNK; ++ch) {
    if (ch + 1 < d.NCHUNK) {
      issue(ch + 1);  // in flight during the MFMA phase below
      __builtin_amdgcn_sched_barrier(0);
    }
    const float4* wq = wp + ((long long)ch * 9) * 64 + lane;
    float4 a_nxt[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_nxt[m] = wq[((long long)m * d.NCHUNK * 9) * 64];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = (tap / 3) * DIL * IW + (tap % 3) * DIL;
      float4 a4[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a4[m] = a_nxt[m];
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {
        if (cp == 2) {
          if (tap + 1 < 9) {
#pragma unroll
            for (int m = 0; m < MT; ++m) a_nxt[m] = wq[((long long)m * d.NCHUNK * 9 + tap + 1) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float bv = bbase[2 * cp * PLANE + toff + r * IW];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float av = cp == 0 ? a4[m].x : cp == 1 ? a4[m].y : cp == 2 ? a4[m].z : a4[m].w;
            acc[m][r] = mfma32(av, bv, acc[m][r]);
          }
        }
      }
    }
    __syncthreads();  // every wave is done reading this chunk
    if (ch + 1 < d.NCHUNK) {
      commit(ch + 1);
      __syncthreads();
    }
  }

  float* yb = y + (long long)b * d.Co * HW;
  const int gw = w0 + (lane & 31);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int gh = h0 + wave * R + r;
    if (gh < d.H && gw < d.W) {
      const int sp = gh * d.W + gw;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int o = m * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          if (o < d.Co) {
            const long long idx = (long long)o * HW + sp;
            yb[idx] = EPI ? apply_epi(epi, acc[m][r][q], o, (long long)b * d.Co * HW + idx) : acc[m][r][q];
          }
        }
    }
  }
}

template <int MT, int TH, int DIL>
int launch(const float* x, const float* wpack, float* y, C2Dims d, hipStream_t st, const char* who, Epi epi) {
  d.nHt = mode::cdiv(d.H, TH);
  d.nWt = mode::cdiv(d.W, 32);
  d.ntiles = d.B * d.nHt * d.nWt;
  constexpr int PLANE = ((TH + 2 * DIL) * (32 + 2 * DIL)) | 1;
  const size_t lds = (size_t)CCH * PLANE * sizeof(float);
  if (epi.shift)  // eval-mode layer with the folded BatchNorm epilogue: its own instantiation
    hipLaunchKernelGGL((conv2d_kernel<MT, TH, DIL, true>), dim3(d.ntiles), dim3(NT), lds, st, x, reinterpret_cast<const float4*>(wpack), y,
                       d, epi);
  else
    hipLaunchKernelGGL((conv2d_kernel<MT, TH, DIL, false>), dim3(d.ntiles), dim3(NT), lds, st, x, reinterpret_cast<const float4*>(wpack), y,
                       d, epi);
  return mode::check_launch(who);
}

template <int DIL>
int dispatch(const float* x, const float* wpack, float* y, const C2Dims& d, int MT, hipStream_t st, const char* who, Epi epi) {
  // 8 rows per tile (two per wave) when that still gives every CU two workgroups, else 4
  const bool big = (long long)d.B * mode::cdiv(d.H, 8) * mode::cdiv(d.W, 32) >= 2 * kNumCU;
  switch (MT) {
    case 1: return big ? launch<1, 8, DIL>(x, wpack, y, d, st, who, epi) : launch<1, 4, DIL>(x, wpack, y, d, st, who, epi);
    case 2: return big ? launch<2, 8, DIL>(x, wpack, y, d, st, who, epi) : launch<2, 4, DIL>(x, wpack, y, d, st, who, epi);
    case 3: return big ? launch<3, 8, DIL>(x, wpack, y, d, st, who, epi) : launch<3, 4, DIL>(x, wpack, y, d, st, who, epi);
    default: return big ? launch<4, 8, DIL>(x, wpack, y, d, st, who, epi) : launch<4, 4, DIL>(x, wpack, y, d, st, who, epi);
  }
}

// rows = output channels of THIS GEMM (Co forward, Ci for the input gradient), K = its reduction channels
int run(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int H, int W, int dilation, int flip, hipStream_t st,
        const char* who, const mode_bn_epilogue* bn = nullptr) {
  MODE_REQUIRE(B >= 0 && K > 0 && rows > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(dilation == 1 || dilation == 2, MODE_ERR_UNSUPPORTED, "%s: dilation %d not implemented (1 or 2)", who, dilation);
  MODE_REQUIRE(rows <= 128, MODE_ERR_UNSUPPORTED, "%s: more than 128 output channels (%d) not supported", who, rows);
  MODE_REQUIRE((long long)std::max(K, rows) * H * W < (1ll << 29), MODE_ERR_UNSUPPORTED, "%s: a sample larger than 2^29 elements", who);
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(x && w && y && wpack, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  C2Dims d;
  d.B = B; d.Ci = K; d.Co = rows; d.H = H; d.W = W;
  d.NCHUNK = mode::cdiv(K, CCH);
  const int MT = mode::cdiv(rows, 32);
  const long long npack = (long long)MT * d.NCHUNK * 9 * 256;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w2d, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, rows, K, MT, d.NCHUNK, flip, bn ? 1 : 0,
                     bn ? *bn : mode_bn_epilogue());
  const Epi epi = make_epi(bn, wpack + npack);
  return dilation == 1 ? dispatch<1>(x, wpack, y, d, MT, st, who, epi) : dispatch<2>(x, wpack, y, d, MT, st, who, epi);
}

}  // namespace

extern "C" size_t mode_conv2d_wpack_bytes(int Ci, int Co) {
  if (Ci <= 0 || Co <= 0) return 0;
  const size_t f = (size_t)mode::cdiv(Co, 32) * mode::cdiv(Ci, CCH) * 9 * 256;
  const size_t b = (size_t)mode::cdiv(Ci, 32) * mode::cdiv(Co, CCH) * 9 * 256;
  size_t n = (f > b ? f : b) + 32 * (size_t)mode::cdiv(Co > Ci ? Co : Ci, 32);  // + the folded BatchNorm shifts
  n = std::max(n, std::max(mode::conv2d_split_wpack_floats(Ci, Co), mode::conv2d_split_wpack_floats(Co, Ci)));
  return n * sizeof(float);
}

extern "C" int mode_conv2d_split_supported(int Ci, int Co, int dilation, int which) {
  if (Ci <= 0 || Co <= 0) return 0;
  return which == 1 ? mode::conv2d_split_supported(Co, Ci, dilation) : mode::conv2d_split_supported(Ci, Co, dilation);
}

extern "C" int mode_conv2d_fwd_split(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci, int H,
                                     int W, int Co, int dilation, mode_stream_t stream) {
  if (bn) {
    int rc = mode::check_bn(bn, "mode_conv2d_fwd_split");
    if (rc != MODE_OK) return rc;
  }
  return mode::conv2d_split_run(x, w, y, wpack, B, Ci, Co, H, W, dilation, 0, mode::as_stream(stream), "mode_conv2d_fwd_split", bn);
}

extern "C" int mode_conv2d_bwd_data_split(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int H, int W, int Co,
                                          int dilation, mode_stream_t stream) {
  return mode::conv2d_split_run(gy, w, gx, wpack, B, Co, Ci, H, W, dilation, 1, mode::as_stream(stream), "mode_conv2d_bwd_data_split", nullptr);
}

// gx = conv^T(gy) + acc: a gradient of the same tensor that is already there, added in the store (acc must not alias gx)
extern "C" int mode_conv2d_bwd_data_split_acc(const float* gy, const float* w, const float* acc, float* gx, float* wpack, int B, int Ci, int H,
                                              int W, int Co, int dilation, mode_stream_t stream) {
  MODE_REQUIRE(acc, MODE_ERR_BAD_ARG, "mode_conv2d_bwd_data_split_acc: null acc");
  return mode::conv2d_split_run(gy, w, gx, wpack, B, Co, Ci, H, W, dilation, 1, mode::as_stream(stream), "mode_conv2d_bwd_data_split_acc", nullptr,
                                acc);
}

// The two calls of a TRAINING step on the two-piece fp16 arithmetic (conv3d_split.hip): amax_* = the maximum buffers
// (MODE_BN_ABSMAX_FLOATS floats) of the activation / gradient and of the weight; acc may be NULL.
extern "C" int mode_conv2d_fwd_split_f16(const float* x, const float* w, const float* amax_x, const float* amax_w, float* y, float* wpack, int B,
                                         int Ci, int H, int W, int Co, int dilation, mode_stream_t stream) {
  MODE_REQUIRE(amax_x && amax_w, MODE_ERR_BAD_ARG, "mode_conv2d_fwd_split_f16: null maximum");
  return mode::conv2d_split_run(x, w, y, wpack, B, Ci, Co, H, W, dilation, 0, mode::as_stream(stream), "mode_conv2d_fwd_split_f16", nullptr,
                                nullptr, amax_x, amax_w);
}

// Inference on the same arithmetic (ABI 31; mode_conv3d_fwd_split_f16_bn is the 3-D twin): folded BatchNorm (+ residual) (+ ReLU), the
// folded weights' maximum taken inside with the pack, the stored output's maximum left in amax_y for the next layer.
extern "C" int mode_conv2d_fwd_split_f16_bn(const float* x, const float* w, const float* amax_x, const mode_bn_epilogue* bn, float* y,
                                            float* amax_y, float* wpack, int B, int Ci, int H, int W, int Co, int dilation,
                                            mode_stream_t stream) {
  const char* who = "mode_conv2d_fwd_split_f16_bn";
  MODE_REQUIRE(bn, MODE_ERR_BAD_ARG, "%s: null BatchNorm epilogue", who);
  int rc = mode::check_bn(bn, who);
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(amax_x && amax_y, MODE_ERR_BAD_ARG, "%s: null maximum buffer", who);
  return mode::conv2d_split_run(x, w, y, wpack, B, Ci, Co, H, W, dilation, 0, mode::as_stream(stream), who, bn, nullptr, amax_x, nullptr, amax_y);
}

extern "C" int mode_conv2d_bwd_data_split_f16(const float* gy, const float* w, const float* amax_g, const float* amax_w, const float* acc,
                                              float* gx, float* wpack, int B, int Ci, int H, int W, int Co, int dilation,
                                              mode_stream_t stream) {
  MODE_REQUIRE(amax_g && amax_w, MODE_ERR_BAD_ARG, "mode_conv2d_bwd_data_split_f16: null maximum");
  return mode::conv2d_split_run(gy, w, gx, wpack, B, Co, Ci, H, W, dilation, 1, mode::as_stream(stream), "mode_conv2d_bwd_data_split_f16", nullptr,
                                acc, amax_g, amax_w);
}

extern "C" int mode_conv2d_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int H, int W, int Co, int dilation,
                               mode_stream_t stream) {
  return run(x, w, y, wpack, B, Ci, Co, H, W, dilation, 0, mode::as_stream(stream), "mode_conv2d_fwd");
}

extern "C" int mode_conv2d_bwd_data(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int H, int W, int Co,
                                    int dilation, mode_stream_t stream) {
  return run(gy, w, gx, wpack, B, Co, Ci, H, W, dilation, 1, mode::as_stream(stream), "mode_conv2d_bwd_data");
}

extern "C" int mode_conv2d_fwd_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci, int H,
                                  int W, int Co, int dilation, mode_stream_t stream) {
  int rc = mode::check_bn(bn, "mode_conv2d_fwd_bn");
  if (rc != MODE_OK) return rc;
  return run(x, w, y, wpack, B, Ci, Co, H, W, dilation, 0, mode::as_stream(stream), "mode_conv2d_fwd_bn", bn);
}

// ---------------------------------------------------------------------------------------------------------------------
// Zero insertion: out (P, 2*Ho, 2*Wo) with out[p][2h][2w] = in[p][h][w] and zeros elsewhere.  The two gradients of the extractor's
// one stride-2 3x3 layer (layer2[0].conv1, models/submodule.py:158) are the stride-1 gradients of the zero-inserted output
// gradient:   gx[c, p] = sum_{o,k} w[o,c,k] * up(gy)[o, p - k + 1],   gW[o,c,k] = sum_p up(gy)[o, p] * x[c, p + k - 1]
// -- 4x the arithmetic of a dedicated stride-2 form, but on the MFMA kernels above (100-117 TFLOP/s) instead of the
// gather-and-MAC kernels (12-17): 0.81 + 0.58 ms -> 0.36 + 0.38 ms per step.
namespace {
__global__ __launch_bounds__(256) void zero_insert2_kernel(const float* __restrict__ in, float* __restrict__ out, long long planes, int Ho,
                                                           int Wo) {
  const int W = 2 * Wo, W4 = W / 4;  // Wo even: a float4 of an even output row = (in[2t], 0, in[2t+1], 0)
  const long long total = planes * 2 * Ho * W4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i % W4);
    const long long r = i / W4;  // output row over all planes
    const int h = (int)(r % (2 * Ho));
    const long long p = r / (2 * Ho);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((h & 1) == 0) {
      const float2 s = *reinterpret_cast<const float2*>(in + (p * Ho + (h >> 1)) * Wo + 2 * t);
      v.x = s.x;
      v.z = s.y;
    }
    *reinterpret_cast<float4*>(out + r * W + 4 * t) = v;
  }
}
}  // namespace

extern "C" int mode_zero_insert2(const float* in, float* out, long long planes, int Ho, int Wo, mode_stream_t stream) {
  MODE_REQUIRE(planes >= 0 && Ho > 0 && Wo > 0, MODE_ERR_BAD_ARG, "mode_zero_insert2: non-positive size");
  MODE_REQUIRE(Wo % 2 == 0, MODE_ERR_UNSUPPORTED, "mode_zero_insert2: the input width must be even (got %d)", Wo);
  if (planes == 0) return MODE_OK;
  MODE_REQUIRE(in && out, MODE_ERR_BAD_ARG, "mode_zero_insert2: null pointer");
  MODE_REQUIRE(((size_t)in % 8) == 0 && ((size_t)out % 16) == 0, MODE_ERR_UNSUPPORTED, "mode_zero_insert2: unaligned tensors");
  const long long total = planes * 2 * Ho * (Wo / 2);
  const int grid = (int)std::min<long long>((total + 255) / 256, 16 * kNumCU);
  hipLaunchKernelGGL(zero_insert2_kernel, dim3(grid), dim3(256), 0, mode::as_stream(stream), in, out, planes, Ho, Wo);
  return mode::check_launch("mode_zero_insert2");
}
