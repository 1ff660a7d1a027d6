// 3x3x3 convolution (pad 1) for the 3D regulariser of MODE's disparity stage, gfx950 / fp32 MFMA.
//
// Reference: the stock nn.Conv3d layers inside convbn_3d (models/submodule.py:20-22) as used by dres0/dres1,
// the hourglasses and the classifiers (models/mode_disparity.py:11-46, 66-80) -- cuDNN NCDHW fp32 there.
//
// Implicit GEMM on v_mfma_f32_32x32x2_f32 with NCDHW kept as the HBM layout (W is contiguous, so one MFMA column
// tile = 32 consecutive w):
//     y[o, (d,h,w)] = sum_{tap, c} W[o, c, tap] * x[c, d+kd-1, h+kh-1, w+kw-1]        D[i = o][j = w]
//   A[i = o][k = c]   : weights, pre-packed in MFMA fragment order (one float4 = 4 channel pairs of one tap), read
//                       straight from global/L2 -- 27 KB..110 KB per layer, shared by every workgroup;
//   B[k = c][j = w]   : an input tile with halo staged in LDS, [8 channels][TD+2][TH+2][34]; lanes 0..31 of a fragment
//                       read 32 consecutive floats (conflict-free), lanes 32..63 the next channel plane.
// A workgroup (4 waves) owns TD x TH output rows of 32 voxels; each wave owns TD*TH/4 rows x all output-channel tiles.
// Input channels are streamed through LDS in chunks of 8.  Backward-data of a stride-1 convolution is the same kernel
// on gy with the weights transposed and flipped (done by the packing kernel).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;
constexpr int CCH = 8;  // input channels per LDS chunk
constexpr int IW = 34;  // 32 + halo

struct CDims {
  int B, Ci, Co, D, H, W;  // stride-1: output dims == input dims
  int nWt, nHt, nDt;       // tiles per axis
  int MT, NCHUNK;
  int ntiles;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// wp[((mt*NCHUNK + ch)*27 + tap)*64 + lane][cp] = Wsrc(o = mt*32 + (lane&31), c = ch*8 + 2*cp + (lane>>5), tap)
//   flip == 0: Wsrc(o,c,tap) = w[o][c][tap]            (forward; w is (Co,Ci,3,3,3), rows = Co, K = Ci)
//   flip == 1: Wsrc(o,c,tap) = w[c][o][26 - tap]       (backward-data: rows = Ci of the conv, K = Co)
__global__ void pack_w3d(const float* __restrict__ w, float* __restrict__ wp, int rows, int K, int MT, int NCHUNK, int flip) {
  const long long total = (long long)MT * NCHUNK * 27 * 64 * 4;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int cp = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    long long r = idx >> 8;
    const int tap = (int)(r % 27);
    r /= 27;
    const int ch = (int)(r % NCHUNK);
    const int mt = (int)(r / NCHUNK);
    const int o = mt * 32 + (lane & 31);
    const int c = ch * CCH + 2 * cp + (lane >> 5);
    float v = 0.f;
    if (o < rows && c < K) v = flip ? w[((long long)c * rows + o) * 27 + (26 - tap)] : w[((long long)o * K + c) * 27 + tap];
    wp[idx] = v;
  }
}

// XCD-aware bijective remap: consecutive block ids round-robin over the 8 XCDs; give each XCD a contiguous
// range of tiles so that neighbouring tiles (shared halos) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
  const int q = n / kNumXCD, r = n % kNumXCD;
  const int xcd = bid % kNumXCD, k = bid / kNumXCD;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int MT, int TD, int TH>
__global__ __launch_bounds__(NT) void conv3d_s1_kernel(const float* __restrict__ x, const float4* __restrict__ wp,
                                                       float* __restrict__ y, CDims d) {
  constexpr int R = TD * TH / 4;  // output rows per wave
  constexpr int ID = TD + 2, IH = TH + 2;
  constexpr int PLANE = ID * IH * IW;
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [CCH][PLANE]

  int t = xcd_remap(blockIdx.x, d.ntiles);
  const int wt = t % d.nWt;
  t /= d.nWt;
  const int ht = t % d.nHt;
  t /= d.nHt;
  const int dt = t % d.nDt;
  const int b = t / d.nDt;
  const int w0 = wt * 32, h0 = ht * TH, d0 = dt * TD;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const long long HW = (long long)d.H * d.W;
  const long long DHW = (long long)d.D * HW;

  f32x16 acc[MT][R];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[m][r] = (f32x16){0};

  int rowoff[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = wave * R + r;
    rowoff[r] = (row / TH) * (IH * IW) + (row % TH) * IW;
  }
  const float* bbase = tile + (lane >> 5) * PLANE + (lane & 31);
  const float* xb = x + (long long)b * d.Ci * DHW;

  for (int ch = 0; ch < d.NCHUNK; ++ch) {
    // ---- stage 8 input channels of the haloed tile (zero padding outside the volume)
    for (int idx = tid; idx < CCH * PLANE; idx += NT) {
      const int c = idx / PLANE;
      int rem = idx - c * PLANE;
      const int dz = rem / (IH * IW);
      rem -= dz * (IH * IW);
      const int hy = rem / IW;
      const int wx = rem - hy * IW;
      const int gd = d0 + dz - 1, gh = h0 + hy - 1, gw = w0 + wx - 1;
      const int cin = ch * CCH + c;
      float v = 0.f;
      if (cin < d.Ci && gd >= 0 && gd < d.D && gh >= 0 && gh < d.H && gw >= 0 && gw < d.W)
        v = xb[cin * DHW + gd * HW + gh * d.W + gw];
      tile[idx] = v;
    }
    __syncthreads();
    // ---- 27 taps x 4 channel pairs x R rows x MT tiles of MFMA
    const float4* wq = wp + ((long long)ch * 27) * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int toff = (tap / 9) * (IH * IW) + ((tap / 3) % 3) * IW + (tap % 3);
      float4 a4[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a4[m] = wq[((long long)m * d.NCHUNK * 27 + tap) * 64];
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float bv = bbase[2 * cp * PLANE + toff + rowoff[r]];
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float av = cp == 0 ? a4[m].x : cp == 1 ? a4[m].y : cp == 2 ? a4[m].z : a4[m].w;
            acc[m][r] = mfma32(av, bv, acc[m][r]);
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: D[i = o][j = w]
  float* yb = y + (long long)b * d.Co * DHW;
  const int gw = w0 + (lane & 31);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = wave * R + r;
    const int gd = d0 + row / TH, gh = h0 + row % TH;
    if (gd < d.D && gh < d.H && gw < d.W) {
      const long long sp = gd * HW + (long long)gh * d.W + gw;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int o = m * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          if (o < d.Co) yb[o * DHW + sp] = acc[m][r][q];
        }
    }
  }
}

template <int MT, int TD, int TH>
int launch_s1(const float* x, const float* wpack, float* y, CDims d, hipStream_t st, const char* who) {
  d.nWt = mode::cdiv(d.W, 32);
  d.nHt = mode::cdiv(d.H, TH);
  d.nDt = mode::cdiv(d.D, TD);
  d.ntiles = d.B * d.nDt * d.nHt * d.nWt;
  const size_t lds = (size_t)CCH * (TD + 2) * (TH + 2) * IW * sizeof(float);
  int rc = mode::allow_lds(conv3d_s1_kernel<MT, TD, TH>, lds, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL((conv3d_s1_kernel<MT, TD, TH>), dim3(d.ntiles), dim3(NT), lds, st, x, reinterpret_cast<const float4*>(wpack),
                     y, d);
  return mode::check_launch(who);
}

// rows = output channels of THIS GEMM (Co for forward, Ci for backward-data), K = its reduction channels.
int conv3d_s1(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int D, int H, int W, int flip,
              hipStream_t st, const char* who) {
  CDims d;
  d.B = B; d.Ci = K; d.Co = rows; d.D = D; d.H = H; d.W = W;
  d.MT = mode::cdiv(rows, 32);
  d.NCHUNK = mode::cdiv(K, CCH);
  MODE_REQUIRE(d.MT <= 2, MODE_ERR_UNSUPPORTED, "%s: more than 64 output channels (%d) not supported", who, rows);
  const long long npack = (long long)d.MT * d.NCHUNK * 27 * 256;
  hipLaunchKernelGGL(pack_w3d, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, rows, K, d.MT, d.NCHUNK, flip);
  // tile choice: keep >= 2 workgroups per CU worth of tiles if possible
  const long long big = (long long)B * mode::cdiv(D, 2) * mode::cdiv(H, 8) * mode::cdiv(W, 32);
  const long long mid = (long long)B * mode::cdiv(D, 2) * mode::cdiv(H, 4) * mode::cdiv(W, 32);
  if (d.MT == 1) {
    if (big >= 2 * kNumCU) return launch_s1<1, 2, 8>(x, wpack, y, d, st, who);
    if (mid >= 2 * kNumCU) return launch_s1<1, 2, 4>(x, wpack, y, d, st, who);
    return launch_s1<1, 1, 4>(x, wpack, y, d, st, who);
  }
  if (big >= 2 * kNumCU) return launch_s1<2, 2, 8>(x, wpack, y, d, st, who);
  if (mid >= 2 * kNumCU) return launch_s1<2, 2, 4>(x, wpack, y, d, st, who);
  return launch_s1<2, 1, 4>(x, wpack, y, d, st, who);
}

int check_conv_args(const void* a, const void* b, const void* c, const void* wp, int B, int Ci, int D, int H, int W, int Co,
                    int stride, const char* who) {
  MODE_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && D > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(stride == 1, MODE_ERR_UNSUPPORTED, "%s: stride %d not implemented (only 1)", who, stride);
  MODE_REQUIRE((long long)Ci * D * H * W < (1ll << 31) && (long long)Co * D * H * W < (1ll << 31), MODE_ERR_UNSUPPORTED,
               "%s: a sample larger than 2^31 elements", who);
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(a && b && c && wp, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  return MODE_OK;
}

}  // namespace

extern "C" size_t mode_conv3d_wpack_bytes(int Ci, int Co) {
  if (Ci <= 0 || Co <= 0) return 0;
  const size_t f = (size_t)mode::cdiv(Co, 32) * mode::cdiv(Ci, CCH) * 27 * 256;
  const size_t b = (size_t)mode::cdiv(Ci, 32) * mode::cdiv(Co, CCH) * 27 * 256;
  return (f > b ? f : b) * sizeof(float);
}

extern "C" int mode_conv3d_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int D, int H, int W, int Co,
                               int stride, mode_stream_t stream) {
  int rc = check_conv_args(x, w, y, wpack, B, Ci, D, H, W, Co, stride, "mode_conv3d_fwd");
  if (rc != MODE_OK || B == 0) return rc;
  return conv3d_s1(x, w, y, wpack, B, Ci, Co, D, H, W, 0, mode::as_stream(stream), "mode_conv3d_fwd");
}

extern "C" int mode_conv3d_bwd_data(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int D, int H, int W,
                                    int Co, int stride, mode_stream_t stream) {
  int rc = check_conv_args(gy, w, gx, wpack, B, Ci, D, H, W, Co, stride, "mode_conv3d_bwd_data");
  if (rc != MODE_OK || B == 0) return rc;
  return conv3d_s1(gy, w, gx, wpack, B, Co, Ci, D, H, W, 1, mode::as_stream(stream), "mode_conv3d_bwd_data");
}

// =====================================================================================================================
// Backward w.r.t. the weight (stride 1):  gW[o][c][tap] = sum_{b,d,h,w} gy[b,o,d,h,w] * x[b,c,d+kd-1,h+kh-1,w+kw-1].
//   D[i = o][j = c] per tap;  A[i = o][k = voxel] = gy tile (LDS),  B[k = voxel][j = c] = x tile shifted by the tap (LDS).
// A workgroup owns a 32x32 (o,c) block and a slice of the spatial tiles (1 x 2 x 32 voxels = 32 k-steps each); its 4 waves
// share the A fragment and split the 27 taps (7,7,7,6 accumulators).  Split-K partials are reduced in a fixed order.
namespace {

constexpr int WTH = 2;                                // rows per spatial tile
constexpr int XPLANE = 3 * (WTH + 2) * IW + 1;        // 409: odd -> lanes (= channels) hit distinct banks
constexpr int GPLANE = WTH * 32 + 1;                  // 65

struct WDims {
  int B, Ci, Co, D, H, W;
  int nWt, nHt;
  int T;  // spatial tiles in total
  int S;  // split-K slices
  int MTo, MTc;
};

__global__ __launch_bounds__(NT) void conv3d_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                               float* __restrict__ part, WDims d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xl = lds;                 // [32][XPLANE]
  float* gl = lds + 32 * XPLANE;   // [32][GPLANE]
  const int s = blockIdx.x, ob = blockIdx.y, cb = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long long HW = (long long)d.H * d.W;
  const long long DHW = (long long)d.D * HW;

  f32x16 acc[7];
  int toff[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    acc[t] = (f32x16){0};
    const int tap = wave + 4 * t;  // < 27 except wave 3, t = 6
    toff[t] = (tap / 9) * ((WTH + 2) * IW) + ((tap / 3) % 3) * IW + (tap % 3);
  }
  const bool last_valid = (wave + 24) < 27;

  for (int tt = s; tt < d.T; tt += d.S) {
    int t = tt;
    const int wt = t % d.nWt;
    t /= d.nWt;
    const int ht = t % d.nHt;
    t /= d.nHt;
    const int gd0 = t % d.D;
    const int b = t / d.D;
    const int w0 = wt * 32, h0 = ht * WTH;
    const float* xb = x + ((long long)b * d.Ci + cb * 32) * DHW;
    const float* gb = gy + ((long long)b * d.Co + ob * 32) * DHW;
    for (int idx = tid; idx < 32 * (XPLANE - 1); idx += NT) {
      const int c = idx / (XPLANE - 1);
      int rem = idx - c * (XPLANE - 1);
      const int dz = rem / ((WTH + 2) * IW);
      rem -= dz * ((WTH + 2) * IW);
      const int hy = rem / IW;
      const int wx = rem - hy * IW;
      const int gd = gd0 + dz - 1, gh = h0 + hy - 1, gw = w0 + wx - 1;
      float v = 0.f;
      if (cb * 32 + c < d.Ci && gd >= 0 && gd < d.D && gh >= 0 && gh < d.H && gw >= 0 && gw < d.W)
        v = xb[c * DHW + gd * HW + gh * d.W + gw];
      xl[c * XPLANE + (idx - c * (XPLANE - 1))] = v;
    }
    for (int idx = tid; idx < 32 * WTH * 32; idx += NT) {
      const int o = idx / (WTH * 32);
      const int rem = idx - o * (WTH * 32);
      const int hy = rem / 32, wx = rem % 32;
      const int gh = h0 + hy, gw = w0 + wx;
      float v = 0.f;
      if (ob * 32 + o < d.Co && gh < d.H && gw < d.W) v = gb[o * DHW + gd0 * HW + gh * d.W + gw];
      gl[o * GPLANE + rem] = v;
    }
    __syncthreads();
    const float* ap = gl + (lane & 31) * GPLANE + (lane >> 5);
    const float* bp = xl + (lane & 31) * XPLANE + (lane >> 5);
#pragma unroll
    for (int row = 0; row < WTH; ++row) {
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const float a = ap[row * 32 + 2 * ks];
        const float* bq = bp + row * IW + 2 * ks;
#pragma unroll
        for (int t6 = 0; t6 < 6; ++t6) acc[t6] = mfma32(a, bq[toff[t6]], acc[t6]);
        if (last_valid) acc[6] = mfma32(a, bq[toff[6]], acc[6]);
      }
    }
    __syncthreads();
  }

  float* pb = part + (((long long)s * d.MTo + ob) * d.MTc + cb) * (27 * 1024);
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = wave + 4 * t;
    if (tap < 27) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        pb[tap * 1024 + i * 32 + (lane & 31)] = acc[t][q];
      }
    }
  }
}

// gw[o][c][tap] (+)= sum_s part[s][o/32][c/32][tap][o%32][c%32]
__global__ void reduce_gw3d(const float* __restrict__ part, float* __restrict__ gw, WDims d, int accumulate) {
  const long long total = (long long)d.Co * d.Ci * 27;
  const long long stride = (long long)d.MTo * d.MTc * 27 * 1024;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(idx % 27);
    long long r = idx / 27;
    const int c = (int)(r % d.Ci);
    const int o = (int)(r / d.Ci);
    const float* p = part + (((long long)(o / 32) * d.MTc + c / 32) * 27 + tap) * 1024 + (o % 32) * 32 + (c % 32);
    float sum = 0.f;
    for (int s = 0; s < d.S; ++s) sum += p[s * stride];
    gw[idx] = accumulate ? gw[idx] + sum : sum;
  }
}

void make_wdims(WDims& d, int B, int Ci, int D, int H, int W, int Co) {
  d.B = B; d.Ci = Ci; d.Co = Co; d.D = D; d.H = H; d.W = W;
  d.nWt = mode::cdiv(W, 32);
  d.nHt = mode::cdiv(H, WTH);
  d.T = B * D * d.nHt * d.nWt;
  d.MTo = mode::cdiv(Co, 32);
  d.MTc = mode::cdiv(Ci, 32);
  int S = mode::cdiv(2 * kNumCU, d.MTo * d.MTc);
  if (S > d.T) S = d.T;
  if (S < 1) S = 1;
  d.S = S;
}

}  // namespace

extern "C" size_t mode_conv3d_bwd_weight_workspace_bytes(int B, int Ci, int D, int H, int W, int Co) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  WDims d;
  make_wdims(d, B, Ci, D, H, W, Co);
  return (size_t)d.S * d.MTo * d.MTc * 27 * 1024 * sizeof(float);
}

extern "C" int mode_conv3d_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int D, int H,
                                      int W, int Co, int stride, int accumulate, mode_stream_t stream) {
  int rc = check_conv_args(gy, x, gw, workspace, B, Ci, D, H, W, Co, stride, "mode_conv3d_bwd_weight");
  if (rc != MODE_OK) return rc;
  hipStream_t st = mode::as_stream(stream);
  if (B == 0) {
    if (!accumulate) return (int)hipMemsetAsync(gw, 0, (size_t)Co * Ci * 27 * sizeof(float), st);
    return MODE_OK;
  }
  WDims d;
  make_wdims(d, B, Ci, D, H, W, Co);
  const size_t lds = (size_t)(32 * XPLANE + 32 * GPLANE) * sizeof(float);
  rc = mode::allow_lds(conv3d_bwd_weight_kernel, lds, "mode_conv3d_bwd_weight");
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(conv3d_bwd_weight_kernel, dim3(d.S, d.MTo, d.MTc), dim3(NT), lds, st, gy, x, workspace, d);
  rc = mode::check_launch("mode_conv3d_bwd_weight");
  if (rc != MODE_OK) return rc;
  const long long n = (long long)Co * Ci * 27;
  hipLaunchKernelGGL(reduce_gw3d, dim3(mode::cdiv(n, 256)), dim3(256), 0, st, workspace, gw, d, accumulate);
  return mode::check_launch("mode_conv3d_bwd_weight(reduce)");
}
