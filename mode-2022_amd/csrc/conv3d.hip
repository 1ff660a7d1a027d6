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
#include <cstdlib>

#include "conv3d_internal.h"
#include "bn_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;
constexpr int CCH = 8;  // input channels per LDS chunk

struct CDims {
  int B, Ci, Co, D, H, W;  // input volume
  int Do, Ho, Wo;          // output volume: (X + 2 - 3) / stride + 1
  int nWt, nHt, nDt;       // output tiles per axis
  int MT, NCHUNK;
  int ntiles;
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// wp[((mt*NCHUNK + ch)*27 + tap)*64 + lane][cp] = Wsrc(o = mt*32 + (lane&31), c = ch*8 + 2*cp + (lane>>5), tap)
//   flip == 0: Wsrc(o,c,tap) = w[o][c][tap]            (forward; w is (Co,Ci,3,3,3), rows = Co, K = Ci)
//   flip == 1: Wsrc(o,c,tap) = w[c][o][26 - tap]       (backward-data: rows = Ci of the conv, K = Co)
//   flip == 2: Wsrc(o,c,tap) = w[c][o][tap]            (transposed conv: w is (Cin,Cout,3,3,3), rows = Cout, K = Cin)
//   fold != 0: row o is scaled by the folded BatchNorm scale of `bn` and block 0 writes the shifts to wp[total + o] (common.h).
__global__ void pack_w3d(const float* __restrict__ w, float* __restrict__ wp, int rows, int K, int MT, int NCHUNK, int flip,
                         int fold, mode_bn_epilogue bn) {
  const long long total = (long long)MT * NCHUNK * 27 * 64 * 4;
  if (fold && blockIdx.x == 0)
    for (int o = threadIdx.x; o < rows; o += blockDim.x) wp[total + o] = fold_shift(bn, o);
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
    if (o < rows && c < K)
      v = flip == 0 ? w[((long long)o * K + c) * 27 + tap] : w[((long long)c * rows + o) * 27 + (flip == 1 ? 26 - tap : tap)];
    if (fold && o < rows) v *= fold_scale(bn, o);
    wp[idx] = v;
  }
}

// Stride 2 (S = 2): the 65 input columns of a row are stored de-interleaved, [33 odd-phase | 32 even-phase], so that the
// 32 lanes of a fragment still read consecutive LDS words: tap kw = 0 -> O[j], kw = 1 -> E[j], kw = 2 -> O[j+1].
template <int S>
__device__ __forceinline__ constexpr int tap_woff(int kw) {
  return S == 1 ? kw : (kw == 0 ? 0 : kw == 1 ? 33 : 1);
}

template <int MT, int TD, int TH, int S, bool EPI>
__global__ __launch_bounds__(NT) void conv3d_kernel(const float* __restrict__ x, const float4* __restrict__ wp,
                                                    float* __restrict__ y, CDims d, Epi epi) {
  constexpr int R = TD * TH / 4;  // output rows per wave
  constexpr int ID = (TD - 1) * S + 3, IH = (TH - 1) * S + 3;
  constexpr int IW = (S == 1) ? 34 : 65;
  constexpr int PLANE = ID * IH * IW;
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [CCH][PLANE]

  // (tile order w, h, d inside an XCD's contiguous range: x is fetched 2.03 x, the depth halo of the 2-row tiles; depth-fastest
  // order was measured -- 2.26 x and the same 1.36 ms: the kernel is MFMA-bound, profiles/traffic.json)
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
  const long long oHW = (long long)d.Ho * d.Wo;
  const long long oDHW = (long long)d.Do * oHW;

  f32x16 acc[MT][R];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[m][r] = (f32x16){0};

  int rowoff[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = wave * R + r;
    rowoff[r] = (row / TH) * S * (IH * IW) + (row % TH) * S * IW;
  }
  const float* bbase = tile + (lane >> 5) * PLANE + (lane & 31);
  const float* xb = x + (long long)b * d.Ci * DHW;

  // ---- staging of the haloed input tile, one 8-channel chunk at a time, software pipelined: the loads of chunk ch+1 are
  // issued before the MFMA phase of chunk ch and written to LDS after it.  The tile is 8 channels x RPC = ID*IH rows of IW
  // floats.  Half-wave hwv owns channel hwv of the chunk: its item j is row j / NG, 32-column group j % NG -- one coalesced
  // 128-byte load per half-wave, and (row, group) are compile-time constants, so an address is rowtab[row] + column.  The 1-2
  // remaining halo columns of every row are spread over the threads.  Loads are unconditional from clamped addresses; the
  // validity is recomputed at the LDS store (a select at the load would wait for it, or keep the masks alive across the MFMAs).
  constexpr int RPC = ID * IH;
  constexpr int NROWS = CCH * RPC;
  constexpr int NG = (S == 1) ? 1 : 2;      // 32-column groups per row
  constexpr int GO = (S == 1) ? 1 : 0;      // first column of group 0
  constexpr int NHALO = (S == 1) ? 2 : 1;   // leftover columns per row: {0, 33} or {64}
  constexpr int NIT = RPC * NG;
  constexpr int NPH = (NROWS * NHALO + NT - 1) / NT;
  static_assert(NT == 32 * CCH, "one half-wave per channel of the chunk");
  int* rowtab = reinterpret_cast<int*>(tile + CCH * PLANE);  // per row: element offset of (c, gd, gh, 0) or -1
  for (int r = tid; r < NROWS; r += NT) {
    const int c = r / RPC;
    const int rem = r - c * RPC;
    const int dz = rem / IH;
    const int hy = rem - dz * IH;
    const int gd = d0 * S + dz - 1, gh = h0 * S + hy - 1;
    rowtab[r] = (gd >= 0 && gd < d.D && gh >= 0 && gh < d.H) ? (int)(c * DHW + gd * HW + gh * d.W) : -1;
  }
  __syncthreads();
  const int hwv = tid >> 5, l32 = tid & 31;
  const int* myrows = rowtab + hwv * RPC;
  float* mytile = tile + hwv * PLANE;
  float vm[NIT], vh[NPH];

  auto issue = [&](int ch) {
    const float* xc = xb + (long long)ch * CCH * DHW;
    const bool cok = ch * CCH + hwv < d.Ci;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int rem = j / NG, g = j % NG;
      const int gw = w0 * S + GO + 32 * g + l32 - 1;
      const int off = myrows[rem];
      const bool ok = cok && off >= 0 && gw >= 0 && gw < d.W;
      vm[j] = xc[(unsigned)(ok ? off + gw : 0)];
    }
#pragma unroll
    for (int k = 0; k < NPH; ++k) {
      const int item = k * NT + tid;
      const int r = item / NHALO, side = item - r * NHALO;
      const int wx = (S == 1) ? (side ? 33 : 0) : 64;
      const int gw = w0 * S + wx - 1;
      const int off = (item < NROWS * NHALO) ? rowtab[r] : -1;
      const bool ok = off >= 0 && gw >= 0 && gw < d.W && ch * CCH + r / RPC < d.Ci;
      vh[k] = xc[(unsigned)(ok ? off + gw : 0)];
    }
  };
  auto commit = [&](int ch) {
    const bool cok = ch * CCH + hwv < d.Ci;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int rem = j / NG, g = j % NG;
      const int wx = GO + 32 * g + l32;
      const int gw = w0 * S + wx - 1;
      const int lw = (S == 1) ? wx : ((wx & 1) ? 33 + (wx >> 1) : (wx >> 1));
      const bool ok = cok && myrows[rem] >= 0 && gw >= 0 && gw < d.W;
      mytile[rem * IW + lw] = ok ? vm[j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < NPH; ++k) {
      const int item = k * NT + tid;
      const int r = item / NHALO, side = item - r * NHALO;
      const int wx = (S == 1) ? (side ? 33 : 0) : 64;
      const int gw = w0 * S + wx - 1;
      const int lw = (S == 1) ? wx : 32;  // column 64 = odd-phase entry O[32] of the [O | E] split
      if (item < NROWS * NHALO) {
        const bool ok = rowtab[r] >= 0 && gw >= 0 && gw < d.W && ch * CCH + r / RPC < d.Ci;
        tile[r * IW + lw] = ok ? vh[k] : 0.f;
      }
    }
  };

  issue(0);
  commit(0);
  __syncthreads();
  for (int ch = 0; ch < d.NCHUNK; ++ch) {
    if (ch + 1 < d.NCHUNK) {
      issue(ch + 1);  // in flight during the MFMA phase below
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- 27 taps x 4 channel pairs x R rows x MT tiles of MFMA
    // weights one tap ahead: the fragment of tap t+1 is requested half-way through the MFMAs of tap t (the compiler on its
    // own requested it right before its first use and waited a full L1/L2 round trip every other tap)
    const float4* wq = wp + ((long long)ch * 27) * 64 + lane;
    float4 a_nxt[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_nxt[m] = wq[((long long)m * d.NCHUNK * 27) * 64];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int toff = (tap / 9) * (IH * IW) + ((tap / 3) % 3) * IW + tap_woff<S>(tap % 3);
      float4 a4[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a4[m] = a_nxt[m];
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {
        if (cp == 2) {
          if (tap + 1 < 27) {
#pragma unroll
            for (int m = 0; m < MT; ++m) a_nxt[m] = wq[((long long)m * d.NCHUNK * 27 + tap + 1) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
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
    __syncthreads();  // every wave is done reading this chunk
    if (ch + 1 < d.NCHUNK) {
      commit(ch + 1);
      __syncthreads();
    }
  }

  // ---- epilogue: D[i = o][j = w]
  float* yb = y + (long long)b * d.Co * oDHW;
  const int gw = w0 + (lane & 31);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = wave * R + r;
    const int gd = d0 + row / TH, gh = h0 + row % TH;
    if (gd < d.Do && gh < d.Ho && gw < d.Wo) {
      const long long sp = gd * oHW + (long long)gh * d.Wo + gw;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int o = m * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          if (o < d.Co) {
            const long long idx = o * oDHW + sp;
            yb[idx] = EPI ? apply_epi(epi, acc[m][r][q], o, (long long)b * d.Co * oDHW + idx) : acc[m][r][q];
          }
        }
    }
  }
}

template <int MT, int TD, int TH, int S>
int launch_conv(const float* x, const float* wpack, float* y, CDims d, hipStream_t st, const char* who, Epi epi) {
  d.nWt = mode::cdiv(d.Wo, 32);
  d.nHt = mode::cdiv(d.Ho, TH);
  d.nDt = mode::cdiv(d.Do, TD);
  d.ntiles = d.B * d.nDt * d.nHt * d.nWt;
  constexpr int kRows = CCH * ((TD - 1) * S + 3) * ((TH - 1) * S + 3);
  const size_t lds = (size_t)kRows * (S == 1 ? 34 : 65) * sizeof(float) + (size_t)kRows * sizeof(int);  // tile + row table
  if (epi.shift) {  // eval-mode layer with the folded BatchNorm epilogue: its own instantiation, the plain kernel is untouched
    int rc = mode::allow_lds(conv3d_kernel<MT, TD, TH, S, true>, lds, who);
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL((conv3d_kernel<MT, TD, TH, S, true>), dim3(d.ntiles), dim3(NT), lds, st, x,
                       reinterpret_cast<const float4*>(wpack), y, d, epi);
    return mode::check_launch(who);
  }
  int rc = mode::allow_lds(conv3d_kernel<MT, TD, TH, S, false>, lds, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL((conv3d_kernel<MT, TD, TH, S, false>), dim3(d.ntiles), dim3(NT), lds, st, x,
                     reinterpret_cast<const float4*>(wpack), y, d, epi);
  return mode::check_launch(who);
}

// rows = output channels of THIS GEMM (Co for forward, Ci for backward-data), K = its reduction channels.
int conv3d_s1(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int D, int H, int W, int flip,
              hipStream_t st, const char* who, const mode_bn_epilogue* bn = nullptr) {
  CDims d;
  d.B = B; d.Ci = K; d.Co = rows; d.D = D; d.H = H; d.W = W;
  d.Do = D; d.Ho = H; d.Wo = W;
  d.MT = mode::cdiv(rows, 32);
  d.NCHUNK = mode::cdiv(K, CCH);
  MODE_REQUIRE(d.MT <= 2, MODE_ERR_UNSUPPORTED, "%s: more than 64 output channels (%d) not supported", who, rows);
  const long long npack = (long long)d.MT * d.NCHUNK * 27 * 256;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w3d, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, rows, K, d.MT, d.NCHUNK, flip, bn ? 1 : 0,
                     bn ? *bn : mode_bn_epilogue());
  const Epi epi = make_epi(bn, wpack + npack);
  // tile choice: keep >= 2 workgroups per CU worth of tiles if possible
  const long long big = (long long)B * mode::cdiv(D, 2) * mode::cdiv(H, 8) * mode::cdiv(W, 32);
  const long long mid = (long long)B * mode::cdiv(D, 2) * mode::cdiv(H, 4) * mode::cdiv(W, 32);
  if (d.MT == 1) {
    if (big >= 2 * kNumCU) return launch_conv<1, 2, 8, 1>(x, wpack, y, d, st, who, epi);
    if (mid >= 2 * kNumCU) return launch_conv<1, 2, 4, 1>(x, wpack, y, d, st, who, epi);
    return launch_conv<1, 1, 4, 1>(x, wpack, y, d, st, who, epi);
  }
  // (the 2x8 tile with two output-channel tiles needs 268 registers -> one wave per SIMD; 2x4 keeps two)
  if (mid >= 2 * kNumCU) return launch_conv<2, 2, 4, 1>(x, wpack, y, d, st, who, epi);
  return launch_conv<2, 1, 4, 1>(x, wpack, y, d, st, who, epi);
}

// Stride-2 forward (also the backward-data of ConvTranspose3d k3 s2 p1 op1): output = ((X - 1) / 2 + 1).
int conv3d_s2(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int D, int H, int W,
              hipStream_t st, const char* who, const mode_bn_epilogue* bn = nullptr) {
  CDims d;
  d.B = B; d.Ci = K; d.Co = rows; d.D = D; d.H = H; d.W = W;
  d.Do = (D - 1) / 2 + 1; d.Ho = (H - 1) / 2 + 1; d.Wo = (W - 1) / 2 + 1;
  d.MT = mode::cdiv(rows, 32);
  d.NCHUNK = mode::cdiv(K, CCH);
  MODE_REQUIRE(d.MT <= 2, MODE_ERR_UNSUPPORTED, "%s: more than 64 output channels (%d) not supported", who, rows);
  const long long npack = (long long)d.MT * d.NCHUNK * 27 * 256;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w3d, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, rows, K, d.MT, d.NCHUNK, 0, bn ? 1 : 0,
                     bn ? *bn : mode_bn_epilogue());
  const Epi epi = make_epi(bn, wpack + npack);
  if (d.MT == 1) return launch_conv<1, 1, 4, 2>(x, wpack, y, d, st, who, epi);
  return launch_conv<2, 1, 4, 2>(x, wpack, y, d, st, who, epi);
}

// ---------------------------------------------------------------------------------------------------------------------
// Transposed convolution k3 s2 p1 op1 (ConvTranspose3d of hourglass conv5/conv6, mode_disparity.py:23, 25) = the
// backward-data of the stride-2 convolution:   out[o, z] = sum_{c, k : z = 2q - 1 + k} Wsrc(o, c, k) * x[c, q],  out = 2 x in.
// Split by output parity per axis: an even output coordinate z = 2q' uses only k = 1 (q = q'); an odd one z = 2q'+1 uses
// k = 2 (q = q') and k = 0 (q = q'+1).  The 8 parity classes of one low-resolution voxel carry 1+2+2+2+4+4+4+8 = 27
// taps, so no MFMA is spent on structural zeros.  D[i = o][j = 32 low-res w]; a wave owns one low-res row and keeps the
// 8 class accumulators live; the two w-parities of a row are stored interleaved as one float2 per lane (coalesced).
// Weight tap (kd*9 + kh*3 + kw) of the n-th MFMA group of deconv3d_kernel, in the order its parity-class loops run.
__host__ __device__ constexpr int deconv_wtap(int n) {
  int i = 0;
  for (int pd = 0; pd < 2; ++pd)
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw)
        for (int td = 0; td <= pd; ++td)
          for (int th = 0; th <= ph; ++th)
            for (int tw = 0; tw <= pw; ++tw) {
              if (i == n) return (pd ? (td ? 0 : 2) : 1) * 9 + (ph ? (th ? 0 : 2) : 1) * 3 + (pw ? (tw ? 0 : 2) : 1);
              ++i;
            }
  return 0;
}

template <int TD, int TH, bool EPI>
__global__ __launch_bounds__(NT, EPI ? 2 : 1) void deconv3d_kernel(const float* __restrict__ x, const float4* __restrict__ wp,
                                                      float* __restrict__ y, CDims d, Epi epi) {
  static_assert(TD * TH == 4, "one low-resolution row per wave");
  constexpr int ID = TD + 1, IH = TH + 1, IW = 33;
  constexpr int PLANE = ID * IH * IW;
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [CCH][PLANE]

  int t = xcd_remap(blockIdx.x, d.ntiles);
  const int wt = t % d.nWt;
  t /= d.nWt;
  const int ht = t % d.nHt;
  t /= d.nHt;
  const int dt = t % d.nDt;
  const int b = t / d.nDt;
  const int mt = blockIdx.y;
  const int w0 = wt * 32, h0 = ht * TH, d0 = dt * TD;  // low-resolution (input) coordinates
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const long long HW = (long long)d.H * d.W;
  const long long DHW = (long long)d.D * HW;
  const long long oHW = (long long)d.Ho * d.Wo;
  const long long oDHW = (long long)d.Do * oHW;

  f32x16 acc[2][2][2];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i >> 2][(i >> 1) & 1][i & 1] = (f32x16){0};

  const int dz = wave / TH, hy = wave % TH;
  const float* bbase = tile + (lane >> 5) * PLANE + dz * (IH * IW) + hy * IW + (lane & 31);
  const float* xb = x + (long long)b * d.Ci * DHW;

  // Staging, software pipelined like conv3d_kernel: half-wave hwv owns channel hwv of the 8-channel chunk; its item j is tile
  // row j (RPC = ID*IH rows of 32 coalesced columns), the 33rd column of every row is one more load for the first 72 threads.
  // The loads of chunk ch+1 are issued before the MFMAs of chunk ch and written to LDS after them.
  constexpr int RPC = ID * IH, NROWS = CCH * RPC;
  static_assert(NROWS <= NT, "one thread per row for the leftover column");
  const int hwv = tid >> 5, l32 = tid & 31;
  int rowoff[RPC];
  unsigned rowok = 0;
#pragma unroll
  for (int j = 0; j < RPC; ++j) {
    const int gd = d0 + j / IH, gh = h0 + j % IH;
    const bool ok = gd < d.D && gh < d.H && w0 + l32 < d.W;
    rowoff[j] = ok ? (int)(gd * HW + gh * d.W) + w0 + l32 : 0;
    rowok |= (ok ? 1u : 0u) << j;
  }
  const int e_c = tid / RPC, e_rem = tid - e_c * RPC;  // leftover-column item of thread tid < NROWS
  const int e_gd = d0 + e_rem / IH, e_gh = h0 + e_rem % IH;
  const bool e_in = tid < NROWS && e_gd < d.D && e_gh < d.H && w0 + 32 < d.W;
  const int e_off = e_in ? (int)(e_gd * HW + e_gh * d.W) + w0 + 32 : 0;
  float vm[RPC], ve;
  auto issue = [&](int ch) {
    const bool cok = ch * CCH + hwv < d.Ci;
    // (a half-wave whose channel lies beyond Ci reads the chunk's FIRST channel instead -- the value is dropped in commit().  Round 6:
    // the base used to include the missing channel itself, i.e. an address up to 7 planes past the end of the last sample -- harmless
    // inside a cached allocation, a memory fault when x ended at the end of a mapping: the full GPU suite hit it once its order changed)
    const float* xc = xb + ((long long)ch * CCH + (cok ? hwv : 0)) * DHW;
#pragma unroll
    for (int j = 0; j < RPC; ++j) vm[j] = xc[(unsigned)(cok ? rowoff[j] : 0)];
    const bool eok = e_in && ch * CCH + e_c < d.Ci;
    ve = xb[eok ? ((long long)ch * CCH + e_c) * DHW + e_off : 0];
  };
  auto commit = [&](int ch) {
    const bool cok = ch * CCH + hwv < d.Ci;
    float* dst = tile + hwv * PLANE + l32;
#pragma unroll
    for (int j = 0; j < RPC; ++j) dst[j * IW] = (cok && ((rowok >> j) & 1)) ? vm[j] : 0.f;
    if (tid < NROWS) tile[tid * IW + 32] = (e_in && ch * CCH + e_c < d.Ci) ? ve : 0.f;
  };

  issue(0);
  commit(0);
  __syncthreads();
  for (int ch = 0; ch < d.NCHUNK; ++ch) {
    if (ch + 1 < d.NCHUNK) {
      issue(ch + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // The 27 taps touch only 8 distinct input voxels per channel pair ((dq_d, dq_h, dq_w) in {0,1}^3): the 32 B operands of the
    // chunk are read from LDS once, up front.  The weight fragment of tap n+2 is requested while tap n runs (the compiler on
    // its own requested every fragment right before its first use and waited for the L1/L2 round trip).
    const float4* wq = wp + (((long long)mt * d.NCHUNK + ch) * 27) * 64 + lane;
    float bop[4][8];
#pragma unroll
    for (int cp = 0; cp < 4; ++cp)
#pragma unroll
      for (int v = 0; v < 8; ++v) bop[cp][v] = bbase[2 * cp * PLANE + (v >> 2) * (IH * IW) + ((v >> 1) & 1) * IW + (v & 1)];
    float4 w_a = wq[deconv_wtap(0) * 64], w_b = wq[deconv_wtap(1) * 64];
    int n = 0;
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw)
#pragma unroll
          for (int td = 0; td <= pd; ++td)
#pragma unroll
            for (int th = 0; th <= ph; ++th)
#pragma unroll
              for (int tw = 0; tw <= pw; ++tw) {
                // parity 0: (k = 1, dq = 0); parity 1: t = 0 -> (k = 2, dq = 0), t = 1 -> (k = 0, dq = 1)
                const int v = ((pd ? td : 0) << 2) | ((ph ? th : 0) << 1) | (pw ? tw : 0);
                const float4 a4 = w_a;
                w_a = w_b;
                if (n + 2 < 27) w_b = wq[deconv_wtap(n + 2) * 64];
                ++n;
                __builtin_amdgcn_sched_barrier(0);
                acc[pd][ph][pw] = mfma32(a4.x, bop[0][v], acc[pd][ph][pw]);
                acc[pd][ph][pw] = mfma32(a4.y, bop[1][v], acc[pd][ph][pw]);
                acc[pd][ph][pw] = mfma32(a4.z, bop[2][v], acc[pd][ph][pw]);
                acc[pd][ph][pw] = mfma32(a4.w, bop[3][v], acc[pd][ph][pw]);
              }
    __syncthreads();
    if (ch + 1 < d.NCHUNK) {
      commit(ch + 1);
      __syncthreads();
    }
  }

  float* yb = y + (long long)b * d.Co * oDHW;
  const int qd = d0 + dz, qh = h0 + hy, qw = w0 + (lane & 31);
  // the folded-BatchNorm shifts of this lane's 16 output channels, once: read inside the store loop (`epi.shift` may alias y as far as
  // the compiler knows) every store waited for a load of its own -- 64 serialised L2 round trips per lane and tile (round 5, ISA scan)
  float shv[16];
  if (EPI) {
#pragma unroll
    for (int q = 0; q < 16; ++q) shv[q] = epi.shift[min(mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), d.Co - 1)];
  }
  if (qd < d.D && qh < d.H && qw < d.W) {
#pragma unroll
    for (int pd = 0; pd < 2; ++pd)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const long long sp = (long long)(2 * qd + pd) * oHW + (long long)(2 * qh + ph) * d.Wo + 2 * qw;
        // the residual of this (pd, ph) row: its 16 float2 loads are issued together BEFORE the row's stores.  Read next to the
        // stores (`add` may alias `y` as far as the compiler knows) every load waited for the store in front of it: 0.43 ms for the
        // 64 -> 32 layer of one pair against 0.24 without the residual, whose 201 MB are 0.04 ms of traffic.
        float2 res[16];
        if (EPI) {
#pragma unroll
          for (int q = 0; q < 16; ++q) res[q] = make_float2(0.f, 0.f);
          if (epi.add) {  // (one uniform test around the 16 loads, not one per load)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int o = min(mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5), d.Co - 1);
              res[q] = *reinterpret_cast<const float2*>(epi.add + (long long)b * d.Co * oDHW + o * oDHW + sp);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        auto emit = [&](int q) {
          const int o = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
          float2 v = make_float2(acc[pd][ph][0][q], acc[pd][ph][1][q]);
          if (EPI) {  // torch's order: (convolution * scale + shift) + residual, then ReLU
            const float sh = shv[q];
            v = make_float2((v.x + sh) + res[q].x, (v.y + sh) + res[q].y);
            if (epi.relu) v = make_float2(relu_nan(v.x), relu_nan(v.y));
          }
          *reinterpret_cast<float2*>(yb + o * oDHW + sp) = v;
        };
        if (mt * 32 + 32 <= d.Co) {  // a full tile of output channels (uniform): 16 stores in one block -- with a test per channel every
#pragma unroll                        // store was its own basic block behind a vmcnt(0), i.e. behind the store in front of it
          for (int q = 0; q < 16; ++q) emit(q);
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5) < d.Co) emit(q);
        }
        if (EPI) __builtin_amdgcn_sched_barrier(0);
      }
  }
}

// x (B,K,D,H,W) -> y (B,rows,2D,2H,2W);  flip selects how `w` is indexed (see pack_w3d): 2 for ConvTranspose3d weights
// (Cin,Cout,27), and also 2 for the backward-data of a stride-2 Conv3d whose weight is (Co,Ci,27) with K = Co, rows = Ci.
int deconv3d(const float* x, const float* w, float* y, float* wpack, int B, int K, int rows, int D, int H, int W, hipStream_t st,
             const char* who, const mode_bn_epilogue* bn = nullptr) {
  CDims d;
  d.B = B; d.Ci = K; d.Co = rows; d.D = D; d.H = H; d.W = W;
  d.Do = 2 * D; d.Ho = 2 * H; d.Wo = 2 * W;
  d.MT = mode::cdiv(rows, 32);
  d.NCHUNK = mode::cdiv(K, CCH);
  const long long npack = (long long)d.MT * d.NCHUNK * 27 * 256;
  if (mode::pack_needed()) hipLaunchKernelGGL(pack_w3d, dim3(mode::cdiv(npack, 256)), dim3(256), 0, st, w, wpack, rows, K, d.MT, d.NCHUNK, 2, bn ? 1 : 0,
                     bn ? *bn : mode_bn_epilogue());
  const Epi epi = make_epi(bn, wpack + npack);
  constexpr int TD = 2, TH = 2;
  d.nWt = mode::cdiv(W, 32);
  d.nHt = mode::cdiv(H, TH);
  d.nDt = mode::cdiv(D, TD);
  d.ntiles = B * d.nDt * d.nHt * d.nWt;
  const size_t lds = (size_t)CCH * (TD + 1) * (TH + 1) * 33 * sizeof(float);
  if (epi.shift)
    hipLaunchKernelGGL((deconv3d_kernel<TD, TH, true>), dim3(d.ntiles, d.MT), dim3(NT), lds, st, x,
                       reinterpret_cast<const float4*>(wpack), y, d, epi);
  else
    hipLaunchKernelGGL((deconv3d_kernel<TD, TH, false>), dim3(d.ntiles, d.MT), dim3(NT), lds, st, x,
                       reinterpret_cast<const float4*>(wpack), y, d, epi);
  return mode::check_launch(who);
}

int check_conv_args(const void* a, const void* b, const void* c, const void* wp, int B, int Ci, int D, int H, int W, int Co,
                    int stride, const char* who, bool allow_s2 = false) {
  MODE_REQUIRE(B >= 0 && Ci > 0 && Co > 0 && D > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(stride == 1 || (stride == 2 && allow_s2), MODE_ERR_UNSUPPORTED, "%s: stride %d not implemented", who, stride);
  MODE_REQUIRE((long long)std::max(Ci, 8) * D * H * W < (1ll << 31) && (long long)std::max(Co, 8) * D * H * W < (1ll << 31),
               MODE_ERR_UNSUPPORTED, "%s: a sample larger than 2^31 elements", who);
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(a && b && c && wp, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  return MODE_OK;
}

}  // namespace

extern "C" size_t mode_conv3d_wpack_bytes(int Ci, int Co) {
  if (Ci <= 0 || Co <= 0) return 0;
  const size_t f = (size_t)mode::cdiv(Co, 32) * mode::cdiv(Ci, CCH) * 27 * 256;
  const size_t b = (size_t)mode::cdiv(Ci, 32) * mode::cdiv(Co, CCH) * 27 * 256;
  size_t n = (f > b ? f : b) + 32 * (size_t)mode::cdiv(Co > Ci ? Co : Ci, 32);  // + the folded BatchNorm shifts
  n = std::max(n, std::max(mode::conv3d_split_wpack_floats(Ci, Co), mode::conv3d_split_wpack_floats(Co, Ci)));
  n = std::max(n, mode::conv3d_s2_split_wpack_floats(Ci, Co));
  n = std::max(n, std::max(mode::deconv3d_split_wpack_floats(Ci, Co), mode::deconv3d_split_wpack_floats(Co, Ci)));
  return n * sizeof(float);
}

extern "C" int mode_conv3d_split_supported(int Ci, int Co, int stride, int which) {
  if (Ci <= 0 || Co <= 0) return 0;
  if (stride == 2) {  // forward: mode_conv3d_fwd_s2_split; input gradient: mode_conv3d_bwd_data_s2_split (even volumes); weight gradient:
                      // mode_conv3d_bwd_weight_s2_split (even volumes, W a multiple of 8; gy in blocks of 64 channels, x in blocks of 32)
    if (which == 0) return mode::conv3d_s2_split_supported(Ci, Co) ? 1 : 0;
    if (which == 2) return (Co % 64 == 0 && Ci % 32 == 0) ? 1 : 0;
    return (which == 1 && mode::deconv3d_split_supported(Co, Ci)) ? 1 : 0;
  }
  if (stride != 1) return 0;
  if (which == 2) return Co > 1;  // weight gradient: any channel counts (32 x 32 blocks, masked)
  return which == 1 ? mode::conv3d_split_supported(Co, Ci) : mode::conv3d_split_supported(Ci, Co);
}

extern "C" int mode_conv3d_fwd_split(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci,
                                     int D, int H, int W, int Co, mode_stream_t stream) {
  const char* who = "mode_conv3d_fwd_split";
  int rc = check_conv_args(x, w, y, wpack, B, Ci, D, H, W, Co, 1, who);
  if (rc == MODE_OK && bn) rc = mode::check_bn(bn, who);
  if (rc != MODE_OK || B == 0) return rc;
  return mode::conv3d_s1_split(x, w, y, wpack, B, Ci, Co, D, H, W, 0, mode::as_stream(stream), who, bn);
}

// Training forward with the BatchNorm batch statistics of the output taken in the kernel's epilogue (no statistics pass over y):
// `stats` receives mode_conv3d_fwd_split_stats_partials() partial pairs per output channel followed by the Co pivots
// (>= mode_bn_workspace_bytes(Co) bytes: the BatchNorm workspace itself), to be handed to mode_bn_train_fwd_prestats.
extern "C" int mode_conv3d_fwd_split_stats_partials(void) { return mode::conv3d_split_stat_partials(); }

extern "C" int mode_conv3d_fwd_split_stats(const float* x, const float* w, float* y, float* wpack, float* stats, int B, int Ci, int D, int H,
                                           int W, int Co, mode_stream_t stream) {
  const char* who = "mode_conv3d_fwd_split_stats";
  int rc = check_conv_args(x, w, y, wpack, B, Ci, D, H, W, Co, 1, who);
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(stats, MODE_ERR_WORKSPACE, "%s: statistics workspace required", who);
  MODE_REQUIRE(B > 0, MODE_ERR_BAD_ARG, "%s: empty batch has no statistics", who);
  return mode::conv3d_s1_split(x, w, y, wpack, B, Ci, Co, D, H, W, 0, mode::as_stream(stream), who, nullptr, stats);
}

// Stride-2 forward on the split-bf16 kernel of conv3d_split_s2.hip (also the input gradient of the transposed convolution, with
// w = its (Cin, Cout, 27) weight read as (Co = Cin, Ci = Cout)); needs mode_conv3d_split_supported(Ci, Co, 2, 0) == 1.
extern "C" int mode_conv3d_fwd_s2_split_amax(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* out_absmax, float* wpack,
                                             int B, int Ci, int D, int H, int W, int Co, mode_stream_t stream) {
  const char* who = "mode_conv3d_fwd_s2_split";
  int rc = check_conv_args(x, w, y, wpack, B, Ci, D, H, W, Co, 2, who, true);
  if (rc == MODE_OK && bn) rc = mode::check_bn(bn, who);
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(!out_absmax || bn, MODE_ERR_BAD_ARG, "%s: the output maximum comes out of the eval epilogue (bn is NULL)", who);
  if (B == 0) return out_absmax ? mode::absmax_begin(out_absmax, mode::as_stream(stream), who) : MODE_OK;
  return mode::conv3d_s2_split(x, w, y, wpack, B, Ci, Co, D, H, W, mode::as_stream(stream), who, bn, out_absmax);
}

extern "C" int mode_conv3d_fwd_s2_split(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci,
                                        int D, int H, int W, int Co, mode_stream_t stream) {
  return mode_conv3d_fwd_s2_split_amax(x, w, bn, y, nullptr, wpack, B, Ci, D, H, W, Co, stream);
}

// Input gradient of the stride-2 convolution = the transposed convolution of gy with the same (Co, Ci, 27) weight, on the split-bf16
// kernel of conv3d_split_deconv.hip; even D, H, W; needs mode_conv3d_split_supported(Ci, Co, 2, 1) == 1.
extern "C" int mode_conv3d_bwd_data_s2_split(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int D, int H, int W,
                                             int Co, mode_stream_t stream) {
  const char* who = "mode_conv3d_bwd_data_s2_split";
  int rc = check_conv_args(gy, w, gx, wpack, B, Ci, D, H, W, Co, 2, who, true);
  if (rc != MODE_OK || B == 0) return rc;
  MODE_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, MODE_ERR_UNSUPPORTED, "%s: stride 2 needs even input sizes (got %dx%dx%d)", who, D, H, W);
  return mode::deconv3d_split(gy, w, gx, wpack, B, Co, Ci, D / 2, H / 2, W / 2, mode::as_stream(stream), who);
}

// ---- the stride-1 layer on the two-piece fp16 arithmetic (three MFMAs per product; csrc/conv3d_split.hip).  amax_*: device floats, the
// largest magnitude of each operand (mode_abs_max), read by the kernels themselves: no host synchronisation.
extern "C" int mode_abs_max(const float* x, long long n, float* out, mode_stream_t stream) {
  return mode::abs_max(x, n, out, mode::as_stream(stream), "mode_abs_max");
}

extern "C" int mode_abs_max_batch(const float* const* device_ptrs, const long long* device_sizes, int n, float* out, mode_stream_t stream) {
  return mode::abs_max_batch(device_ptrs, device_sizes, n, out, mode::as_stream(stream), "mode_abs_max_batch");
}

extern "C" int mode_conv3d_fwd_split_f16(const float* x, const float* w, const float* amax_x, const float* amax_w, float* y, float* wpack, int B,
                                         int Ci, int D, int H, int W, int Co, mode_stream_t stream) {
  const char* who = "mode_conv3d_fwd_split_f16";
  int rc = check_conv_args(x, w, y, wpack, B, Ci, D, H, W, Co, 1, who);
  if (rc != MODE_OK || B == 0) return rc;
  MODE_REQUIRE(amax_x && amax_w, MODE_ERR_BAD_ARG, "%s: null operand maximum", who);
  return mode::conv3d_s1_split(x, w, y, wpack, B, Ci, Co, D, H, W, 0, mode::as_stream(stream), who, nullptr, nullptr, nullptr, amax_x, amax_w);
}

// Eval mode on the fp16 arithmetic: y = relu?(bn(conv(x)) [+ add]) with the BatchNorm scale folded into the weights BEFORE they are scaled
// and split (their maximum is taken inside, where they are packed, and kept in wpack); amax_y receives the maximum of y for the next layer.
extern "C" int mode_conv3d_fwd_split_f16_bn(const float* x, const float* w, const float* amax_x, const mode_bn_epilogue* bn, float* y,
                                            float* amax_y, float* wpack, int B, int Ci, int D, int H, int W, int Co, mode_stream_t stream) {
  const char* who = "mode_conv3d_fwd_split_f16_bn";
  int rc = check_conv_args(x, w, y, wpack, B, Ci, D, H, W, Co, 1, who);
  if (rc == MODE_OK) {
    MODE_REQUIRE(bn, MODE_ERR_BAD_ARG, "%s: null BatchNorm epilogue", who);
    rc = mode::check_bn(bn, who);
  }
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(amax_x && amax_y, MODE_ERR_BAD_ARG, "%s: null maximum buffer", who);
  if (B == 0) return mode::absmax_begin(amax_y, mode::as_stream(stream), who);  // (an empty output's maximum is zero)
  return mode::conv3d_s1_split(x, w, y, wpack, B, Ci, Co, D, H, W, 0, mode::as_stream(stream), who, bn, nullptr, nullptr, amax_x, nullptr, amax_y);
}

// gx = conv^T(gy) [+ acc]; amax_g = max |gy|, amax_w = max |w|
extern "C" int mode_conv3d_bwd_data_split_f16(const float* gy, const float* w, const float* amax_g, const float* amax_w, const float* acc, float* gx,
                                              float* wpack, int B, int Ci, int D, int H, int W, int Co, mode_stream_t stream) {
  const char* who = "mode_conv3d_bwd_data_split_f16";
  int rc = check_conv_args(gy, w, gx, wpack, B, Ci, D, H, W, Co, 1, who);
  if (rc != MODE_OK || B == 0) return rc;
  MODE_REQUIRE(amax_g && amax_w && acc != gx, MODE_ERR_BAD_ARG, "%s: null operand maximum, or acc is the output tensor", who);
  return mode::conv3d_s1_split(gy, w, gx, wpack, B, Co, Ci, D, H, W, 1, mode::as_stream(stream), who, nullptr, nullptr, acc, amax_g, amax_w);
}

// The two input gradients with a gradient that is already there added in the store: gx = conv^T(gy) + acc (acc must not alias gx).
extern "C" int mode_conv3d_bwd_data_split_acc_supported(int Ci, int Co, int stride) {
  if (Ci <= 0 || Co <= 0) return 0;
  if (stride == 1) return mode::conv3d_split_supported(Co, Ci) ? 1 : 0;
  return (stride == 2 && mode::deconv3d_split_bn_supported(Co, Ci)) ? 1 : 0;
}

extern "C" int mode_conv3d_bwd_data_split_acc(const float* gy, const float* w, const float* acc, float* gx, float* wpack, int B, int Ci, int D,
                                              int H, int W, int Co, int stride, mode_stream_t stream) {
  const char* who = "mode_conv3d_bwd_data_split_acc";
  int rc = check_conv_args(gy, w, gx, wpack, B, Ci, D, H, W, Co, stride, who, stride == 2);
  if (rc != MODE_OK || B == 0) return rc;
  MODE_REQUIRE(acc && acc != gx, MODE_ERR_BAD_ARG, "%s: acc must be a tensor of gx's shape that is not gx", who);
  if (stride == 1) return mode::conv3d_s1_split(gy, w, gx, wpack, B, Co, Ci, D, H, W, 1, mode::as_stream(stream), who, nullptr, nullptr, acc);
  MODE_REQUIRE(stride == 2 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, MODE_ERR_UNSUPPORTED, "%s: stride 1, or stride 2 with even input sizes (got stride %d, %dx%dx%d)",
               who, stride, D, H, W);
  return mode::deconv3d_split(gy, w, gx, wpack, B, Co, Ci, D / 2, H / 2, W / 2, mode::as_stream(stream), who, nullptr, acc);
}

// ConvTranspose3d k3 s2 p1 op1 on the same kernel: x (B, Cin, D, H, W), w (Cin, Cout, 27) -> y (B, Cout, 2D, 2H, 2W).
extern "C" int mode_deconv3d_split_supported(int Cin, int Cout) { return mode::deconv3d_split_supported(Cin, Cout) ? 1 : 0; }

extern "C" int mode_deconv3d_fwd_split(const float* x, const float* w, float* y, float* wpack, int B, int Cin, int D, int H, int W, int Cout,
                                       mode_stream_t stream) {
  const char* who = "mode_deconv3d_fwd_split";
  int rc = check_conv_args(x, w, y, wpack, B, Cin, D, H, W, Cout, 1, who);
  if (rc != MODE_OK || B == 0) return rc;
  return mode::deconv3d_split(x, w, y, wpack, B, Cin, Cout, D, H, W, mode::as_stream(stream), who);
}

extern "C" int mode_deconv3d_split_bn_supported(int Cin, int Cout) { return mode::deconv3d_split_bn_supported(Cin, Cout) ? 1 : 0; }

extern "C" int mode_deconv3d_fwd_split_bn_amax(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* out_absmax,
                                               float* wpack, int B, int Cin, int D, int H, int W, int Cout, mode_stream_t stream) {
  const char* who = "mode_deconv3d_fwd_split_bn";
  int rc = check_conv_args(x, w, y, wpack, B, Cin, D, H, W, Cout, 1, who);
  if (rc == MODE_OK) rc = mode::check_bn(bn, who);
  if (rc != MODE_OK) return rc;
  if (B == 0) return out_absmax ? mode::absmax_begin(out_absmax, mode::as_stream(stream), who) : MODE_OK;
  return mode::deconv3d_split(x, w, y, wpack, B, Cin, Cout, D, H, W, mode::as_stream(stream), who, bn, nullptr, out_absmax);
}

extern "C" int mode_deconv3d_fwd_split_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Cin,
                                          int D, int H, int W, int Cout, mode_stream_t stream) {
  return mode_deconv3d_fwd_split_bn_amax(x, w, bn, y, nullptr, wpack, B, Cin, D, H, W, Cout, stream);
}

extern "C" int mode_conv3d_bwd_data_split(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int D, int H, int W,
                                          int Co, mode_stream_t stream) {
  const char* who = "mode_conv3d_bwd_data_split";
  int rc = check_conv_args(gy, w, gx, wpack, B, Ci, D, H, W, Co, 1, who);
  if (rc != MODE_OK || B == 0) return rc;
  return mode::conv3d_s1_split(gy, w, gx, wpack, B, Co, Ci, D, H, W, 1, mode::as_stream(stream), who, nullptr);
}

extern "C" int mode_conv3d_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Ci, int D, int H, int W, int Co,
                               int stride, mode_stream_t stream) {
  int rc = check_conv_args(x, w, y, wpack, B, Ci, D, H, W, Co, stride, "mode_conv3d_fwd", true);
  if (rc != MODE_OK || B == 0) return rc;
  if (stride == 2) return conv3d_s2(x, w, y, wpack, B, Ci, Co, D, H, W, mode::as_stream(stream), "mode_conv3d_fwd");
  if (Co == 1) return mode::conv3d_co1_fwd(x, w, y, B, Ci, D, H, W, mode::as_stream(stream), "mode_conv3d_fwd");
  return conv3d_s1(x, w, y, wpack, B, Ci, Co, D, H, W, 0, mode::as_stream(stream), "mode_conv3d_fwd");
}

extern "C" int mode_conv3d_fwd_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Ci,
                                  int D, int H, int W, int Co, int stride, mode_stream_t stream) {
  const char* who = "mode_conv3d_fwd_bn";
  int rc = check_conv_args(x, w, y, wpack, B, Ci, D, H, W, Co, stride, who, true);
  if (rc == MODE_OK) rc = mode::check_bn(bn, who);
  if (rc != MODE_OK || B == 0) return rc;
  MODE_REQUIRE(Co > 1, MODE_ERR_UNSUPPORTED, "%s: the single-output-channel kernels have no epilogue", who);
  if (stride == 2) return conv3d_s2(x, w, y, wpack, B, Ci, Co, D, H, W, mode::as_stream(stream), who, bn);
  return conv3d_s1(x, w, y, wpack, B, Ci, Co, D, H, W, 0, mode::as_stream(stream), who, bn);
}

extern "C" int mode_conv3d_bwd_data(const float* gy, const float* w, float* gx, float* wpack, int B, int Ci, int D, int H, int W,
                                    int Co, int stride, mode_stream_t stream) {
  int rc = check_conv_args(gy, w, gx, wpack, B, Ci, D, H, W, Co, stride, "mode_conv3d_bwd_data", true);
  if (rc != MODE_OK || B == 0) return rc;
  if (stride == 2) {
    // gx = transposed convolution of gy with the same weights, indexed w[o][c][tap] = w[cin_of_deconv][cout_of_deconv][tap]
    MODE_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, MODE_ERR_UNSUPPORTED,
                 "mode_conv3d_bwd_data: stride 2 needs even input sizes (got %dx%dx%d)", D, H, W);
    return deconv3d(gy, w, gx, wpack, B, Co, Ci, D / 2, H / 2, W / 2, mode::as_stream(stream), "mode_conv3d_bwd_data");
  }
  if (Co == 1) return mode::conv3d_co1_bwd_data(gy, w, gx, B, Ci, D, H, W, mode::as_stream(stream), "mode_conv3d_bwd_data");
  return conv3d_s1(gy, w, gx, wpack, B, Co, Ci, D, H, W, 1, mode::as_stream(stream), "mode_conv3d_bwd_data");
}

// =====================================================================================================================
// Backward w.r.t. the weight:  gW[o][c][tap] = sum_{b,q} gy[b,o,q] * x[b,c, S*q + k - 1]   (q = output voxel, S = stride).
//   D[i = o][j = c] per tap;  A[i = o][k = voxel] = gy tile (LDS),  B[k = voxel][j = c] = x tile shifted by the tap (LDS).
// A workgroup owns a 32x32 (o,c) block and a slice of the spatial tiles (WTH rows x 32 output voxels); its 4 waves share
// the A fragment and split the 27 taps (7,7,7,6 accumulators).  Split-K partials are reduced in a fixed order.
// The same kernel serves ConvTranspose3d (k3 s2 p1 op1): gWt[ci][co][k] = sum_q x_in[ci,q] * gy_out[co, 2q + k - 1], i.e.
// call it with (gy := transposed-conv input, x := transposed-conv output gradient, S = 2).
namespace {

struct WDims {
  int B, Ci, Co, D, H, W;  // x volume (high resolution)
  int Do, Ho, Wo;          // gy volume
  int nWt, nHt;
  int T;  // spatial tiles in total
  int S;  // split-K slices
  int MTo, MTc;
};

template <int S, int WTH>
struct WGeom {
  static constexpr int XR = (WTH - 1) * S + 3;          // x rows per tile
  static constexpr int XW = (S == 1) ? 34 : 65;         // x columns per tile
  static constexpr int XP0 = 3 * XR * XW;
  static constexpr int XPLANE = XP0 | 1;                // odd -> lanes (= channels) hit distinct banks
  static constexpr int GPLANE = WTH * 32 + 1;
  static constexpr size_t LDS = (size_t)(32 * XPLANE + 32 * GPLANE) * sizeof(float);
};

template <int S, int WTH>
__global__ __launch_bounds__(NT) void conv3d_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                               float* __restrict__ part, WDims d) {
  using G = WGeom<S, WTH>;
  constexpr int XR = G::XR, XW = G::XW, XPLANE = G::XPLANE, GPLANE = G::GPLANE;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xl = lds;                 // [32][XPLANE]
  float* gl = lds + 32 * XPLANE;   // [32][GPLANE]
  const int s = blockIdx.x, ob = blockIdx.y, cb = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long long HW = (long long)d.H * d.W;
  const long long DHW = (long long)d.D * HW;
  const long long oHW = (long long)d.Ho * d.Wo;
  const long long oDHW = (long long)d.Do * oHW;

  f32x16 acc[7];
  int toff[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    acc[t] = (f32x16){0};
    const int tap = wave + 4 * t;  // < 27 except wave 3, t = 6
    toff[t] = (tap / 9) * (XR * XW) + ((tap / 3) % 3) * XW + (tap % 3);
  }
  const bool last_valid = (wave + 24) < 27;

  for (int tt = s; tt < d.T; tt += d.S) {
    int t = tt;
    const int wt = t % d.nWt;
    t /= d.nWt;
    const int ht = t % d.nHt;
    t /= d.nHt;
    const int qd = t % d.Do;
    const int b = t / d.Do;
    const int w0 = wt * 32, h0 = ht * WTH;  // output (gy) coordinates
    const float* xb = x + ((long long)b * d.Ci + cb * 32) * DHW;
    const float* gb = gy + ((long long)b * d.Co + ob * 32) * oDHW;
    {
      // Branch-free row staging (see conv3d_kernel): the x tile is 32*3*XR rows of XW floats; a half-wave loads 32
      // consecutive columns of a row per instruction, 8 loads per thread in flight; leftover columns one per thread.
      constexpr int NROWS = 32 * 3 * XR;
      constexpr int NG = (XW - 1) / 32;          // full 32-column groups per row: 1 (34 cols) or 2 (65 cols)
      constexpr int NLEFT = XW - 32 * NG;        // 2 or 1 leftover columns
      constexpr int WNF = 8;                     // loads in flight per thread and batch (24 drops the kernel to one wave per SIMD: 1.9 vs 1.2 ms)
      const int hwv = tid >> 5, l32 = tid & 31;
#pragma unroll 1
      for (int kb = 0; kb < NROWS * NG; kb += 8 * WNF) {
        float t8[WNF];
#pragma unroll
        for (int j = 0; j < WNF; ++j) {
          const int item = kb + j * 8 + hwv;
          const int r = item / NG, g = item - r * NG;
          const int c = r / (3 * XR), rem = r - c * (3 * XR);
          const int gd = qd * S + rem / XR - 1, gh = h0 * S + rem % XR - 1;
          const int wx = (NLEFT == 2 ? 1 : 0) + 32 * g + l32;
          const int gw = w0 * S + wx - 1;
          const bool ok = item < NROWS * NG && cb * 32 + c < d.Ci && gd >= 0 && gd < d.D && gh >= 0 && gh < d.H && gw >= 0 && gw < d.W;
          const float v = xb[ok ? c * DHW + gd * HW + gh * d.W + gw : 0];
          t8[j] = ok ? v : 0.f;
        }
#pragma unroll
        for (int j = 0; j < WNF; ++j) {
          const int item = kb + j * 8 + hwv;
          const int r = item / NG, g = item - r * NG;
          const int c = r / (3 * XR), rem = r - c * (3 * XR);
          const int wx = (NLEFT == 2 ? 1 : 0) + 32 * g + l32;
          if (item < NROWS * NG) xl[c * XPLANE + rem * XW + wx] = t8[j];
        }
      }
#pragma unroll 1
      for (int item = tid; item < NROWS * NLEFT; item += NT) {
        const int r = item / NLEFT, side = item - r * NLEFT;
        const int c = r / (3 * XR), rem = r - c * (3 * XR);
        const int gd = qd * S + rem / XR - 1, gh = h0 * S + rem % XR - 1;
        const int wx = (NLEFT == 2) ? (side ? XW - 1 : 0) : XW - 1;
        const int gw = w0 * S + wx - 1;
        const bool ok = cb * 32 + c < d.Ci && gd >= 0 && gd < d.D && gh >= 0 && gh < d.H && gw >= 0 && gw < d.W;
        const float v = xb[ok ? c * DHW + gd * HW + gh * d.W + gw : 0];
        xl[c * XPLANE + rem * XW + wx] = ok ? v : 0.f;
      }
      constexpr int GROWS = 32 * WTH;  // (output channel, row) pairs of the gy tile
#pragma unroll 1
      for (int kb = 0; kb < GROWS; kb += 64) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = kb + j * 8 + hwv;
          const int o = r / WTH, gh = h0 + r % WTH, gw = w0 + l32;
          const bool ok = r < GROWS && ob * 32 + o < d.Co && gh < d.Ho && gw < d.Wo;
          const float v = gb[ok ? o * oDHW + qd * oHW + gh * d.Wo + gw : 0];
          t8[j] = ok ? v : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = kb + j * 8 + hwv;
          if (r < GROWS) gl[(r / WTH) * GPLANE + (r % WTH) * 32 + l32] = t8[j];
        }
      }
    }
    __syncthreads();
    const float* ap = gl + (lane & 31) * GPLANE + (lane >> 5);
    const float* bp = xl + (lane & 31) * XPLANE + (lane >> 5) * S;
#pragma unroll
    for (int row = 0; row < WTH; ++row) {
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const float a = ap[row * 32 + 2 * ks];
        const float* bq = bp + row * S * XW + 2 * ks * S;
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

// Stride-2 variant, software pipelined.  The generic kernel above alternates "stage a tile" and "27 x 16 MFMAs on it" with a
// barrier in between, and at two workgroups per CU (75 KB of LDS each) there is nothing else to hide the staging behind:
// 37 TFLOP/s.  Here the NEXT tile travels global -> registers while the MFMAs of the current one run (78 loads per thread in
// flight, issued before the k-loop, consumed after it), so a step costs max(load, MFMA) + the LDS stores.  The item -> (channel,
// row, column group) map is chosen so that everything but the half-wave's channel is a compile-time constant: a tile needs 9
// row offsets, 2 column offsets and an 11-bit validity mask, the 72 addresses are sums of those.
// OB = output-channel blocks per workgroup (256 threads each): with OB = 2 the two blocks of a 64-channel gy share one staged x
// tile -- half the loads, LDS stores and prefetch registers per thread for the same MFMAs.
template <int OB>
__global__ __launch_bounds__(NT * OB, 2) void conv3d_bwd_weight_s2_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                     float* __restrict__ part, WDims d) {
  using G = WGeom<2, 1>;
  constexpr int XR = G::XR, XW = G::XW, XPLANE = G::XPLANE, GPLANE = G::GPLANE;
  static_assert(XR == 3 && XW == 65, "item map below assumes the 3 x 3 x 65 tile");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xl = lds;                 // [32][XPLANE]
  float* gl = lds + 32 * XPLANE;   // [32 * OB][GPLANE]
  constexpr int NTH = NT * OB, HWV = 8 * OB;                 // threads, half-waves
  constexpr int NQ = 4 / OB, NX = 18 * NQ;                   // channels per half-wave, x items per thread
  constexpr int NE = (288 + NTH - 1) / NTH;                  // column-64 items per thread
  const int s = blockIdx.x, cb = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) & 3, oh = tid >> 8;            // tap group, output-channel block of this wave
  const int ob0 = blockIdx.y * OB, ob = ob0 + oh;
  const int hwv = tid >> 5, l32 = tid & 31;
  const int HW = d.H * d.W, DHW = d.D * HW;          // (host guarantees Ci * DHW < 2^29)
  const int oHW = d.Ho * d.Wo, oDHW = d.Do * oHW;

  f32x16 acc[7];
  int toff[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    acc[t] = (f32x16){0};
    const int tap = min(wave + 4 * t, 26);  // wave 3, t = 6 would be tap 27: it repeats tap 26 and drops the result
    toff[t] = (tap / 9) * (XR * XW) + ((tap / 3) % 3) * XW + (tap % 3);
  }

  // x tile = 288 (channel, depth, row) rows of 65 columns.  Columns 0..63: item jj of half-wave hwv is channel hwv + 8 * (jj / 18),
  // row (jj % 18) / 2, column group jj & 1.  Column 64: one item per thread (+ 32 threads a second one).  gy tile: 32 x 32.
  float px[NX], pe[NE], pg[4];
  unsigned chan_ok = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) chan_ok |= (cb * 32 + hwv + HWV * q < d.Ci ? 1u : 0u) << q;
  const bool all_chan = cb * 32 + 32 <= d.Ci;  // (block-uniform)
  const int e_r0 = tid, e_r1 = tid + NTH;  // rows of the column-64 items (the second one only with OB = 1)
  const int e_c0 = min(e_r0, 287) / 9, e_m0 = min(e_r0, 287) - 9 * e_c0, e_c1 = min(e_r1, 287) / 9, e_m1 = min(e_r1, 287) - 9 * e_c1;

  // issues the loads of tile tt, returns its validity mask (bits 0..8 rows, 9..10 column groups, 11 column 64, 12 gy column)
  auto prefetch = [&](int tt) -> unsigned {
    int t = tt;
    const int wt = t % d.nWt;
    t /= d.nWt;
    const int ht = t % d.nHt;
    t /= d.nHt;
    const int qd = t % d.Do;
    const int b = t / d.Do;
    const int w0 = wt * 32, h0 = ht;
    const float* xb = x + ((long long)b * d.Ci + cb * 32) * (long long)DHW;
    const float* gb = gy + ((long long)b * d.Co + ob0 * 32) * (long long)oDHW;
    unsigned m = 0;
    int rowoff[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const int gd = 2 * qd + r / 3 - 1, gh = 2 * h0 + r % 3 - 1;
      const bool ok = gd >= 0 && gd < d.D && gh >= 0 && gh < d.H;
      rowoff[r] = ok ? gd * HW + gh * d.W : 0;
      m |= (ok ? 1u : 0u) << r;
    }
    const int gw0 = 2 * w0 + l32 - 1, gw1 = gw0 + 32, gw2 = 2 * w0 + 63;
    m |= (gw0 >= 0 && gw0 < d.W ? 1u : 0u) << 9;
    m |= (gw1 < d.W ? 1u : 0u) << 10;
    m |= (gw2 < d.W ? 1u : 0u) << 11;
    m |= (w0 + l32 < d.Wo ? 1u : 0u) << 12;
    const int cbase = hwv * DHW;
    if (all_chan && (m & 0x1ffu) == 0x1ffu) {
      // interior tile (every row and channel exists -- all but the first depth / row of a sample): no per-item masks, the two
      // column groups read a clamped column and are masked by one select each at the LDS store (PMC: the general form below
      // costs 4.9 VALU + 3.4 SALU instructions per MFMA)
      m |= 1u << 13;
      const int c0 = cbase + ((m >> 9) & 1 ? gw0 : 0), c1 = cbase + ((m >> 10) & 1 ? gw1 : 0);
#pragma unroll
      for (int jj = 0; jj < NX; ++jj) {
        const int q = jj / 18, r = (jj % 18) / 2, g = jj & 1;
        px[jj] = xb[(g ? c1 : c0) + q * HWV * DHW + rowoff[r]];
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < NX; ++jj) {
        const int q = jj / 18, r = (jj % 18) / 2, g = jj & 1;
        const bool ok = ((chan_ok >> q) & 1) && ((m >> r) & 1) && ((m >> (9 + g)) & 1);
        const int off = cbase + q * HWV * DHW + rowoff[r] + (g ? gw1 : gw0);
        px[jj] = xb[ok ? off : 0];
      }
    }
    {
      const int gd0 = 2 * qd + e_m0 / 3 - 1, gh0 = 2 * h0 + e_m0 % 3 - 1;
      const bool ok0 = e_r0 < 288 && ((m >> e_m0) & 1) && ((m >> 11) & 1) && cb * 32 + e_c0 < d.Ci;
      pe[0] = xb[ok0 ? e_c0 * DHW + gd0 * HW + gh0 * d.W + gw2 : 0];
      if (NE > 1) {
        const int gd1 = 2 * qd + e_m1 / 3 - 1, gh1 = 2 * h0 + e_m1 % 3 - 1;
        const bool ok1 = e_r1 < 288 && ((m >> e_m1) & 1) && ((m >> 11) & 1) && cb * 32 + e_c1 < d.Ci;
        pe[NE - 1] = xb[ok1 ? e_c1 * DHW + gd1 * HW + gh1 * d.W + gw2 : 0];
      }
    }
    const int gbase = qd * oHW + h0 * d.Wo + w0 + l32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = HWV * j + hwv;  // row of the 32 * OB gy tile
      const bool ok = ((m >> 12) & 1) && ob0 * 32 + o < d.Co;
      pg[j] = gb[ok ? o * oDHW + gbase : 0];
    }
    return m;
  };
  // registers -> LDS for the tile whose mask is m (masked elements become zeros)
  auto store = [&](unsigned m) {
    float* xrow = xl + hwv * XPLANE + l32;
    if ((m >> 13) & 1) {  // interior tile: only the column masks
      const bool ok0 = (m >> 9) & 1, ok1 = (m >> 10) & 1;
#pragma unroll
      for (int jj = 0; jj < NX; ++jj) {
        const int q = jj / 18, r = (jj % 18) / 2, g = jj & 1;
        xrow[q * HWV * XPLANE + r * XW + 32 * g] = (g ? ok1 : ok0) ? px[jj] : 0.f;
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < NX; ++jj) {
        const int q = jj / 18, r = (jj % 18) / 2, g = jj & 1;
        const bool ok = ((chan_ok >> q) & 1) && ((m >> r) & 1) && ((m >> (9 + g)) & 1);
        xrow[q * HWV * XPLANE + r * XW + 32 * g] = ok ? px[jj] : 0.f;
      }
    }
    if (e_r0 < 288) {
      const bool ok0 = ((m >> e_m0) & 1) && ((m >> 11) & 1) && cb * 32 + e_c0 < d.Ci;
      xl[e_c0 * XPLANE + e_m0 * XW + 64] = ok0 ? pe[0] : 0.f;
    }
    if (NE > 1 && e_r1 < 288) {
      const bool ok1 = ((m >> e_m1) & 1) && ((m >> 11) & 1) && cb * 32 + e_c1 < d.Ci;
      xl[e_c1 * XPLANE + e_m1 * XW + 64] = ok1 ? pe[NE - 1] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = HWV * j + hwv;
      const bool ok = ((m >> 12) & 1) && ob0 * 32 + o < d.Co;
      gl[o * GPLANE + l32] = ok ? pg[j] : 0.f;
    }
  };

  const int s_x = xcd_remap(s, d.S);  // XCD-aware tile order (neighbouring tiles share halos in one L2), see the ring kernel
  unsigned mask = 0;
  if (s_x < d.T) mask = prefetch(s_x);
  const float* ap = gl + (oh * 32 + (lane & 31)) * GPLANE + (lane >> 5);
  const float* bp = xl + (lane & 31) * XPLANE + (lane >> 5) * 2;
  for (int tt = s_x; tt < d.T; tt += d.S) {
    store(mask);
    __syncthreads();
    if (tt + d.S < d.T) mask = prefetch(tt + d.S);
    __builtin_amdgcn_sched_barrier(0);
    // operands of k-step ks+1 are read from LDS before the 7 MFMAs of k-step ks are issued (the straightforward loop waited
    // for an LDS round trip at the head of every k-step, and once more inside the branch around the 7th tap: all four waves
    // now run 7 taps -- wave 3's last one is a duplicate whose result is dropped -- and there is no branch)
    float a_n = ap[0], b_n[7];
#pragma unroll
    for (int t7 = 0; t7 < 7; ++t7) b_n[t7] = bp[toff[t7]];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const float a = a_n;
      float bb[7];
#pragma unroll
      for (int t7 = 0; t7 < 7; ++t7) bb[t7] = b_n[t7];
      if (ks + 1 < 16) {
        a_n = ap[2 * (ks + 1)];
#pragma unroll
        for (int t7 = 0; t7 < 7; ++t7) b_n[t7] = bp[4 * (ks + 1) + toff[t7]];
      }
      __builtin_amdgcn_sched_barrier(0);  // (the compiler would sink these reads to their first use)
#pragma unroll
      for (int t7 = 0; t7 < 7; ++t7) acc[t7] = mfma32(a, bb[t7], acc[t7]);
    }
    __syncthreads();
  }

  float* pb = part + (((long long)s * d.MTo + (ob < d.MTo ? ob : 0)) * d.MTc + cb) * (27 * 1024);
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = wave + 4 * t;
    if (tap < 27 && ob < d.MTo) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        pb[tap * 1024 + i * 32 + (lane & 31)] = acc[t][q];
      }
    }
  }
}

// Stride-1 variant with a rolling depth window: a work unit is a (b, h-tile, w-tile) column times a run of DC consecutive
// depths; the three x planes d-1, d, d+1 live in an LDS ring and only plane d+1 is staged per step (the halo re-read per
// 64 output voxels drops from 408 to 136 floats per channel).  Same fragment maps and partial layout as the kernel above.

__global__ __launch_bounds__(NT) void conv3d_bwd_weight_ring_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                    float* __restrict__ part, WDims d, int nDc, int units, int ring_dc) {
  constexpr int WTH = 2, XR = WTH + 2, XW = 34, PS = XR * XW;  // plane = 136 floats
    constexpr int XPLANE = 3 * PS + 1;                           // 409 (odd)
  constexpr int GPLANE = WTH * 32 + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xl = lds;                 // [32][3 ring slots][XR][XW]
  float* gl = lds + 32 * XPLANE;   // [32][GPLANE]
  const int s = blockIdx.x, ob = blockIdx.y, cb = blockIdx.z;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long long HW = (long long)d.H * d.W;
  const long long DHW = (long long)d.D * HW;

  f32x16 acc[7];
  int kd[7], khw[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    acc[t] = (f32x16){0};
    const int tap = min(wave + 4 * t, 26);  // wave 3, t = 6 would be tap 27: it repeats tap 26 and drops the result (no branch)
    kd[t] = tap / 9;
    khw[t] = ((tap / 3) % 3) * XW + (tap % 3);
  }
  const int hwv = tid >> 5, l32 = tid & 31;
  const int HWi = d.H * d.W, DHWi = d.D * HWi;  // (host guarantees 32-bit element offsets within a sample)
  unsigned chan_ok = 0, gchan_ok = 0;           // x item j is channel 2 * j + (hwv >> 2), gy item j is channel 4 * j + (hwv >> 1)
#pragma unroll
  for (int j = 0; j < 16; ++j) chan_ok |= (cb * 32 + 2 * j + (hwv >> 2) < d.Ci ? 1u : 0u) << j;
#pragma unroll
  for (int j = 0; j < 8; ++j) gchan_ok |= (ob * 32 + 4 * j + (hwv >> 1) < d.Co ? 1u : 0u) << j;

  // XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, so give every XCD a contiguous range of units --
  // neighbouring tiles share halo rows, and they only meet in an L2 if they run on the same XCD (PMC: 1.20 GB fetched per launch
  // with the plain mapping, for 0.81 GB of operands)
  for (int u = xcd_remap(s, d.S); u < units; u += d.S) {
    int t = u;
    const int dc = t % nDc;
    t /= nDc;
    const int wt = t % d.nWt;
    t /= d.nWt;
    const int ht = t % d.nHt;
    const int b = t / d.nHt;
    const int w0 = wt * 32, h0 = ht * WTH;
    const int dlo = dc * ring_dc, dhi = min(d.D, dlo + ring_dc);
    const float* xb = x + ((long long)b * d.Ci + cb * 32) * DHW;
    const float* gb = gy + ((long long)b * d.Co + ob * 32) * DHW;

    // Software pipeline over the depths of the unit: while the MFMAs of depth dd run, plane dd+2 and the gy rows of depth
    // dd+1 travel global -> registers (25 loads per thread); they are written to LDS (ring slot of plane dd-1, which is dead
    // by then) after the k-loop.  Loads are unconditional from clamped addresses, the masks are applied at the LDS store.
    const int gh_x = h0 + (hwv & 3) - 1;      // every item of a thread is the same tile row ...
    const int gw_x = w0 + l32;                // ... and column
    const bool rc_ok = gh_x >= 0 && gh_x < d.H && gw_x < d.W;
    const int xoff = (hwv >> 2) * DHWi + (rc_ok ? gh_x * d.W + gw_x : 0);  // + 2 * item * DHW + z * HW
    const int hr = tid >> 1, hside = tid & 1;  // halo item: (channel, row) = (hr >> 2, hr & 3), left / right column
    const int gh_h = h0 + (hr & 3) - 1, gw_h = hside ? w0 + 32 : w0 - 1;
    const bool h_ok = cb * 32 + (hr >> 2) < d.Ci && gh_h >= 0 && gh_h < d.H && gw_h >= 0 && gw_h < d.W;
    const int hoff = h_ok ? (hr >> 2) * DHWi + gh_h * d.W + gw_h : 0;
    const int gh_g = h0 + (hwv & 1);           // gy item j: output channel 4 * j + (hwv >> 1), row hwv & 1
    const bool g_ok = gh_g < d.H && gw_x < d.W;
    const int goff = (hwv >> 1) * DHWi + (g_ok ? gh_g * d.W + gw_x : 0);
    float px[16], ph, pg[8];
    auto load_plane = [&](int z) {
      const int zo = (z >= 0 && z < d.D ? z : 0) * HWi;
#pragma unroll
      for (int j = 0; j < 16; ++j) px[j] = xb[((chan_ok >> j) & 1) && rc_ok ? xoff + 2 * j * DHWi + zo : 0];
      ph = xb[hoff + (h_ok ? zo : 0)];
    };
    auto store_plane = [&](int z) {
      const bool zok = z >= 0 && z < d.D;
      float* dst = xl + (hwv >> 2) * XPLANE + ((z + 3) % 3) * PS + (hwv & 3) * XW + 1 + l32;
#pragma unroll
      for (int j = 0; j < 16; ++j) dst[2 * j * XPLANE] = (zok && rc_ok && ((chan_ok >> j) & 1)) ? px[j] : 0.f;
      xl[(hr >> 2) * XPLANE + ((z + 3) % 3) * PS + (hr & 3) * XW + (hside ? 33 : 0)] = (zok && h_ok) ? ph : 0.f;
    };
    auto load_gy = [&](int z) {
#pragma unroll
      for (int j = 0; j < 8; ++j) pg[j] = gb[((gchan_ok >> j) & 1) && g_ok ? goff + 4 * j * DHWi + z * HWi : 0];
    };
    auto store_gy = [&]() {
      float* dst = gl + (hwv >> 1) * GPLANE + (hwv & 1) * 32 + l32;
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[4 * j * GPLANE] = (g_ok && ((gchan_ok >> j) & 1)) ? pg[j] : 0.f;
    };
    // prologue: planes dlo-1, dlo, dlo+1 and the gy rows of depth dlo
    load_plane(dlo - 1);
    load_gy(dlo);
    store_plane(dlo - 1);
    store_gy();
    load_plane(dlo);
    store_plane(dlo);
    load_plane(dlo + 1);
    store_plane(dlo + 1);
    __syncthreads();

    for (int dd = dlo; dd < dhi; ++dd) {
      const bool more = dd + 1 < dhi;
      if (more) {
        load_plane(dd + 2);
        load_gy(dd + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      int toff[7];
#pragma unroll
      for (int t7 = 0; t7 < 7; ++t7) toff[t7] = ((dd + kd[t7] + 2) % 3) * PS + khw[t7];  // depth dd + kd - 1
      const float* ap = gl + (lane & 31) * GPLANE + (lane >> 5);
      const float* bp = xl + (lane & 31) * XPLANE + (lane >> 5);
      // 32 k-steps (2 rows x 16 voxel pairs); the operands of step i+1 are read from LDS before the 7 MFMAs of step i are issued
      float a_n = ap[0], b_n[7];
#pragma unroll
      for (int t7 = 0; t7 < 7; ++t7) b_n[t7] = bp[toff[t7]];
#pragma unroll
      for (int i = 0; i < WTH * 16; ++i) {
        const float a = a_n;
        float bb[7];
#pragma unroll
        for (int t7 = 0; t7 < 7; ++t7) bb[t7] = b_n[t7];
        if (i + 1 < WTH * 16) {
          const int row = (i + 1) / 16, ks = (i + 1) % 16;
          a_n = ap[row * 32 + 2 * ks];
#pragma unroll
          for (int t7 = 0; t7 < 7; ++t7) b_n[t7] = bp[row * XW + 2 * ks + toff[t7]];
        }
        __builtin_amdgcn_sched_barrier(0);  // (the compiler would sink these reads to their first use)
#pragma unroll
        for (int t7 = 0; t7 < 7; ++t7) acc[t7] = mfma32(a, bb[t7], acc[t7]);
      }
      __syncthreads();
      if (more) {
        store_plane(dd + 2);
        store_gy();
      }
      __syncthreads();
    }
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

// gw[o][c][tap] (+)= sum_s part[s][o/32][c/32][tap][o%32][c%32].  A block of 256 threads handles 32 consecutive elements of the
// partial layout: thread (e, g) sums the slices s = g, g+8, ... in order (coalesced 128-byte reads per slice), the 8 group sums
// are combined through LDS in a fixed association -- deterministic, and 8x the parallelism of one thread per element.
__global__ __launch_bounds__(256) void reduce_gw3d(const float* __restrict__ part, float* __restrict__ gw, WDims d, int accumulate) {
  __shared__ float sh[8][33];
  const long long stride = (long long)d.MTo * d.MTc * 27 * 1024;
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
    const int tap = (int)(r % 27);
    r /= 27;
    const int cb = (int)(r % d.MTc), ob = (int)(r / d.MTc);
    const int o = ob * 32 + i, c = cb * 32 + j;
    if (o < d.Co && c < d.Ci) {
      float* q = gw + ((long long)o * d.Ci + c) * 27 + tap;
      *q = accumulate ? *q + sum : sum;
    }
  }
}

constexpr int WTH1 = 2, WTH2 = 1;  // output rows per spatial tile for stride 1 / 2

void make_wdims(WDims& d, int B, int Ci, int D, int H, int W, int Co, int stride) {
  d.B = B; d.Ci = Ci; d.Co = Co; d.D = D; d.H = H; d.W = W;
  d.Do = (D - 1) / stride + 1; d.Ho = (H - 1) / stride + 1; d.Wo = (W - 1) / stride + 1;
  d.nWt = mode::cdiv(d.Wo, 32);
  d.nHt = mode::cdiv(d.Ho, stride == 1 ? WTH1 : WTH2);
  d.T = B * d.Do * d.nHt * d.nWt;
  d.MTo = mode::cdiv(Co, 32);
  d.MTc = mode::cdiv(Ci, 32);
  int S = mode::cdiv(2 * kNumCU, d.MTo * d.MTc);
  if (S > d.T) S = d.T;
  if (S < 1) S = 1;
  d.S = S;
}

}  // namespace

extern "C" size_t mode_conv3d_bwd_weight_workspace_bytes(int B, int Ci, int D, int H, int W, int Co, int stride) {
  if (B <= 0 || Ci <= 0 || Co <= 0 || D <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2)) return 0;
  WDims d;
  make_wdims(d, B, Ci, D, H, W, Co, stride);
  const size_t generic = (size_t)d.S * d.MTo * d.MTc * 27 * 1024;
  const size_t co1 = (Co == 1 && stride == 1) ? mode::conv3d_co1_bwd_weight_workspace_floats(B, Ci, D, H, W) : 0;
  return std::max(generic, co1) * sizeof(float);
}

extern "C" int mode_conv3d_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int D, int H,
                                      int W, int Co, int stride, int accumulate, mode_stream_t stream) {
  int rc = check_conv_args(gy, x, gw, workspace, B, Ci, D, H, W, Co, stride, "mode_conv3d_bwd_weight", true);
  if (rc != MODE_OK) return rc;
  hipStream_t st = mode::as_stream(stream);
  if (B == 0) {
    if (!accumulate) return mode::zero_floats(gw, (size_t)Co * Ci * 27, st, "mode_conv3d_bwd_weight");
    return MODE_OK;
  }
  if (Co == 1 && stride == 1)
    return mode::conv3d_co1_bwd_weight(gy, x, gw, workspace, B, Ci, D, H, W, accumulate, st, "mode_conv3d_bwd_weight");
  WDims d;
  make_wdims(d, B, Ci, D, H, W, Co, stride);
  if (stride == 1) {
    const size_t lds = WGeom<1, WTH1>::LDS;  // same footprint: 3 planes of (WTH+2) x 34 per channel + the gy tile
    if ((long long)std::max(Ci, Co) * D * H * W < (1ll << 29)) {  // 32-bit element offsets within a sample
      rc = mode::allow_lds(conv3d_bwd_weight_ring_kernel, lds, "mode_conv3d_bwd_weight");
      if (rc != MODE_OK) return rc;
      // depths per work unit: every unit starts with three un-pipelined plane loads, so as deep as the volume allows while
      // every workgroup still gets >= 2 units
      int ring_dc = D;
      while (ring_dc > 6 && (long long)B * d.nHt * d.nWt * mode::cdiv(D, ring_dc) < 2ll * d.S) ring_dc = mode::cdiv(ring_dc, 2);
      static const char* dc_env = getenv("MODE_RING_DC");  // (tuning override)
      if (dc_env && atoi(dc_env) > 0) ring_dc = atoi(dc_env);
      const int nDc = mode::cdiv(D, ring_dc);
      const int units = B * d.nHt * d.nWt * nDc;
      if (d.S > units) d.S = units;  // never more than the workspace query assumed
      hipLaunchKernelGGL(conv3d_bwd_weight_ring_kernel, dim3(d.S, d.MTo, d.MTc), dim3(NT), lds, st, gy, x, workspace, d, nDc, units,
                         ring_dc);
    } else {
      rc = mode::allow_lds(conv3d_bwd_weight_kernel<1, WTH1>, lds, "mode_conv3d_bwd_weight");
      if (rc != MODE_OK) return rc;
      hipLaunchKernelGGL((conv3d_bwd_weight_kernel<1, WTH1>), dim3(d.S, d.MTo, d.MTc), dim3(NT), lds, st, gy, x, workspace, d);
    }
  } else {
    const size_t lds = WGeom<2, WTH2>::LDS;
    // 32-bit element offsets within a sample of x (Ci channels at D x H x W) and of gy (Co channels at the output resolution)
    if (std::max((long long)Ci * D * H * W, (long long)Co * d.Do * d.Ho * d.Wo) < (1ll << 29)) {
      if (d.MTo >= 2) {  // two output-channel blocks per workgroup share the staged x tile
        const size_t lds2 = lds + (size_t)32 * WGeom<2, WTH2>::GPLANE * sizeof(float);
        rc = mode::allow_lds(conv3d_bwd_weight_s2_kernel<2>, lds2, "mode_conv3d_bwd_weight");
        if (rc != MODE_OK) return rc;
        hipLaunchKernelGGL(conv3d_bwd_weight_s2_kernel<2>, dim3(d.S, mode::cdiv(d.MTo, 2), d.MTc), dim3(2 * NT), lds2, st, gy, x, workspace, d);
      } else {
        rc = mode::allow_lds(conv3d_bwd_weight_s2_kernel<1>, lds, "mode_conv3d_bwd_weight");
        if (rc != MODE_OK) return rc;
        hipLaunchKernelGGL(conv3d_bwd_weight_s2_kernel<1>, dim3(d.S, d.MTo, d.MTc), dim3(NT), lds, st, gy, x, workspace, d);
      }
    } else {
      rc = mode::allow_lds(conv3d_bwd_weight_kernel<2, WTH2>, lds, "mode_conv3d_bwd_weight");
      if (rc != MODE_OK) return rc;
      hipLaunchKernelGGL((conv3d_bwd_weight_kernel<2, WTH2>), dim3(d.S, d.MTo, d.MTc), dim3(NT), lds, st, gy, x, workspace, d);
    }
  }
  rc = mode::check_launch("mode_conv3d_bwd_weight");
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(reduce_gw3d, dim3(mode::cdiv((long long)d.MTo * d.MTc * 27 * 1024, 32)), dim3(256), 0, st, workspace, gw, d, accumulate);
  return mode::check_launch("mode_conv3d_bwd_weight(reduce)");
}

static int bwd_weight_split(const float* gy, const float* x, const float* amax_g, const float* amax_x, float* gw, float* workspace, int B, int Ci,
                            int D, int H, int W, int Co, int accumulate, mode_stream_t stream, const char* who);

extern "C" int mode_conv3d_bwd_weight_split(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int D, int H,
                                            int W, int Co, int accumulate, mode_stream_t stream) {
  return bwd_weight_split(gy, x, nullptr, nullptr, gw, workspace, B, Ci, D, H, W, Co, accumulate, stream, "mode_conv3d_bwd_weight_split");
}

// The same on the two-piece fp16 arithmetic (mode_conv3d_fwd_split_f16): amax_g / amax_x = device floats max |gy| / max |x|
extern "C" int mode_conv3d_bwd_weight_split_f16(const float* gy, const float* x, const float* amax_g, const float* amax_x, float* gw, float* workspace,
                                                int B, int Ci, int D, int H, int W, int Co, int accumulate, mode_stream_t stream) {
  MODE_REQUIRE(amax_g && amax_x, MODE_ERR_BAD_ARG, "mode_conv3d_bwd_weight_split_f16: null operand maximum");
  return bwd_weight_split(gy, x, amax_g, amax_x, gw, workspace, B, Ci, D, H, W, Co, accumulate, stream, "mode_conv3d_bwd_weight_split_f16");
}

static int bwd_weight_split(const float* gy, const float* x, const float* amax_g, const float* amax_x, float* gw, float* workspace, int B, int Ci,
                            int D, int H, int W, int Co, int accumulate, mode_stream_t stream, const char* who) {
  int rc = check_conv_args(gy, x, gw, workspace, B, Ci, D, H, W, Co, 1, who);
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(mode_conv3d_split_supported(Ci, Co, 1, 2) == 1 && (long long)std::max(Ci, Co) * D * H * W < (1ll << 29), MODE_ERR_UNSUPPORTED,
               "%s: layer not covered by the split kernels (single output channel, or a sample beyond 2^29 elements)", who);
  hipStream_t st = mode::as_stream(stream);
  if (B == 0) {
    if (!accumulate) return mode::zero_floats(gw, (size_t)Co * Ci * 27, st, "mode_conv3d_bwd_weight");
    return MODE_OK;
  }
  WDims d;
  make_wdims(d, B, Ci, D, H, W, Co, 1);
  mode::WgradSplitDims q;
  q.Ci = Ci; q.Co = Co; q.D = D; q.H = H; q.W = W;
  q.nWt = d.nWt; q.nHt = d.nHt; q.MTo = d.MTo; q.MTc = d.MTc;
  // one workgroup per CU and (o, c) block pair (152 KB of LDS each); depths per unit as for the fp32 ring kernel
  int S = mode::cdiv(kNumCU, d.MTo * d.MTc);
  int ring_dc = D;
  while (ring_dc > 6 && (long long)B * d.nHt * d.nWt * mode::cdiv(D, ring_dc) < 2ll * S) ring_dc = mode::cdiv(ring_dc, 2);
  q.ring_dc = ring_dc;
  q.nDc = mode::cdiv(D, ring_dc);
  q.units = B * d.nHt * d.nWt * q.nDc;
  if (S > q.units) S = q.units;
  q.S = d.S = S;  // (<= the S of make_wdims, which sized the workspace)
  rc = mode::conv3d_bww_split_launch(gy, x, workspace, q, st, who, amax_g, amax_x);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(reduce_gw3d, dim3(mode::cdiv((long long)d.MTo * d.MTc * 27 * 1024, 32)), dim3(256), 0, st, workspace, gw, d, accumulate);
  return mode::check_launch("mode_conv3d_bwd_weight_split(reduce)");
}

// Weight gradient of the stride-2 convolution on the split-bf16 kernel of conv3d_split_wgrad_s2.hip: x (B, Ci, D, H, W), gy (B, Co,
// D/2, H/2, W/2); D and H even, W a multiple of 8; needs mode_conv3d_split_supported(Ci, Co, 2, 2) == 1.  With (gy := the input of a
// ConvTranspose3d, x := the gradient of its output) the result is that layer's (Cin, Cout, 27) weight gradient.
extern "C" int mode_conv3d_bwd_weight_s2_split(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int D, int H,
                                               int W, int Co, int accumulate, mode_stream_t stream) {
  const char* who = "mode_conv3d_bwd_weight_s2_split";
  int rc = check_conv_args(gy, x, gw, workspace, B, Ci, D, H, W, Co, 2, who, true);
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(mode_conv3d_split_supported(Ci, Co, 2, 2) == 1, MODE_ERR_UNSUPPORTED, "%s: %d -> %d channels are not covered by the split kernel",
               who, Ci, Co);
  MODE_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 8 == 0, MODE_ERR_UNSUPPORTED, "%s: needs even D, H and W a multiple of 8 (got %dx%dx%d)", who,
               D, H, W);
  MODE_REQUIRE(std::max((long long)std::min(Ci, 32) * D * H * W, (long long)std::min(Co, 64) * (D / 2) * (H / 2) * (W / 2)) < (1ll << 29),
               MODE_ERR_UNSUPPORTED, "%s: a channel block of one sample exceeds 2^29 elements", who);
  hipStream_t st = mode::as_stream(stream);
  if (B == 0) {
    if (!accumulate) return mode::zero_floats(gw, (size_t)Co * Ci * 27, st, "mode_conv3d_bwd_weight");
    return MODE_OK;
  }
  WDims d;
  make_wdims(d, B, Ci, D, H, W, Co, 2);
  mode::WgradS2SplitDims q;
  q.Ci = Ci; q.Co = Co; q.D = D; q.H = H; q.W = W; q.Do = d.Do; q.Ho = d.Ho; q.Wo = d.Wo;
  q.nWt = mode::cdiv(d.Wo, 16); q.MTo = d.MTo; q.MTc = d.MTc;
  // one workgroup per CU (130 KB of LDS each); a unit starts with two un-pipelined stagings, so as deep as the volume allows while
  // every workgroup still gets >= 2 units
  int S = std::max(1, kNumCU / (mode::cdiv(d.MTo, 2) * d.MTc));
  int ring_dc = d.Do;
  while (ring_dc > 3 && (long long)B * d.Ho * q.nWt * mode::cdiv(d.Do, ring_dc) < 2ll * S) ring_dc = mode::cdiv(ring_dc, 2);
  q.ring_dc = ring_dc;
  q.nDc = mode::cdiv(d.Do, ring_dc);
  q.units = B * d.Ho * q.nWt * q.nDc;
  q.S = std::min(std::min(S, q.units), d.S);  // never more partials than the workspace query assumed
  d.S = q.S;
  rc = mode::conv3d_bww_s2_split_launch(gy, x, workspace, q, st, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(reduce_gw3d, dim3(mode::cdiv((long long)d.MTo * d.MTc * 27 * 1024, 32)), dim3(256), 0, st, workspace, gw, d, accumulate);
  return mode::check_launch("mode_conv3d_bwd_weight_s2_split(reduce)");
}

// Transposed convolution k3 s2 p1 op1 (= backward-data of the stride-2 convolution).
//   x (B, Cin, D, H, W), w: `w_is_conv` = 0 -> ConvTranspose3d weight (Cin, Cout, 3,3,3);
//                          `w_is_conv` = 1 -> Conv3d weight (Cin_of_this_call = conv Co, Cout = conv Ci) i.e. (Co, Ci, 3,3,3)
//   both index as w[cin][cout][tap], so one packing mode serves both.       y (B, Cout, 2D, 2H, 2W)
extern "C" int mode_deconv3d_fwd(const float* x, const float* w, float* y, float* wpack, int B, int Cin, int D, int H, int W,
                                 int Cout, mode_stream_t stream) {
  int rc = check_conv_args(x, w, y, wpack, B, Cin, D, H, W, Cout, 1, "mode_deconv3d_fwd");
  if (rc != MODE_OK || B == 0) return rc;
  MODE_REQUIRE((long long)Cout * D * H * W * 8 < (1ll << 31), MODE_ERR_UNSUPPORTED, "mode_deconv3d_fwd: output sample too large");
  return deconv3d(x, w, y, wpack, B, Cin, Cout, D, H, W, mode::as_stream(stream), "mode_deconv3d_fwd");
}

extern "C" int mode_deconv3d_fwd_bn(const float* x, const float* w, const mode_bn_epilogue* bn, float* y, float* wpack, int B, int Cin,
                                    int D, int H, int W, int Cout, mode_stream_t stream) {
  const char* who = "mode_deconv3d_fwd_bn";
  int rc = check_conv_args(x, w, y, wpack, B, Cin, D, H, W, Cout, 1, who);
  if (rc == MODE_OK) rc = mode::check_bn(bn, who);
  if (rc != MODE_OK || B == 0) return rc;
  MODE_REQUIRE((long long)Cout * D * H * W * 8 < (1ll << 31), MODE_ERR_UNSUPPORTED, "%s: output sample too large", who);
  return deconv3d(x, w, y, wpack, B, Cin, Cout, D, H, W, mode::as_stream(stream), who, bn);
}
