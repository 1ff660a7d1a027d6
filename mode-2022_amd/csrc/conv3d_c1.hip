// Single-output-channel 3x3x3 convolution: the last layer of classif1-3, Conv3d(32 -> 1) (models/mode_disparity.py:76-80).
//
// With one output channel the implicit-GEMM tile of conv3d.hip is 31/32 padding (1.47 ms forward, 1.71 ms weight gradient at
// the benchmark volume).  The layer is really a memory-bound stencil: 201 MB of input per sample for 6.3 MB of output.
//   forward        : vector-ALU stencil; a thread owns 4 outputs along h so that one LDS read feeds up to 3 FMAs
//   weight gradient: gW[c][tap] = sum_q x[c,q] * gy[q + 1 - k]  as an MFMA GEMM with  D[i = c][j = tap]  (27 of 32 columns
//                    used): A[i = c][k = voxel] = x tile, B[k = voxel][j = tap] = the gy halo tile read at a per-lane tap
//                    offset; split-K over voxel tiles with a fixed-order reduction
//   input gradient : gx[c][u] = sum_t w[c][t] * gy[u - off(t)]  as an MFMA GEMM with  D[i = c][j = voxel]  and the 27 taps as K
#include "conv3d_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 256;

// ------------------------------------------------------------------------------------------------------- forward
constexpr int FTD = 2, FTH = 16, FCC = 4;                 // tile 2 x 16 x 32 outputs, 4 input channels per LDS chunk
constexpr int FID = FTD + 2, FIH = FTH + 2, FIW = 34;
constexpr int FPLANE = FID * FIH * FIW;                   // 2448 floats per channel
constexpr int FROWS = FCC * FID * FIH;                    // 288 rows of 34

__global__ __launch_bounds__(NT) void conv3d_co1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ y, int B, int Ci, int D, int H, int W, int nDt,
                                                            int nHt, int nWt) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [FCC][FID][FIH][FIW]
  int* rowtab = reinterpret_cast<int*>(tile + FCC * FPLANE);
  int t = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring tiles (shared halos) on one XCD
  const int wt = t % nWt;
  t /= nWt;
  const int ht = t % nHt;
  t /= nHt;
  const int dt = t % nDt;
  const int b = t / nDt;
  const int w0 = wt * 32, h0 = ht * FTH, d0 = dt * FTD;
  const int tid = threadIdx.x;
  const int dz = tid >> 7, hq = (tid >> 5) & 3, wx = tid & 31;
  const long long HW = (long long)H * W, DHW = (long long)D * HW;
  const float* xb = x + (long long)b * Ci * DHW;

  for (int r = tid; r < FROWS; r += NT) {
    const int c = r / (FID * FIH), rem = r - c * (FID * FIH);
    const int gd = d0 + rem / FIH - 1, gh = h0 + rem % FIH - 1;
    rowtab[r] = (gd >= 0 && gd < D && gh >= 0 && gh < H) ? (int)(c * DHW + gd * HW + gh * W) : -1;
  }
  __syncthreads();

  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int hwv = tid >> 5, l32 = tid & 31;
  const int nchunk = (Ci + FCC - 1) / FCC;
  for (int ch = 0; ch < nchunk; ++ch) {
    const float* xc = xb + (long long)ch * FCC * DHW;
#pragma unroll 1
    for (int kb = 0; kb < FROWS; kb += 64) {
      float t8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = kb + j * 8 + hwv;
        const int off = r < FROWS ? rowtab[r] : -1;
        const int gw = w0 + l32;
        const bool ok = off >= 0 && gw < W && ch * FCC + r / (FID * FIH) < Ci;
        const float v = xc[ok ? off + gw : 0];
        t8[j] = ok ? v : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = kb + j * 8 + hwv;
        if (r < FROWS) tile[r * FIW + 1 + l32] = t8[j];
      }
    }
#pragma unroll 1
    for (int item = tid; item < FROWS * 2; item += NT) {
      const int r = item >> 1, side = item & 1;
      const int off = rowtab[r];
      const int gw = side ? w0 + 32 : w0 - 1;
      const bool ok = off >= 0 && gw >= 0 && gw < W && ch * FCC + r / (FID * FIH) < Ci;
      const float v = xc[ok ? off + gw : 0];
      tile[r * FIW + (side ? 33 : 0)] = ok ? v : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < FCC; ++c) {
      const int cin = min(ch * FCC + c, Ci - 1);  // out-of-range channels were staged as zeros; keep the weight read valid
      const float* wc = w + cin * 27;             // wave-uniform: scalar loads
      const float* tp = tile + c * FPLANE + dz * (FIH * FIW) + (hq * 4) * FIW + wx;
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float v = tp[kd * (FIH * FIW) + r * FIW + kw];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              const int o = r - kh;
              if (o >= 0 && o < 4) acc[o] += wc[kd * 9 + kh * 3 + kw] * v;
            }
          }
    }
    __syncthreads();
  }
  const int gd = d0 + dz, gw = w0 + wx;
  if (gd < D && gw < W) {
    float* yb = y + (long long)b * DHW + gd * HW + gw;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int gh = h0 + hq * 4 + o;
      if (gh < H) yb[(long long)gh * W] = acc[o];
    }
  }
}

// ------------------------------------------------------------------------------------------------------- forward on MFMA
// The vector-ALU stencil above needs 864 FMAs per output voxel and runs at 18 TFLOP/s (0.31 ms for 403 MB of input).  MFMA
// form with the 27 taps as the GEMM-M dimension:  Z[t][p] = sum_c w[c][t] * x[c][p]  for every INPUT voxel p (A[i = t][k = c] in
// 16 registers, B[k = c][j = 32 voxels along w] straight from global memory in fragment layout, no LDS staging of x), then
// y[u] = sum_t Z[t][u + off(t)]: the Z planes of one input depth go through LDS and every thread gathers 9 values per kd
// for its outputs; the three depths an output needs arrive on consecutive steps of a rolling loop over the input planes
// (kd = 0 from plane d-1, kd = 1 from d, kd = 2 from d+1), carried in two registers per output.  Fixed summation order.
__device__ __forceinline__ f32x16 mfma32b(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

constexpr int ZTH = 16, ZIH = ZTH + 2, ZIW = 34, ZPL = ZIH * ZIW;  // 16 x 32 outputs per plane; Z tile [27][18][34] = 66 KB
constexpr int ZDC = 12;                                            // output depths per work unit (2 halo planes on top)
static_assert(ZIH + 2 == 4 * 5, "18 row groups + 36 halo-column positions in 2 groups = 5 groups per wave");

template <bool ONE>
__global__ __launch_bounds__(NT, 2) void conv3d_co1_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 float* __restrict__ y, int B, int Ci, int D, int H, int W, int nDc,
                                                                 int nHt, int nWt) {
  extern __shared__ __attribute__((aligned(16))) float zl[];  // [27][ZIH][ZIW]
  int t = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring tiles (shared halos) on one XCD
  const int wt = t % nWt;
  t /= nWt;
  const int ht = t % nHt;
  t /= nHt;
  const int dc = t % nDc;
  const int b = t / nDc;
  const int w0 = wt * 32, h0 = ht * ZTH, dlo = dc * ZDC, dhi = min(D, dlo + ZDC);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, kh = lane >> 5;
  const long long HW = (long long)H * W, DHW = (long long)D * HW;
  const float* xb = x + (long long)b * Ci * DHW;
  const int NCB = (Ci + 31) >> 5;

  // A fragments of channel block 0 (the only one when Ci <= 32): A[i = tap j][k = channel 2 ks + kh]
  float a0[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int c = 2 * ks + kh;
    a0[ks] = (j < 27 && c < Ci) ? w[c * 27 + j] : 0.f;
  }
  // this wave's 5 position groups: global (row, column) offset, validity, LDS column
  int poff[5], pz[5];
  bool pok[5];
#pragma unroll
  for (int g5 = 0; g5 < 5; ++g5) {
    const int g = wave * 5 + g5;
    int row, gw, zc;
    bool in = true;
    if (g < ZIH) {
      row = g;
      gw = w0 + j;
      zc = 1 + j;
    } else {
      const int p = (g - ZIH) * 32 + j;  // halo columns: position p = (row, side)
      in = p < 2 * ZIH;
      row = in ? p >> 1 : 0;
      gw = (p & 1) ? w0 + 32 : w0 - 1;
      zc = (p & 1) ? 33 : 0;
    }
    const int gh = h0 + row - 1;
    pok[g5] = in && gh >= 0 && gh < H && gw >= 0 && gw < W;
    poff[g5] = pok[g5] ? gh * W + gw : 0;
    pz[g5] = in ? row * ZIW + zc : -1;
  }
  const int wx = tid & 31, hq = tid >> 5;  // gather: outputs (hq, wx) and (hq + 8, wx)
  float om1[2] = {0.f, 0.f}, o0[2] = {0.f, 0.f};

  // B fragments of one input plane (channel block 0): 5 groups x 16 channel pairs per lane, loaded unconditionally from
  // clamped addresses.  With ONE (Ci <= 32) the next plane's fragments are requested right after the MFMAs of the current one,
  // so they travel while the Z tile goes through LDS.
  float bv[5][16];
  auto load_group = [&](int dz, int g5) {
    const float* xp = xb + (long long)dz * HW;  // uniform base + 32-bit lane offsets (host: Ci * D * H * W < 2^30)
    unsigned dhw = (unsigned)DHW;
    asm volatile("" : "+s"(dhw));  // opaque: keeps the 80 lane offsets from being hoisted out of the plane loop as 80 live registers
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      bv[g5][ks] = xp[(pok[g5] && 2 * ks + kh < Ci) ? (unsigned)poff[g5] + (unsigned)(2 * ks + kh) * dhw : 0u];
  };
  if (ONE && dlo - 1 >= 0) {
#pragma unroll
    for (int g5 = 0; g5 < 5; ++g5) load_group(dlo - 1, g5);
  }

  for (int dz = dlo - 1; dz <= dhi; ++dz) {
    float s[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    const bool valid = dz >= 0 && dz < D;                 // (block-uniform)
    const bool more = ONE && dz + 1 <= dhi && dz + 1 < D;  // the next plane exists: request it group by group
    if (!valid && more) {
#pragma unroll
      for (int g5 = 0; g5 < 5; ++g5) load_group(dz + 1, g5);
    }
    if (valid) {
      if (!ONE) {
#pragma unroll
        for (int g5 = 0; g5 < 5; ++g5) load_group(dz, g5);
      }
      f32x16 acc[5];
#pragma unroll
      for (int g5 = 0; g5 < 5; ++g5) acc[g5] = (f32x16){0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)  // the 5 groups interleaved: consecutive MFMAs never wait on each other's accumulator
#pragma unroll
        for (int g5 = 0; g5 < 5; ++g5) acc[g5] = mfma32b(a0[ks], (pok[g5] && 2 * ks + kh < Ci) ? bv[g5][ks] : 0.f, acc[g5]);
      if (more) {  // the fragments are consumed: fetch the next plane into their registers
#pragma unroll
        for (int g5 = 0; g5 < 5; ++g5) load_group(dz + 1, g5);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!ONE) {  // further channel blocks (Ci > 32): plain loop
        for (int cb = 1; cb < NCB; ++cb) {
#pragma unroll
          for (int g5 = 0; g5 < 5; ++g5) {
            const float* xp = xb + ((long long)cb * 32 + kh) * DHW + (long long)dz * HW + poff[g5];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
              const int c = cb * 32 + 2 * ks + kh;
              const bool ok = pok[g5] && c < Ci;
              const float xv = *(ok ? xp + 2 * ks * DHW : xb);  // (not xp[0]: xp itself may lie past the last channel / outside the plane)
              const float av = (j < 27 && c < Ci) ? w[c * 27 + j] : 0.f;
              acc[g5] = mfma32b(av, ok ? xv : 0.f, acc[g5]);
            }
          }
        }
      }
#pragma unroll
      for (int g5 = 0; g5 < 5; ++g5) {
        if (pz[g5] >= 0) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int i = (q & 3) + 8 * (q >> 2) + 4 * kh;
            if (i < 27) zl[i * ZPL + pz[g5]] = acc[g5][q];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const float* zp = zl + (hq + 8 * o) * ZIW + wx;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
          float v = 0.f;
#pragma unroll
          for (int k9 = 0; k9 < 9; ++k9) v += zp[(kd * 9 + k9) * ZPL + (k9 / 3) * ZIW + (k9 % 3)];
          s[kd][o] = v;
        }
      }
      __syncthreads();
    }
    const int dout = dz - 1;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int gh = h0 + hq + 8 * o, gw = w0 + wx;
      if (dout >= dlo && dout < dhi && gh < H && gw < W) y[(long long)b * DHW + dout * HW + (long long)gh * W + gw] = om1[o] + s[2][o];
      om1[o] = o0[o] + s[1][o];
      o0[o] = s[0][o];
    }
  }
}

// ------------------------------------------------------------------------------------------------------- input gradient
// gx[c][u] = sum_t w[c][t] * gy[u - off(t)] = sum_k w[c][26 - k] * gy[u + off(k)]   (off(k) = (kd-1, kh-1, kw-1)):
// D[i = c][j = 32 voxels along w], K = the 27 taps (28 with a zero column): A[i = c][k] = w[c][26 - k] lives in 14 registers,
// B[k][j] = the single-channel gy halo tile (5.4 KB of LDS) read at the per-lane offset of tap k.  14 MFMAs per 32 voxels x
// 32 channels -- the generic kernel spends 108 on its zero-padded 8-channel chunk -- so the kernel is bound by the
// 201 MB per sample of gx it writes.
constexpr int BTD = 2, BTH = 8;                           // tile 2 x 8 x 32 voxels, 4 rows per wave
constexpr int BID = BTD + 2, BIH = BTH + 2, BIW = 34;     // gy halo tile [4][10][34]

// grid = (tiles, ceil(Ci/32))
__global__ __launch_bounds__(NT) void conv3d_co1_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                 float* __restrict__ gx, int B, int Ci, int D, int H, int W, int nDt,
                                                                 int nHt, int nWt) {
  __shared__ float tile[BID * BIH * BIW];
  int t = xcd_remap(blockIdx.x, gridDim.x);  // neighbouring tiles (shared halos) on one XCD
  const int wt = t % nWt;
  t /= nWt;
  const int ht = t % nHt;
  t /= nHt;
  const int dt = t % nDt;
  const int b = t / nDt;
  const int cb = blockIdx.y;
  const int w0 = wt * 32, h0 = ht * BTH, d0 = dt * BTD;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long long HW = (long long)H * W, DHW = (long long)D * HW;
  const float* gb = gy + (long long)b * DHW;

  for (int idx = tid; idx < BID * BIH * BIW; idx += NT) {
    const int dz = idx / (BIH * BIW), rem = idx - dz * (BIH * BIW);
    const int hy = rem / BIW, wx = rem - hy * BIW;
    const int gd = d0 + dz - 1, gh = h0 + hy - 1, gw = w0 + wx - 1;
    const bool ok = gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
    const float v = gb[ok ? gd * HW + gh * W + gw : 0];
    tile[idx] = ok ? v : 0.f;
  }
  float a[14];
  int toff[14];
  const int c = cb * 32 + (lane & 31);
#pragma unroll
  for (int ks = 0; ks < 14; ++ks) {
    const int k = 2 * ks + (lane >> 5);
    a[ks] = (k < 27 && c < Ci) ? w[c * 27 + 26 - k] : 0.f;
    const int kk = k < 27 ? k : 26;
    toff[ks] = (kk / 9) * (BIH * BIW) + ((kk / 3) % 3) * BIW + kk % 3;
  }
  __syncthreads();

  f32x16 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (f32x16){0};
  const float* tp = tile + (lane & 31);
#pragma unroll
  for (int ks = 0; ks < 14; ++ks)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 4 + r;  // (dz, hy) = (row / BTH, row % BTH)
      acc[r] = mfma32b(a[ks], tp[(row / BTH) * (BIH * BIW) + (row % BTH) * BIW + toff[ks]], acc[r]);
    }

  float* gxb = gx + ((long long)b * Ci + cb * 32) * DHW;
  const int gw = w0 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = wave * 4 + r;
    const int gd = d0 + row / BTH, gh = h0 + row % BTH;
    if (gd < D && gh < H && gw < W) {
      const long long sp = gd * HW + (long long)gh * W + gw;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (cb * 32 + i < Ci) gxb[i * DHW + sp] = acc[r][q];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------- weight gradient
constexpr int GTH = 8;                      // tile 1 x 8 x 32 input voxels = 128 k-steps, 32 per wave
constexpr int XS = GTH * 32 + 1;            // 257: odd channel stride -> conflict-free A fragments
constexpr int GH = GTH + 2, GW = 34, GPL = GH * GW;  // gy halo tile [3][10][34]

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// grid = (S, ceil(Ci/32)); part[(s*MTc + cb)*1024 + c*32 + tap]
__global__ __launch_bounds__(NT) void conv3d_co1_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                   float* __restrict__ part, int B, int Ci, int D, int H, int W,
                                                                   int nHt, int nWt, int T, int S) {
  __shared__ float xl[32 * XS];
  __shared__ float gl[3 * GPL];
  __shared__ float red[4 * 1024];
  const int s = blockIdx.x, cb = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int hwv = tid >> 5, l32 = tid & 31;
  const long long HW = (long long)H * W, DHW = (long long)D * HW;
  const int j = lane & 31;                     // this lane's tap column
  const int tap = j < 27 ? j : 0;
  const int toff = (2 - tap / 9) * GPL + (2 - (tap / 3) % 3) * GW + (2 - tap % 3);
  f32x16 acc = {0};

  for (int tt = s; tt < T; tt += S) {
    int t = tt;
    const int wt = t % nWt;
    t /= nWt;
    const int ht = t % nHt;
    t /= nHt;
    const int d0 = t % D;
    const int b = t / D;
    const int w0 = wt * 32, h0 = ht * GTH;
    const float* xb = x + ((long long)b * Ci + cb * 32) * DHW + d0 * HW;
    const float* gb = gy + (long long)b * DHW;
    // x tile: 32 channels x 8 rows x 32 voxels, one coalesced half-wave load per (channel, row)
#pragma unroll 1
    for (int kb = 0; kb < 256; kb += 64) {
      float t8[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int r = kb + jj * 8 + hwv;  // (channel, row) = (r >> 3, r & 7)
        const int c = r >> 3, gh = h0 + (r & 7), gw = w0 + l32;
        const bool ok = cb * 32 + c < Ci && gh < H && gw < W;
        const float v = xb[ok ? c * DHW + (long long)gh * W + gw : 0];
        t8[jj] = ok ? v : 0.f;
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int r = kb + jj * 8 + hwv;
        xl[(r >> 3) * XS + (r & 7) * 32 + l32] = t8[jj];
      }
    }
    // gy halo tile G[zz][yy][xx] = gy[d0-1+zz][h0-1+yy][w0-1+xx]
    for (int idx = tid; idx < 3 * GPL; idx += NT) {
      const int zz = idx / GPL, rem = idx - zz * GPL;
      const int yy = rem / GW, xx = rem - yy * GW;
      const int gd = d0 - 1 + zz, gh = h0 - 1 + yy, gw = w0 - 1 + xx;
      const bool ok = gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
      const float v = gb[ok ? gd * HW + (long long)gh * W + gw : 0];
      gl[idx] = ok ? v : 0.f;
    }
    __syncthreads();
    // wave v: rows 2v, 2v+1; A[i = c][k] = x[c][row][2ks + (lane>>5)], B[k][j = tap] = G[toff_j + row*34 + 2ks + (lane>>5)]
    const float* ap = xl + (lane & 31) * XS + (lane >> 5);
    const float* bp = gl + toff + (lane >> 5);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr;
#pragma unroll 8
      for (int ks = 0; ks < 16; ++ks) acc = mfma32(ap[row * 32 + 2 * ks], bp[row * GW + 2 * ks], acc);
    }
    __syncthreads();
  }
  // cross-wave reduction in a fixed order, then one 32x32 partial per workgroup
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
    red[wave * 1024 + i * 32 + (lane & 31)] = acc[q];
  }
  __syncthreads();
  float* pb = part + ((long long)s * gridDim.y + cb) * 1024;
  for (int idx = tid; idx < 1024; idx += NT) pb[idx] = (red[idx] + red[1024 + idx]) + (red[2048 + idx] + red[3072 + idx]);
}

__global__ void conv3d_co1_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int Ci, int S, int MTc,
                                         int accumulate) {
  // one wave per output element (c, tap): lane l sums slices l, l + 64, ... in order, then a fixed shuffle tree combines the 64
  // lane sums -- deterministic, and 64-way parallel instead of one thread walking ~1000 slices
  const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // c*27 + tap
  const int lane = threadIdx.x & 63;
  if (idx >= Ci * 27) return;
  const int c = idx / 27, tap = idx - c * 27;
  const float* p = part + (long long)(c / 32) * 1024 + (c % 32) * 32 + tap;
  float sum = 0.f;
  for (int s = lane; s < S; s += 64) sum += p[(long long)s * MTc * 1024];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if (lane == 0) gw[idx] = accumulate ? gw[idx] + sum : sum;
}

int co1_splits(int T, int MTc) {
  int S = mode::cdiv(4 * kNumCU, MTc);
  if (S > T) S = T;
  return S < 1 ? 1 : S;
}

}  // namespace

namespace mode {

int conv3d_co1_fwd(const float* x, const float* w, float* y, int B, int Ci, int D, int H, int W, hipStream_t st, const char* who) {
  if (Ci <= 32 && (long long)Ci * D * H * W < (1ll << 30)) return conv3d_co1_fwd_small(x, w, y, B, Ci, D, H, W, st, who);
  if ((long long)Ci * D * H * W < (1ll << 30)) {  // MFMA form (the vector-ALU stencil below is kept for larger samples)
    const int nDc = cdiv(D, ZDC), nHt = cdiv(H, ZTH), nWt = cdiv(W, 32);
    const size_t lds = (size_t)27 * ZPL * sizeof(float);
    auto kern = conv3d_co1_fwd_mfma_kernel<false>;  // (Ci > 32 here: up to 32 channels run on classif_head.hip's kernel, above)
    int rc = allow_lds(kern, lds, who);
    if (rc != MODE_OK) return rc;
    hipLaunchKernelGGL(kern, dim3(B * nDc * nHt * nWt), dim3(NT), lds, st, x, w, y, B, Ci, D, H, W, nDc, nHt, nWt);
    return check_launch(who);
  }
  const int nDt = cdiv(D, FTD), nHt = cdiv(H, FTH), nWt = cdiv(W, 32);
  const size_t lds = (size_t)FCC * FPLANE * sizeof(float) + (size_t)FROWS * sizeof(int);
  hipLaunchKernelGGL(conv3d_co1_fwd_kernel, dim3(B * nDt * nHt * nWt), dim3(NT), lds, st, x, w, y, B, Ci, D, H, W, nDt, nHt, nWt);
  return check_launch(who);
}

int conv3d_co1_bwd_data(const float* gy, const float* w, float* gx, int B, int Ci, int D, int H, int W, hipStream_t st,
                        const char* who) {
  const int nDt = cdiv(D, BTD), nHt = cdiv(H, BTH), nWt = cdiv(W, 32);
  hipLaunchKernelGGL(conv3d_co1_bwd_data_kernel, dim3(B * nDt * nHt * nWt, cdiv(Ci, 32)), dim3(NT), 0, st, gy, w, gx, B, Ci, D, H, W,
                     nDt, nHt, nWt);
  return check_launch(who);
}

size_t conv3d_co1_bwd_weight_workspace_floats(int B, int Ci, int D, int H, int W) {
  const int MTc = cdiv(Ci, 32);
  const int T = B * D * cdiv(H, GTH) * cdiv(W, 32);
  return (size_t)co1_splits(T, MTc) * MTc * 1024;
}

int conv3d_co1_bwd_weight(const float* gy, const float* x, float* gw, float* workspace, int B, int Ci, int D, int H, int W,
                          int accumulate, hipStream_t st, const char* who) {
  const int MTc = cdiv(Ci, 32);
  const int nHt = cdiv(H, GTH), nWt = cdiv(W, 32);
  const int T = B * D * nHt * nWt;
  const int S = co1_splits(T, MTc);
  hipLaunchKernelGGL(conv3d_co1_bwd_weight_kernel, dim3(S, MTc), dim3(NT), 0, st, gy, x, workspace, B, Ci, D, H, W, nHt, nWt, T, S);
  int rc = check_launch(who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(conv3d_co1_reduce_kernel, dim3(cdiv(Ci * 27, 4)), dim3(256), 0, st, workspace, gw, Ci, S, MTc, accumulate);
  return check_launch(who);
}

}  // namespace mode
