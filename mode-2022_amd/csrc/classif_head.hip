// The classifier heads of the 3-D regulariser in TRAINING, with the BatchNorm + ReLU between their two convolutions never written:
//   classifN = Sequential(convbn_3d(32, 32), ReLU, Conv3d(32, 1))   (models/mode_disparity.py:76-80, 127-129; convbn_3d submodule.py:20-22)
//
// y = conv3d(out_N) (B, C, D, H, W) is what the first convolution leaves in HBM (403 MB at the benchmark shape).  The composition of separate
// operators then read y twice and wrote a = relu(bn(y)) once (BatchNorm), read a (32 -> 1 convolution), and in the backward read a
// (its weight gradient), wrote g = dL/da (its input gradient), read g + y twice and wrote dL/dy (BatchNorm backward): 13 passes over
// 403 MB per head and step.  Here:
//   forward   statistics pass over y (bn_act.hip) + `classif_fwd_kernel`: the single-channel convolution of conv3d_c1.hip with
//             a = relu(fma(y, scale[c], shift[c])) formed in registers as the B fragments arrive, and the residual add of
//             `cost2 = classif2(out2) + cost1` (mode_disparity.py:128-129) in its store: 2 passes;
//   backward  `classif_bww_kernel`: ONE pass over y gives the weight gradient of the 32 -> 1 convolution AND both sums of the BatchNorm
//             backward.  With m = [a > 0], g1 = dL/dcost and the tap offset off(t):
//                 Q[c][t] = sum_q m[c][q] (y[c][q] - mean[c]) g1[q - off(t)],      M[c][t] = sum_q m[c][q] g1[q - off(t)]
//             (two GEMMs with K = voxels over the same staged tiles) give
//                 gW[c][t] = scale[c] Q[c][t] + beta[c] M[c][t]                 (a = scale (y - mean) + beta on the mask)
//                 sum_q m g (y - mean) = sum_t w[c][t] Q[c][t],   sum_q m g = sum_t w[c][t] M[c][t]     (g = dL/da = sum_t w[c][t] g1[q - off(t)])
//             -- the two reductions bn_bwd_stats_kernel takes over g and y, without g ever existing;
//             `classif_bwd_apply_kernel`: the input gradient of the 32 -> 1 convolution (conv3d_co1_bwd_data: K = the 27 taps) with the
//             BatchNorm backward's apply pass in its store, dL/dy = A m g + Bc y + Cc: reads y, writes dL/dy: 3 passes in all.
// HBM-bound by design: 403 MB of y per pass against 12.6 MB of everything else (B = 2, 32 x 48 x 256 x 128).
#include "bn_internal.h"
#include "conv3d_internal.h"

#include <algorithm>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 256;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// ------------------------------------------------------------------------------------------------------- forward
// conv3d_co1_fwd_mfma_kernel<true> (conv3d_c1.hip: the 27 taps as GEMM-M, Z[t][p] = sum_c w[c][t] a[c][p] per INPUT voxel p straight from
// global memory in fragment layout, then the 27-term shift-and-add through LDS with a rolling depth loop) with a = relu(fma(y, sc, sh)).
constexpr int ZTH = 16, ZIH = ZTH + 2, ZIW = 34, ZPL = ZIH * ZIW;
constexpr int ZDC = 12;
static_assert(ZIH + 2 == 4 * 5, "18 row groups + 36 halo-column positions in 2 groups = 5 groups per wave");

// BNRELU false: the plain single-channel convolution (mode_conv3d_fwd with Co = 1, Ci <= 32: eval mode, other callers).
// C32: exactly 32 channels (the network's heads): request addresses without vector arithmetic, see below; otherwise Ci <= 32, clamped.
template <bool BNRELU, bool C32>
__global__ __launch_bounds__(NT, 2) void classif_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ add, float* __restrict__ y, int B, int Ci, int D,
                                                            int H, int W, int nDc, int nHt, int nWt) {
  extern __shared__ __attribute__((aligned(16))) float zl[];  // [27][ZIH][ZIW], then 32 (scale, shift) pairs
  float2* coefl = reinterpret_cast<float2*>(zl + 27 * ZPL);
  int t = xcd_remap(blockIdx.x, gridDim.x);
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

  if (BNRELU && tid < 32) coefl[tid] = tid < Ci ? make_float2(scale[tid], shift[tid]) : make_float2(0.f, 0.f);
  // K-step ks of the MFMA chain contracts the channel pair (ks, ks + 16): lanes 0..31 supply channel ks, lanes 32..63 channel ks + 16.
  // (Any pairing works as long as A and B agree; with this one the channel offset of a request is ks * D*H*W for the WHOLE wave, i.e. it
  // goes into the scalar base of the load and costs no vector instruction: the pairing (2 ks, 2 ks + 1) needed a multiply-add per request,
  // 160 of the ~700 vector instructions a wave issued per plane beside its 80 MFMAs.)
  float a0[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int c = ks + 16 * kh;
    a0[ks] = (j < 27 && c < Ci) ? w[min(c, Ci - 1) * 27 + min(j, 26)] : 0.f;
  }
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
  if (C32) {  // the upper half-wave's channels start 16 planes further
#pragma unroll
    for (int g5 = 0; g5 < 5; ++g5) poff[g5] += (int)((unsigned)(16 * kh) * (unsigned)DHW);
  }
  const int wx = tid & 31, hq = tid >> 5;
  float om1[2] = {0.f, 0.f}, o0[2] = {0.f, 0.f};

  float bv[5][16];
  auto load_group = [&](int dz, int g5) {
    const float* xp = xb + (long long)dz * HW;
    // UNCONDITIONAL loads: positions outside the volume read voxel 0 of the half-wave's first channel and are zeroed by a select before
    // the MFMA, channels beyond Ci read a lower channel against a zero weight (a0).  A condition in the address (`ok ? off : 0` with a
    // short-circuit &&) made every one of the 80 loads its own exec-masked basic block, and the vmcnt bookkeeping across those blocks
    // degenerated to vmcnt(0).  Address = wave-uniform base (plane, K-step channel) + this lane's 32-bit position offset.
    if (C32) {
      const unsigned pbyte = (unsigned)poff[g5] << 2;  // (byte offset: the scalar-base + 32-bit-lane-offset form of global_load)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        bv[g5][ks] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xp + (long long)ks * DHW) + pbyte);
    } else {
      unsigned dhw = (unsigned)DHW;
      asm volatile("" : "+s"(dhw));  // opaque: the 80 lane offsets are recomputed per plane, not kept live across the loop
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) bv[g5][ks] = xp[(unsigned)poff[g5] + (unsigned)min(ks + 16 * kh, Ci - 1) * dhw];
    }
  };
  // The planes this unit reads: dlo - 1 .. dhi, clipped to the volume (a plane outside it contributes nothing).  Every iteration issues
  // the SAME sequence of memory instructions -- the vector-memory counter is in-order and its waits are static counts, so one conditional
  // request in the loop turns every later wait into vmcnt(0) and the prefetch into a drain.
  const int dstart = max(dlo - 1, 0), dend = min(dhi, D - 1);
#pragma unroll
  for (int g5 = 0; g5 < 5; ++g5) load_group(dstart, g5);
  __syncthreads();  // coefl
  const int gwx = w0 + wx;
  long long oidx[2];
  bool ook[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int gh = h0 + hq + 8 * o;
    ook[o] = gh < H && gwx < W;
    oidx[o] = ook[o] ? (long long)b * DHW + (long long)gh * W + gwx : 0;
  }

  for (int dz = dstart; dz <= dend; ++dz) {
    float s[3][2];
    const int dout = dz - 1;
    // the residual of the plane this iteration completes: requested first, so that it is OLDER than the 80 fragment loads below
    float addv[2] = {0.f, 0.f};
    if (add) {
#pragma unroll
      for (int o = 0; o < 2; ++o) addv[o] = add[oidx[o] + (long long)max(dout, 0) * HW];
    }
    // Group after group: the 16 MFMAs of a position group form one accumulator chain (issue interval = dependent latency = 64 cycles
    // for 32x32x2 f32: no stall), and as soon as they are issued the group's 16 registers are re-requested for the NEXT plane -- those
    // loads then travel under the MFMAs of the other four groups and the LDS phases (~7k cycles) instead of only under the LDS phases:
    // with all 80 requests behind the last MFMA a plane was a load phase (bandwidth share of the CU: ~16k cycles for both workgroups'
    // 156 KB) followed by a matrix phase (~10k), 30k in all.  (The last iteration re-requests its own plane: L2 hits, results unused.)
    f32x16 acc[5];
    int co = 16 * kh;
    const int dnext = min(dz + 1, dend);
#pragma unroll
    for (int g5 = 0; g5 < 5; ++g5) {
      // opaque INDEX (not pointer: an opaque pointer loses its address space and the reads become flat loads, which count on the
      // vector-memory counter too): the 16 coefficient pairs are re-read from LDS per group, not hoisted as 32 live registers
      asm volatile("" : "+v"(co));
      acc[g5] = (f32x16){0};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (BNRELU) {
          const float2 cf = coefl[co + ks];
          float a = __builtin_fmaf(bv[g5][ks], cf.x, cf.y);  // the BatchNorm apply pass's own expression (bn_act.hip)
          asm("" : "+v"(a));  // keeps neighbouring elements from being SLP-packed into v_pk_fma_f32 (does not overlap with MFMAs)
          acc[g5] = mfma32(a0[ks], pok[g5] ? relu_nan(a) : 0.f, acc[g5]);
        } else {
          acc[g5] = mfma32(a0[ks], pok[g5] ? bv[g5][ks] : 0.f, acc[g5]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      load_group(dnext, g5);
      __builtin_amdgcn_sched_barrier(0);
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
        for (int k9 = 0; k9 < 9; ++k9) {
          v += zp[(kd * 9 + k9) * ZPL + (k9 / 3) * ZIW + (k9 % 3)];
          asm volatile("" : "+v"(v));  // (the six chains of a thread stay scalar: no v_pk_add_f32)
        }
        s[kd][o] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      if (dout >= dlo && ook[o]) y[oidx[o] + (long long)dout * HW] = om1[o] + s[2][o] + addv[o];
      om1[o] = o0[o] + s[1][o];
      o0[o] = s[0][o];
    }
  }
  if (dend < dhi) {  // dhi == D: the plane behind the volume is zero, output plane D - 1 is complete
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      if (ook[o]) {
        const long long idx = oidx[o] + (long long)dend * HW;
        y[idx] = add ? om1[o] + add[idx] : om1[o];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------- backward, pass 1
// conv3d_co1_bwd_weight_kernel (conv3d_c1.hip: D[i = c][j = tap], K = voxels; A = the staged y tile, B = the g1 halo tile read at a per-lane
// tap offset) with TWO accumulators: A1 = m (y - mean), A2 = m, m = [fma(y, sc, sh) > 0] formed when the fragment is read (lane = channel,
// so mean / sc / sh are three registers).  Voxels outside the volume are staged as NaN: their mask is false.
constexpr int GTH = 8;
constexpr int XS = GTH * 32 + 1;
constexpr int GH = GTH + 2, GW = 34, GPL = GH * GW;

// grid = S workgroups; part[s * 2048 + {0: Q, 1024: M} + c * 32 + tap]
__global__ __launch_bounds__(NT) void classif_bww_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, float* __restrict__ part, int B, int Ci, int D,
                                                         int H, int W, int nHt, int nWt, int T, int S) {
  __shared__ __attribute__((aligned(16))) float xl[32 * XS];  // 32 896 B; the cross-wave reduction (2 x 4 x 1024 floats) reuses it
  __shared__ float gl[3 * GPL];
  const int s = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int hwv = tid >> 5, l32 = tid & 31;
  const long long HW = (long long)H * W, DHW = (long long)D * HW;
  const int j = lane & 31;  // this lane's tap column (B, D) and channel row (A)
  const int tap = j < 27 ? j : 0;
  const int toff = (2 - tap / 9) * GPL + (2 - (tap / 3) % 3) * GW + (2 - tap % 3);
  const float mu = j < Ci ? mean[j] : 0.f, sc = j < Ci ? scale[j] : 0.f, sh = j < Ci ? shift[j] : 0.f;
  const float nanv = __builtin_nanf("");
  f32x16 accq = {0}, accm = {0};

  for (int tt = s; tt < T; tt += S) {
    int t = tt;
    const int wt = t % nWt;
    t /= nWt;
    const int ht = t % nHt;
    t /= nHt;
    const int d0 = t % D;
    const int b = t / D;
    const int w0 = wt * 32, h0 = ht * GTH;
    const float* xb = x + (long long)b * Ci * DHW + d0 * HW;
    const float* gb = gy + (long long)b * DHW;
#pragma unroll 1
    for (int kb = 0; kb < 256; kb += 64) {
      float t8[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int r = kb + jj * 8 + hwv;  // (channel, row) = (r >> 3, r & 7)
        const int c = r >> 3, gh = h0 + (r & 7), gw = w0 + l32;
        const bool ok = c < Ci && gh < H && gw < W;
        const float v = xb[ok ? c * DHW + (long long)gh * W + gw : 0];
        t8[jj] = ok ? v : nanv;
      }
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int r = kb + jj * 8 + hwv;
        xl[(r >> 3) * XS + (r & 7) * 32 + l32] = t8[jj];
      }
    }
    for (int idx = tid; idx < 3 * GPL; idx += NT) {
      const int zz = idx / GPL, rem = idx - zz * GPL;
      const int yy = rem / GW, xx = rem - yy * GW;
      const int gd = d0 - 1 + zz, gh = h0 - 1 + yy, gw = w0 - 1 + xx;
      const bool ok = gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
      const float v = gb[ok ? gd * HW + (long long)gh * W + gw : 0];
      gl[idx] = ok ? v : 0.f;
    }
    __syncthreads();
    const float* ap = xl + (lane & 31) * XS + (lane >> 5);
    const float* bp = gl + toff + (lane >> 5);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int row = wave * 2 + rr;
#pragma unroll 8
      for (int ks = 0; ks < 16; ++ks) {
        const float yv = ap[row * 32 + 2 * ks], gv = bp[row * GW + 2 * ks];
        const bool m = __builtin_fmaf(yv, sc, sh) > 0.f;  // false for the NaN of a voxel outside the volume
        accq = mfma32(m ? yv - mu : 0.f, gv, accq);
        accm = mfma32(m ? 1.f : 0.f, gv, accm);
      }
    }
    __syncthreads();
  }
  // cross-wave reduction in a fixed order, then two 32 x 32 partials per workgroup
  float* red = xl;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
    red[wave * 1024 + i * 32 + (lane & 31)] = accq[q];
    red[4096 + wave * 1024 + i * 32 + (lane & 31)] = accm[q];
  }
  __syncthreads();
  float* pb = part + (long long)s * 2048;
  for (int idx = tid; idx < 2048; idx += NT) {
    const float* r = red + (idx >> 10) * 4096 + (idx & 1023);
    float lo = r[0] + r[1024], hi = r[2048] + r[3072];
    asm volatile("" : "+v"(lo), "+v"(hi));  // (keeps the unrolled iterations from being packed into v_pk_add_f32)
    pb[idx] = lo + hi;
  }
}

// The same pass for rows that are multiples of 16 bytes (W % 4 == 0; every shape of the network), built for the memory system:
//  * a tile is 256 consecutive positions of (TR = 256 / TC) rows x TC columns, TC = 128 / 64 / 32 by the width: with TC == W the 1 KB of a
//    channel is ONE contiguous run, read by one wave instruction as 64 x 16 bytes (the 8 x 32 tile above: eight 128-byte pieces 4 bytes
//    per lane, and every piece of every channel on the same DRAM bank: the channel planes are 3 x 2^21 bytes apart);
//  * the next tile travels global -> registers (8 x 16 B + 7 x 4 B per thread, ONE round trip) under the 64 MFMAs of the current one;
//  * A fragments are 16-byte LDS reads (channel pitch 260 floats: the 16 lanes of a read group hit 16 different 16-byte slots), one read
//    feeding the four K-steps whose voxel pairs are (8 m + e, 8 m + 4 + e), e = 0..3.
constexpr int XS2 = 260;

template <int TC>
__global__ __launch_bounds__(NT) void classif_bww2_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                          const float* __restrict__ mean, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float* __restrict__ part, int B, int Ci, int D,
                                                          int H, int W, int nHt, int nWt, int T, int S) {
  constexpr int TR = 256 / TC, GWD = TC + 2, GHT = TR + 2, GPL2 = GHT * GWD, NG = (3 * GPL2 + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float xl[32 * XS2];  // 33 280 B; the cross-wave reduction (2 x 4 x 1024 floats) reuses it
  __shared__ float gl[3 * GPL2];
  const int s = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long long HW = (long long)H * W, DHW = (long long)D * HW;
  const int j = lane & 31, khalf = lane >> 5;
  const int tap = j < 27 ? j : 0;
  const int toff = (2 - tap / 9) * GPL2 + (2 - (tap / 3) % 3) * GWD + (2 - tap % 3);
  const float mu = j < Ci ? mean[j] : 0.f, sc = j < Ci ? scale[j] : 0.f, sh = j < Ci ? shift[j] : 0.f;
  const float nanv = __builtin_nanf("");
  f32x16 accq = {0}, accm = {0};

  // staging map: float4 f = tid + 256 i covers channel (tid >> 6) + 4 i, voxels 4 (tid & 63) .. + 3 of the tile
  const int v0 = 4 * (tid & 63), srow = v0 / TC, scol = v0 % TC;
  float4 px[8];
  float pg[NG];
  unsigned pxok = 0, pgok = 0;
  auto request = [&](int tt) {
    int t = tt;
    const int wt = t % nWt;
    t /= nWt;
    const int ht = t % nHt;
    t /= nHt;
    const int d0 = t % D;
    const int b = t / D;
    const int w0 = wt * TC, h0 = ht * TR;
    const float* xb = x + (long long)b * Ci * DHW + d0 * HW;
    const float* gb = gy + (long long)b * DHW;
    // branch-free: validity as 0 / 1 integers (a short-circuit && becomes a branch per load), addresses clamped by multiplication
    const int gh = h0 + srow, gw = w0 + scol;
    const unsigned vok = (unsigned)(gh < H) & (unsigned)(gw < W);
    const long long voff = (long long)vok * ((long long)gh * W + gw);
    pxok = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = (tid >> 6) + 4 * i;
      px[i] = *reinterpret_cast<const float4*>(xb + (long long)min(c, Ci - 1) * DHW + voff);
      pxok |= (vok & (unsigned)(c < Ci)) << i;
    }
    pgok = 0;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const int idx = min(tid + NT * i, 3 * GPL2 - 1);
      const int zz = idx / GPL2, rem = idx - zz * GPL2;
      const int yy = rem / GWD, xx = rem - yy * GWD;
      const int gd = d0 - 1 + zz, gh2 = h0 - 1 + yy, gw2 = w0 - 1 + xx;
      const unsigned ok = (unsigned)(gd >= 0) & (unsigned)(gd < D) & (unsigned)(gh2 >= 0) & (unsigned)(gh2 < H) & (unsigned)(gw2 >= 0) &
                          (unsigned)(gw2 < W);
      pg[i] = gb[(long long)ok * (gd * HW + (long long)gh2 * W + gw2)];
      pgok |= ok << i;
    }
  };
  if (s < T) request(s);
  // the three per-channel constants are used inside the loop: have them arrive HERE -- a wait placed at their first use in the loop body
  // would be a static vmcnt(0) in every iteration (the counter is in-order) and drain the tile prefetch in front of the MFMAs
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int tt = s; tt < T; tt += S) {
    __syncthreads();  // every wave is done with the fragments of the tile before
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = (tid >> 6) + 4 * i;
      const bool ok = (pxok >> i) & 1;  // voxels outside the volume (and channels beyond Ci): NaN, their mask is false
      *reinterpret_cast<float4*>(xl + c * XS2 + v0) = ok ? px[i] : make_float4(nanv, nanv, nanv, nanv);
    }
#pragma unroll
    for (int i = 0; i < NG; ++i)
      if (tid + NT * i < 3 * GPL2) gl[tid + NT * i] = ((pgok >> i) & 1) ? pg[i] : 0.f;
    __syncthreads();
    request(tt + S < T ? tt + S : tt);  // travels under the MFMAs below; UNCONDITIONAL (the last tile re-requests itself: cache hits,
                                         // unused) -- a conditional request would make every later wait in this loop a vmcnt(0)
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the 15 requests behind the 64 MFMAs)
    const float* ap = xl + j * XS2 + wave * 64 + 4 * khalf;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 a4 = *reinterpret_cast<const float4*>(ap + 8 * m);
      const int v = wave * 64 + 8 * m;  // (compile-time per wave up to `wave`: row / column of the 8-voxel block)
      const float* bp = gl + toff + (v / TC) * GWD + (v % TC) + 4 * khalf;
      const float ye[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gv = bp[e];
        const bool mk = __builtin_fmaf(ye[e], sc, sh) > 0.f;  // false for the NaN of a voxel outside the volume
        accq = mfma32(mk ? ye[e] - mu : 0.f, gv, accq);
        accm = mfma32(mk ? 1.f : 0.f, gv, accm);
      }
    }
  }
  __syncthreads();
  float* red = xl;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = (q & 3) + 8 * (q >> 2) + 4 * khalf;
    red[wave * 1024 + i * 32 + j] = accq[q];
    red[4096 + wave * 1024 + i * 32 + j] = accm[q];
  }
  __syncthreads();
  float* pb = part + (long long)s * 2048;
  for (int idx = tid; idx < 2048; idx += NT) {
    const float* r = red + (idx >> 10) * 4096 + (idx & 1023);
    float lo = r[0] + r[1024], hi = r[2048] + r[3072];
    asm volatile("" : "+v"(lo), "+v"(hi));  // (keeps the unrolled iterations from being packed into v_pk_add_f32)
    pb[idx] = lo + hi;
  }
}

// One block per channel: sums the S partial pairs of its 27 taps (one wave per tap at a time, lanes over the slices, fixed butterfly), then
//   gw[c][t] (+)= scale[c] Q[c][t] + beta[c] M[c][t];   sg = sum_t w[c][t] M[c][t];   sgy = sum_t w[c][t] Q[c][t]  (= sum g (y - mean))
//   ggamma[c] (+)= invstd sgy;  gbeta[c] (+)= sg;  coef[c] = (A, Bc, Cc, 0) of bn_bwd_apply_kernel (bn_act.hip) for the apply pass.
__global__ __launch_bounds__(NT) void classif_bwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ w,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ scale, float* __restrict__ gw,
                                                                float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                                float4* __restrict__ coef, int S, double count, int accumulate) {
  __shared__ double qs[27], ms[27];
  const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = wave; t < 27; t += NT / 64) {
    const float* p = part + c * 32 + t;
    double q = 0.0, m = 0.0;
    for (int s = lane; s < S; s += 64) {
      q += (double)p[(long long)s * 2048];
      m += (double)p[(long long)s * 2048 + 1024];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      q += __shfl_xor(q, off, 64);
      m += __shfl_xor(m, off, 64);
    }
    if (lane == 0) {
      qs[t] = q;
      ms[t] = m;
    }
  }
  __syncthreads();
  if (threadIdx.x < 27) {
    const int t = threadIdx.x;
    const float v = (float)((double)scale[c] * qs[t] + (double)beta[c] * ms[t]);
    gw[c * 27 + t] = accumulate ? gw[c * 27 + t] + v : v;
  }
  if (threadIdx.x == 0) {
    double sgy = 0.0, sg = 0.0;
    for (int t = 0; t < 27; ++t) {  // fixed order
      sgy += (double)w[c * 27 + t] * qs[t];
      sg += (double)w[c * 27 + t] * ms[t];
    }
    const double is = invstd[c], mu = mean[c], gm = gamma[c];
    const double dgamma = is * sgy;  // sgy is already taken about the mean
    const double A = gm * is;
    coef[c] = make_float4((float)A, (float)(-A * is * dgamma / count), (float)(A * (mu * is * dgamma - sg) / count), 0.f);
    if (accumulate) {
      ggamma[c] += (float)dgamma;
      gbeta[c] += (float)sg;
    } else {
      ggamma[c] = (float)dgamma;
      gbeta[c] = (float)sg;
    }
  }
}

// ------------------------------------------------------------------------------------------------------- backward, pass 2
// conv3d_co1_bwd_data_kernel (conv3d_c1.hip: D[i = c][j = 32 voxels along w], K = the 27 taps, B = the g1 halo tile at the per-lane offset
// of tap k) with the BatchNorm backward's apply pass in its store: dL/dy = A [fma(y, sc, sh) > 0] g + Bc y + Cc.
constexpr int BTD = 2, BTH = 8;
constexpr int BID = BTD + 2, BIH = BTH + 2, BIW = 34;

__global__ __launch_bounds__(NT) void classif_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                               const float* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float4* __restrict__ coef,
                                                               float* __restrict__ gx, int B, int Ci, int D, int H, int W, int nDt,
                                                               int nHt, int nWt) {
  __shared__ float tile[BID * BIH * BIW];
  __shared__ __attribute__((aligned(16))) float ctab[32 * 8];  // per channel: sc, sh, A, Bc | Cc
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int wt = t % nWt;
  t /= nWt;
  const int ht = t % nHt;
  t /= nHt;
  const int dt = t % nDt;
  const int b = t / nDt;
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
  if (tid < 32) {
    const int c = tid < Ci ? tid : Ci - 1;
    const float4 k = coef[c];
    ctab[tid * 8 + 0] = scale[c];
    ctab[tid * 8 + 1] = shift[c];
    ctab[tid * 8 + 2] = k.x;
    ctab[tid * 8 + 3] = k.y;
    ctab[tid * 8 + 4] = k.z;
  }
  float a[14];
  int toff[14];
  const int c = lane & 31;
#pragma unroll
  for (int ks = 0; ks < 14; ++ks) {
    const int k = 2 * ks + (lane >> 5);
    a[ks] = (k < 27 && c < Ci) ? w[c * 27 + 26 - k] : 0.f;
    const int kk = k < 27 ? k : 26;
    toff[ks] = (kk / 9) * (BIH * BIW) + ((kk / 3) % 3) * BIW + kk % 3;
  }
  // this lane's 4 rows: validity and the (clamped) voxel offset inside a channel
  const int gw = w0 + (lane & 31);
  long long sp[4];
  bool rok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = wave * 4 + r;
    const int gd = d0 + row / BTH, gh = h0 + row % BTH;
    rok[r] = gd < D && gh < H && gw < W;
    sp[r] = rok[r] ? gd * HW + (long long)gh * W + gw : 0;
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
      const int row = wave * 4 + r;
      acc[r] = mfma32(a[ks], tp[(row / BTH) * (BIH * BIW) + (row % BTH) * BIW + toff[ks]], acc[r]);
    }

  const float* yb = y + (long long)b * Ci * DHW;
  float* gxb = gx + (long long)b * Ci * DHW;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
    const int ic = i < Ci ? i : Ci - 1;
    const float4 k0 = *reinterpret_cast<const float4*>(ctab + ic * 8);
    const float Cc = ctab[ic * 8 + 4];
    float yv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) yv[r] = yb[ic * DHW + sp[r]];
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // (scalar chains with opaque intermediates: no packed fp32 on freshly loaded values, see the 16-byte kernel)
      float yr = yv[r];
      asm volatile("" : "+v"(yr));
      float t = __builtin_fmaf(yr, k0.x, k0.y);
      asm volatile("" : "+v"(t));
      const float g = t > 0.f ? acc[r][q] : 0.f;
      float v = __builtin_fmaf(k0.w, yr, Cc);
      asm volatile("" : "+v"(v));
      float res = __builtin_fmaf(k0.z, g, v);
      asm volatile("" : "+v"(res));
      if (rok[r] && i < Ci) gxb[i * DHW + sp[r]] = res;
    }
  }
}

// The same pass for rows that are multiples of 16 bytes (W % 4 == 0): a wave owns 128 consecutive positions of (128 / TC) rows x TC
// columns and the N index of its four accumulators is interleaved, accumulator r = positions 4 j + r: a lane then holds FOUR CONSECUTIVE
// voxels of each of its 16 channels, i.e. one 16-byte load of y and one 16-byte store of dL/dy per channel (64 + 64 dword accesses in the
// kernel above), a wave instruction covers 2 channels x 512 contiguous bytes, and all 16 loads of a lane are requested before the MFMAs.
// AMAX: also leaves the largest finite |dL/dy| it writes (bn_internal.h): the scale of the fp16 convolution gradients that read it.
template <int TC, bool AMAX = false>
__global__ __launch_bounds__(NT) void classif_bwd_apply2_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                const float* __restrict__ y, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float4* __restrict__ coef,
                                                                float* __restrict__ gx, int B, int Ci, int D, int H, int W, int nHt,
                                                                int nWt, unsigned* __restrict__ amax) {
  constexpr int WR = 128 / TC, TRW = 4 * WR, GWD = TC + 2, GHT = TRW + 2, GPL2 = GHT * GWD;
  __shared__ float tile[3 * GPL2];
  __shared__ __attribute__((aligned(16))) float ctab[32 * 8];  // per channel: sc, sh, A, Bc | Cc
  int t = xcd_remap(blockIdx.x, gridDim.x);
  const int wt = t % nWt;
  t /= nWt;
  const int ht = t % nHt;
  t /= nHt;
  const int d0 = t % D;
  const int b = t / D;
  const int w0 = wt * TC, h0 = ht * TRW;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, half = lane >> 5;
  const long long HW = (long long)H * W, DHW = (long long)D * HW;
  const float* gb = gy + (long long)b * DHW;
  // this lane's four voxels: 4 j .. 4 j + 3 of the wave's 128 positions
  const int n0 = 4 * j, lrow = wave * WR + n0 / TC, lcol = n0 % TC;
  const int gh = h0 + lrow, gw = w0 + lcol;
  const bool vok = gh < H && gw < W;
  const unsigned sp = vok ? (unsigned)(d0 * HW + (long long)gh * W + gw) : 0u;  // (C * D * H * W < 2^30: host check)
  const float* yb = y + (long long)b * Ci * DHW;
  float* gxb = gx + (long long)b * Ci * DHW;
  // Request order matters: the vector-memory counter is in-order, so everything the MFMAs need (weights, the g1 tile, the coefficient
  // table) is requested FIRST and the 16 x 16 bytes of y LAST -- the wait in front of the first MFMA then leaves the y loads in flight.
  float a[14];
  int toff[14];
#pragma unroll
  for (int ks = 0; ks < 14; ++ks) {
    const int k = 2 * ks + half;
    const float wv = w[min(j, Ci - 1) * 27 + 26 - min(k, 26)];
    a[ks] = ((unsigned)(k < 27) & (unsigned)(j < Ci)) ? wv : 0.f;
    const int kk = k < 27 ? k : 26;
    toff[ks] = (kk / 9) * GPL2 + ((kk / 3) % 3) * GWD + kk % 3;
  }
  for (int idx = tid; idx < 3 * GPL2; idx += NT) {
    const int dz = idx / GPL2, rem = idx - dz * GPL2;
    const int hy = rem / GWD, wx = rem - hy * GWD;
    const int gd = d0 + dz - 1, gh2 = h0 + hy - 1, gw2 = w0 + wx - 1;
    const unsigned ok = (unsigned)(gd >= 0) & (unsigned)(gd < D) & (unsigned)(gh2 >= 0) & (unsigned)(gh2 < H) & (unsigned)(gw2 >= 0) &
                        (unsigned)(gw2 < W);
    const float v = gb[(long long)ok * (gd * HW + (long long)gh2 * W + gw2)];
    tile[idx] = ok ? v : 0.f;
  }
  if (tid < 32) {
    const int c = tid < Ci ? tid : Ci - 1;
    const float4 k = coef[c];
    ctab[tid * 8 + 0] = scale[c];
    ctab[tid * 8 + 1] = shift[c];
    ctab[tid * 8 + 2] = k.x;
    ctab[tid * 8 + 3] = k.y;
    ctab[tid * 8 + 4] = k.z;
  }
  __builtin_amdgcn_sched_barrier(0);
  float4 yv[16];
  {
    unsigned dhw = (unsigned)DHW;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = (q & 3) + 8 * (q >> 2) + 4 * half;
      yv[q] = *reinterpret_cast<const float4*>(yb + (unsigned)min(i, Ci - 1) * dhw + sp);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  unsigned mx = 0;
  f32x16 acc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = (f32x16){0};
  const float* tp = tile + lrow * GWD + lcol;
#pragma unroll
  for (int ks = 0; ks < 14; ++ks)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = mfma32(a[ks], tp[toff[ks] + r], acc[r]);

#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = (q & 3) + 8 * (q >> 2) + 4 * half;
    const int ic = i < Ci ? i : Ci - 1;
    const float4 k0 = *reinterpret_cast<const float4*>(ctab + ic * 8);
    const float Cc = ctab[ic * 8 + 4];
    const float ye[4] = {yv[q].x, yv[q].y, yv[q].z, yv[q].w};
    // Scalar chains, every intermediate opaque: left alone the SLP vectoriser packs neighbouring elements into v_pk_fma_f32 with op_sel
    // on the register pairs a 16-byte load has just written.  Packed fp32 does not overlap with the other waves' MFMAs on this SIMD
    //  -- and on freshly loaded pairs it is the instruction shape behind round 4's non-repeatable adjoint kernel:
    // with it this kernel gave different bits in 1 of 11 steps when a second process shared the GPU (tests/test_gpu_repeat.py).
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float yr = ye[r];
      asm volatile("" : "+v"(yr));
      float t = __builtin_fmaf(yr, k0.x, k0.y);
      asm volatile("" : "+v"(t));
      const float g = t > 0.f ? acc[r][q] : 0.f;
      float v = __builtin_fmaf(k0.w, yr, Cc);
      asm volatile("" : "+v"(v));
      float res = __builtin_fmaf(k0.z, g, v);
      asm volatile("" : "+v"(res));
      o[r] = res;
    }
    if (vok && i < Ci) {
      *reinterpret_cast<float4*>(gxb + (unsigned)i * (unsigned)DHW + sp) = make_float4(o[0], o[1], o[2], o[3]);
      if (AMAX) mx = max(max(mx, mode::absmax_mag(o[0])), max(max(mode::absmax_mag(o[1]), mode::absmax_mag(o[2])), mode::absmax_mag(o[3])));
    }
  }
  if (AMAX) {
    __syncthreads();  // (the tile is read for the last time by the MFMAs above)
    mode::absmax_block_commit(mx, amax, reinterpret_cast<unsigned*>(tile));
  }
}

int tile_cols(int W) { return W > 64 ? 128 : (W > 32 ? 64 : 32); }  // columns of a 256-position tile of the 16-byte kernels
int classif_tiles(int B, int D, int H, int W) {
  const int TC = tile_cols(W);
  return B * D * mode::cdiv(H, 256 / TC) * mode::cdiv(W, TC);
}

int classif_splits(int T, int per_cu = 4) {
  int S = per_cu * kNumCU;
  if (S > T) S = T;
  return S < 1 ? 1 : S;
}

}  // namespace

extern "C" size_t mode_classif_workspace_bytes(int B, int C, int D, int H, int W) {
  if (B <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  const int T = std::max(B * D * mode::cdiv(H, GTH) * mode::cdiv(W, 32), classif_tiles(B, D, H, W));
  const size_t bww = (size_t)classif_splits(T) * 2048 * sizeof(float) + 32 * sizeof(float4);
  const size_t bn = mode_bn_workspace_bytes(C);
  return bww > bn ? bww : bn;
}

static int check_classif(const char* who, int B, int C, int D, int H, int W) {
  MODE_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE(C <= 32, MODE_ERR_UNSUPPORTED, "%s: %d channels (the fused head takes at most 32: classifN of the reference has 32)", who, C);
  MODE_REQUIRE((long long)C * D * H * W < (1ll << 30), MODE_ERR_UNSUPPORTED, "%s: sample of %lld elements exceeds the 32-bit offsets of the kernels", who,
               (long long)C * D * H * W);
  return MODE_OK;
}

extern "C" int mode_classif_train_fwd(const float* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                      long long* num_batches_tracked, float momentum, float eps, const float* w, const float* add,
                                      float* cost, float* save_mean, float* save_invstd, float* save_scale, float* save_shift,
                                      float* workspace, int B, int C, int D, int H, int W, mode_stream_t stream) {
  const char* who = "mode_classif_train_fwd";
  int rc = check_classif(who, B, C, D, H, W);
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(y && w && cost, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  hipStream_t st = mode::as_stream(stream);
  rc = mode::bn_train_coefficients(y, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, save_mean, save_invstd,
                                   save_scale, save_shift, workspace, B, C, (long long)D * H * W, st, who);
  if (rc != MODE_OK) return rc;
  const int nDc = mode::cdiv(D, ZDC), nHt = mode::cdiv(H, ZTH), nWt = mode::cdiv(W, 32);
  const size_t lds = (size_t)27 * ZPL * sizeof(float) + 32 * sizeof(float2);
  auto kern = C == 32 ? classif_fwd_kernel<true, true> : classif_fwd_kernel<true, false>;
  rc = mode::allow_lds(kern, lds, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(kern, dim3(B * nDc * nHt * nWt), dim3(NT), lds, st, y, w, save_scale, save_shift, add, cost, B, C, D, H, W, nDc, nHt, nWt);
  return mode::check_launch(who);
}

// conv3d_c1.hip's forward for Ci <= 32 runs on the kernel above without the BatchNorm prologue (same tiles, the static request sequence)
int mode::conv3d_co1_fwd_small(const float* x, const float* w, float* y, int B, int Ci, int D, int H, int W, hipStream_t st, const char* who) {
  const int nDc = mode::cdiv(D, ZDC), nHt = mode::cdiv(H, ZTH), nWt = mode::cdiv(W, 32);
  const size_t lds = (size_t)27 * ZPL * sizeof(float) + 32 * sizeof(float2);
  auto kern = Ci == 32 ? classif_fwd_kernel<false, true> : classif_fwd_kernel<false, false>;
  int rc = mode::allow_lds(kern, lds, who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(kern, dim3(B * nDc * nHt * nWt), dim3(NT), lds, st, x, w, nullptr, nullptr, nullptr, y, B, Ci, D, H, W, nDc, nHt, nWt);
  return mode::check_launch(who);
}

extern "C" int mode_classif_train_bwd(const float* gcost, const float* y, const float* w, const float* gamma, const float* beta,
                                      const float* save_mean, const float* save_invstd, const float* save_scale, const float* save_shift,
                                      float* gy, float* gw, float* ggamma, float* gbeta, int accumulate, float* workspace, int B, int C,
                                      int D, int H, int W, mode_stream_t stream) {
  return mode_classif_train_bwd_amax(gcost, y, w, gamma, beta, save_mean, save_invstd, save_scale, save_shift, gy, gw, ggamma, gbeta,
                                     accumulate, workspace, B, C, D, H, W, nullptr, stream);
}

// ... and leaves the largest finite |gy| in gy_absmax (MODE_BN_ABSMAX_FLOATS floats, may be NULL): this IS a BatchNorm backward, and the
// 32 -> 32 convolution in front reads gy in both of its gradients (mode_bn_train_bwd_amax has the contract)
extern "C" int mode_classif_train_bwd_amax(const float* gcost, const float* y, const float* w, const float* gamma, const float* beta,
                                           const float* save_mean, const float* save_invstd, const float* save_scale,
                                           const float* save_shift, float* gy, float* gw, float* ggamma, float* gbeta, int accumulate,
                                           float* workspace, int B, int C, int D, int H, int W, float* gy_absmax, mode_stream_t stream) {
  const char* who = "mode_classif_train_bwd";
  float* amax = gy_absmax;
  int rc = check_classif(who, B, C, D, H, W);
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(gcost && y && w && gamma && beta && save_mean && save_invstd && save_scale && save_shift && gy && gw && ggamma && gbeta && workspace,
               MODE_ERR_BAD_ARG, "%s: null pointer", who);
  MODE_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0, MODE_ERR_UNSUPPORTED, "%s: unaligned workspace", who);
  hipStream_t st = mode::as_stream(stream);
  // rows that are multiples of 16 bytes (and 16-byte aligned tensors): the kernels built on 16-byte accesses; anything else: the dword forms
  const bool fast = W % 4 == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0;
  if (amax && fast) {
    rc = mode::absmax_begin(amax, st, who);
    if (rc != MODE_OK) return rc;
  }
  const int nHt = mode::cdiv(H, GTH), nWt = mode::cdiv(W, 32);
  const int T = fast ? classif_tiles(B, D, H, W) : B * D * nHt * nWt;
  const int S = classif_splits(T, fast ? 3 : 4);  // (classif_bww2_kernel: 144 registers, three workgroups per CU)
  float* part = workspace;
  float4* coef = reinterpret_cast<float4*>(workspace + (size_t)S * 2048);
  if (fast) {
    const int TC = tile_cols(W), TR = 256 / TC;
    const int nHt2 = mode::cdiv(H, TR), nWt2 = mode::cdiv(W, TC);
    if (TC == 128)
      hipLaunchKernelGGL(classif_bww2_kernel<128>, dim3(S), dim3(NT), 0, st, gcost, y, save_mean, save_scale, save_shift, part, B, C, D, H, W, nHt2, nWt2, T, S);
    else if (TC == 64)
      hipLaunchKernelGGL(classif_bww2_kernel<64>, dim3(S), dim3(NT), 0, st, gcost, y, save_mean, save_scale, save_shift, part, B, C, D, H, W, nHt2, nWt2, T, S);
    else
      hipLaunchKernelGGL(classif_bww2_kernel<32>, dim3(S), dim3(NT), 0, st, gcost, y, save_mean, save_scale, save_shift, part, B, C, D, H, W, nHt2, nWt2, T, S);
  } else {
    hipLaunchKernelGGL(classif_bww_kernel, dim3(S), dim3(NT), 0, st, gcost, y, save_mean, save_scale, save_shift, part, B, C, D, H, W, nHt, nWt, T, S);
  }
  rc = mode::check_launch(who);
  if (rc != MODE_OK) return rc;
  hipLaunchKernelGGL(classif_bwd_reduce_kernel, dim3(C), dim3(NT), 0, st, part, w, gamma, beta, save_mean, save_invstd, save_scale, gw, ggamma,
                     gbeta, coef, S, (double)B * D * H * W, accumulate);
  rc = mode::check_launch(who);
  if (rc != MODE_OK) return rc;
  if (fast) {
    const int TC = tile_cols(W), TRW = 4 * (128 / TC);
    const int nHa = mode::cdiv(H, TRW), nWa = mode::cdiv(W, TC);
    const dim3 grid(B * D * nHa * nWa);
    unsigned* am = reinterpret_cast<unsigned*>(amax);
#define MODE_CLASSIF_APPLY2(TCV)                                                                                                          \
  if (amax)                                                                                                                                \
    hipLaunchKernelGGL((classif_bwd_apply2_kernel<TCV, true>), grid, dim3(NT), 0, st, gcost, w, y, save_scale, save_shift, coef, gy, B, C, D, H, \
                       W, nHa, nWa, am);                                                                                                   \
  else                                                                                                                                     \
    hipLaunchKernelGGL((classif_bwd_apply2_kernel<TCV, false>), grid, dim3(NT), 0, st, gcost, w, y, save_scale, save_shift, coef, gy, B, C, D, H, \
                       W, nHa, nWa, am);
    if (TC == 128) {
      MODE_CLASSIF_APPLY2(128)
    } else if (TC == 64) {
      MODE_CLASSIF_APPLY2(64)
    } else {
      MODE_CLASSIF_APPLY2(32)
    }
#undef MODE_CLASSIF_APPLY2
    return mode::check_launch(who);
  }
  const int nDt = mode::cdiv(D, BTD), nHt2 = mode::cdiv(H, BTH);
  hipLaunchKernelGGL(classif_bwd_apply_kernel, dim3(B * nDt * nHt2 * nWt), dim3(NT), 0, st, gcost, w, y, save_scale, save_shift, coef, gy, B, C, D,
                     H, W, nDt, nHt2, nWt);
  rc = mode::check_launch(who);
  // (rows that are not multiples of 16 bytes: the maximum that was asked for comes from a pass of its own)
  return (rc != MODE_OK || !amax) ? rc : mode::abs_max(gy, (long long)B * C * D * H * W, amax, st, who);
}
