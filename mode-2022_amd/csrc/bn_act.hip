// BatchNorm (+ residual add) (+ ReLU) for (B, C, S) fp32 tensors (S = D*H*W or H*W), gfx950.
//
// Reference: nn.BatchNorm3d / nn.BatchNorm2d inside convbn_3d / convbn (models/submodule.py:15-22, torch defaults eps 1e-5,
// momentum 0.1, affine, running statistics, per-replica batch statistics) followed by the residual adds and ReLUs of
// hourglass.forward / ModeDisparity.forward (models/mode_disparity.py:27-46, 115-129).  The reference runs these as separate
// kernels (BN: read, read, write; add: read, read, write; ReLU: read, write); here a layer is two launches / two HBM passes
// in training (statistics, then normalise + add + ReLU with the per-channel finalisation folded into the prologue of every
// block) and one launch / one pass in eval mode.  Pure HBM roofline kernels:
//   train fwd  : (2 reads [+1 read of the skip tensor] + 1 write) * 4 B per element
//   train bwd  : reduce pass (reads gout, y [, out]) + apply pass (reads gout, y [, out], writes gy [, gadd])
// Statistics are reduced per thread / per block in fp32 over short runs and combined across blocks in fp64.
#include "bn_internal.h"
#include <cstdlib>

namespace {

constexpr int NT = 256;       // threads per block (512 measured in the step, round 4: 0.241 / 0.361 against 0.236 / 0.357 ms for the big 3-D layer)
constexpr int NW = NT / 64;  // waves per block

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-wide sum of two values; result valid in thread 0
__device__ __forceinline__ void block_sum2(float& a, float& b, float* sh /* [2 * NW] */) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sh[wave] = a;
    sh[NW + wave] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = sh[0];
    b = sh[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) {  // fixed order
      a += sh[w];
      b += sh[NW + w];
    }
  }
}

// Statistics can be taken over GROUPS of samples: group g = samples [g*B/G, (g+1)*B/G) (G = 1: the whole batch).  The paired
// feature extractor pushes the left and the right images through the network as one batch while keeping the reference's
// per-call BatchNorm statistics: two groups.
// grid = (nsplit, C, G).  partial[((g*C + c)*nsplit + split)*2 + {0,1}] = sum(y), sum(y*y) over this block's share of (b, s).
__global__ __launch_bounds__(NT) void bn_stats_kernel(const float* __restrict__ y, float* __restrict__ partial, int B, int C,
                                                      long long S, int nsplit, unsigned* __restrict__ zero) {
  __shared__ float sh[2 * NW];
  const int c = blockIdx.y, split = blockIdx.x;
  const int Bg = B / gridDim.z, b0 = blockIdx.z * Bg;
  const long long prow = (long long)blockIdx.z * C + c;
  // (the buffer of the apply pass's tracked maximum, if there is one, is zeroed here: a launch of its own costs 5 us in front of every pass)
  if (zero && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < MODE_BN_ABSMAX_FLOATS; i += NT) zero[i] = 0u;
  const long long S4 = (S & 3) ? 0 : (S >> 2);  // rows that are not multiples of 16 bytes: the scalar loop below takes the whole row
  // Shifted sums: everything is accumulated relative to a pivot -- the first element of the group's first sample of this channel
  // (bn_pivot, re-read by the finalisation) -- so that var = E[(y-K)^2] - E[y-K]^2 does not cancel when |mean| >> std
  // (torch / MIOpen use Welford; with K inside the data range the shifted form is as accurate and stays a single pass).
  const float K = y[((long long)b0 * C + c) * S];
  float s0 = 0.f, s1 = 0.f;
  for (int b = b0; b < b0 + Bg; ++b) {
    const float4* p = reinterpret_cast<const float4*>(y + ((long long)b * C + c) * S);
    // four independent loads per iteration (see bn_apply_kernel); the sums are taken in the same order as element by element
    auto acc1 = [&](const float4& v) {
      const float dx = v.x - K, dy = v.y - K, dz = v.z - K, dw = v.w - K;
      s0 += (dx + dy) + (dz + dw);
      s1 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    };
    // block `split` takes the 4 KB pieces split, split + nsplit, ... of the row (not one contiguous slice): concurrently running
    // blocks then read neighbouring addresses instead of addresses a multiple of 96 KB apart (same access pattern as the apply
    // kernels; measured neutral on the 403 MB tensors)
    const long long st = (long long)nsplit * NT;
    long long i = (long long)split * NT + threadIdx.x;
    for (; i + 3 * st < S4; i += 4 * st) {
      const float4 v0 = p[i], v1 = p[i + st], v2 = p[i + 2 * st], v3 = p[i + 3 * st];
      acc1(v0);
      acc1(v1);
      acc1(v2);
      acc1(v3);
    }
    for (; i < S4; i += st) acc1(p[i]);
    {  // scalar path (S % 4 != 0), spread over the blocks like the vector path; empty otherwise
      const float* q = y + ((long long)b * C + c) * S;
      for (long long i = (S4 << 2) + (long long)split * NT + threadIdx.x; i < S; i += (long long)nsplit * NT) {
        const float dq = q[i] - K;
        s0 += dq;
        s1 += dq * dq;
      }
    }
  }
  block_sum2(s0, s1, sh);
  if (threadIdx.x == 0) {
    partial[(prow * nsplit + split) * 2] = s0;
    partial[(prow * nsplit + split) * 2 + 1] = s1;
  }
}

// Block-wide fp64 sum of this channel's per-block partials (pairs); every block of a channel walks them in the same
// order, so all blocks derive bit-identical coefficients.  Result valid in thread 0.
__device__ __forceinline__ void reduce_partials(const float* __restrict__ partial, int c, int nsplit, double& s0, double& s1,
                                                double* shd /* [2 * NW] */) {
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nsplit; i += NT) {
    const float2 v = reinterpret_cast<const float2*>(partial)[(long long)c * nsplit + i];
    a += (double)v.x;
    b += (double)v.y;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    shd[wave] = a;
    shd[NW + wave] = b;
  }
  __syncthreads();
  s0 = shd[0];
  s1 = shd[NW];
#pragma unroll
  for (int w = 1; w < NW; ++w) {  // fixed order, the same in every block
    s0 += shd[w];
    s1 += shd[NW + w];
  }
  __syncthreads();  // shd may be reused by the next call
}

// Where the affine coefficients of the apply pass come from.
//   TRAIN: batch statistics from the per-block partial sums of bn_stats_kernel; the block (chunk 0, sample 0) of each channel
//          also stores mean / invstd for the backward pass and updates the running statistics (unbiased variance), and the
//          one of channel 0 counts the batch in num_batches_tracked.
//   EVAL : running statistics, same arithmetic order as torch's eval-mode batch_norm.
struct BnCoefArgs {
  const float* partial;
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  long long* num_batches_tracked;
  float* save_mean;
  float* save_invstd;
  float* save_scale;  // the float32 affine coefficients of the apply pass (optional): the backward pass re-derives the ReLU
  float* save_shift;  // mask from y with exactly these instead of reading `out`
  float momentum, eps;
  int nsplit;
  double count;  // elements per (group, channel)
  int groups;    // statistics groups (TRAIN)
  int Bg;        // samples per group
  const float* pivot;  // (TRAIN) per (group, channel) pivot of the shifted sums; null: the first element of the group's first sample
  unsigned* amax;      // optional: atomicMax of the bit patterns of the finite |out| values (the scale source of the fp16 convolutions)
};

// grid = (chunks, B*C): out = y*scale[c] + shift[c] (+ add) (relu)
// AMAX: also leaves the largest finite |out| in *k.amax (its own instantiation: the untracked passes keep their registers and occupancy)
template <bool RELU, bool ADD, bool TRAIN, bool AMAX = false>
__global__ __launch_bounds__(NT) void bn_apply_kernel(const float* __restrict__ y, const float* __restrict__ add, BnCoefArgs k,
                                                      float* __restrict__ out, int C, long long S) {
  __shared__ double shd[2 * NW];
  __shared__ float coef[2];
  const int bc = blockIdx.y;
  const int c = bc % C;
  if (TRAIN) {
    const int b = bc / C, g = b / k.Bg;
    double s0, s1;
    reduce_partials(k.partial, g * C + c, k.nsplit, s0, s1, shd);
    if (threadIdx.x == 0) {
      const double dm = s0 / k.count;  // mean of y - K, K = the pivot of bn_stats_kernel
      const double mean = (k.pivot ? (double)k.pivot[g * C + c] : (double)y[((long long)(g * k.Bg) * C + c) * S]) + dm;
      double var = s1 / k.count - dm * dm;
      if (var < 0.0) var = 0.0;
      const double invstd = 1.0 / sqrt(var + (double)k.eps);
      const double sc = (double)k.gamma[c] * invstd;
      coef[0] = (float)sc;
      coef[1] = (float)((double)k.beta[c] - mean * sc);
      if (blockIdx.x == 0 && b == g * k.Bg) {  // first sample of the group: its saved statistics
        k.save_mean[g * C + c] = (float)mean;
        k.save_invstd[g * C + c] = (float)invstd;
        if (k.save_scale) {
          k.save_scale[g * C + c] = coef[0];
          k.save_shift[g * C + c] = coef[1];
        }
      }
    }
    // running statistics: one block per channel applies the groups' updates IN ORDER (what consecutive calls of the module
    // would do), each rounded to float like a separate call
    if (blockIdx.x == 0 && bc < C && k.running_mean) {
      for (int gg = 0; gg < k.groups; ++gg) {
        double t0, t1;
        reduce_partials(k.partial, gg * C + c, k.nsplit, t0, t1, shd);
        if (threadIdx.x == 0) {
          const double dm = t0 / k.count;
          const double mean = (k.pivot ? (double)k.pivot[gg * C + c] : (double)y[((long long)(gg * k.Bg) * C + c) * S]) + dm;
          double var = t1 / k.count - dm * dm;
          if (var < 0.0) var = 0.0;
          const double unbiased = k.count > 1.0 ? var * k.count / (k.count - 1.0) : var;
          k.running_mean[c] = (float)((1.0 - k.momentum) * (double)k.running_mean[c] + k.momentum * mean);
          k.running_var[c] = (float)((1.0 - k.momentum) * (double)k.running_var[c] + k.momentum * unbiased);
        }
      }
    }
    if (blockIdx.x == 0 && bc == 0 && threadIdx.x == 0 && k.num_batches_tracked) *k.num_batches_tracked += k.groups;
  } else if (threadIdx.x == 0) {
    const float sc = k.gamma[c] / sqrtf(k.running_var[c] + k.eps);
    coef[0] = sc;
    coef[1] = k.beta[c] - k.running_mean[c] * sc;
  }
  __syncthreads();
  const float sc = coef[0], sh = coef[1];
  const long long base = (long long)bc * S;
  const long long S4 = (S & 3) ? 0 : (S >> 2);  // rows that are not multiples of 16 bytes: the scalar loop below takes the whole row
  const float4* yp = reinterpret_cast<const float4*>(y + base);
  const float4* ap = reinterpret_cast<const float4*>(add + base);
  float4* op = reinterpret_cast<float4*>(out + base);
  // four independent 16-byte loads per thread and iteration: the compiler emits the plain loop as load -> wait -> store, i.e.
  // ONE load in flight per thread (32 KB per CU at full occupancy: Little's law caps that at ~5.3 TB/s on this chip)
  auto one = [&](float4 v, const float4& a) {
    v.x = __builtin_fmaf(v.x, sc, sh);  // one rounding: the backward pass repeats exactly this to rebuild the ReLU mask
    v.y = __builtin_fmaf(v.y, sc, sh);
    v.z = __builtin_fmaf(v.z, sc, sh);
    v.w = __builtin_fmaf(v.w, sc, sh);
    if (ADD) {
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (RELU) {
      v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w);
    }
    return v;
  };
  // (k.amax) the largest finite magnitude this thread writes: the consumer's fp16 scale comes out of this pass instead of one of its own
  unsigned mx = 0;
  constexpr bool track = AMAX;
  auto mag = [](float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f) & 0x7fffffffu;
    return u < 0x7f800000u ? u : 0u;
  };
  auto put = [&](float4* dst, float4 v) {
    if (track) mx = max(max(mx, mag(v.x)), max(max(mag(v.y), mag(v.z)), mag(v.w)));
    *dst = v;
  };
  const long long st = (long long)gridDim.x * NT;
  long long i = (long long)blockIdx.x * NT + threadIdx.x;
  for (; i + 3 * st < S4; i += 4 * st) {
    const float4 v0 = yp[i], v1 = yp[i + st], v2 = yp[i + 2 * st], v3 = yp[i + 3 * st];
    float4 a0 = v0, a1 = v1, a2 = v2, a3 = v3;
    if (ADD) {
      a0 = ap[i]; a1 = ap[i + st]; a2 = ap[i + 2 * st]; a3 = ap[i + 3 * st];
    }
    put(op + i, one(v0, a0));
    put(op + i + st, one(v1, a1));
    put(op + i + 2 * st, one(v2, a2));
    put(op + i + 3 * st, one(v3, a3));
  }
  for (; i < S4; i += st) put(op + i, one(yp[i], ADD ? ap[i] : yp[i]));
  for (long long i = (S4 << 2) + (long long)blockIdx.x * NT + threadIdx.x; i < S; i += (long long)gridDim.x * NT) {  // scalar path (S % 4 != 0)
      float v = __builtin_fmaf(y[base + i], sc, sh);
      if (ADD) v += add[base + i];
      if (RELU) v = relu_nan(v);
      if (track) mx = max(mx, mag(v));
      out[base + i] = v;
    }
  if (track) mode::absmax_block_commit(mx, k.amax, reinterpret_cast<unsigned*>(shd));  // (shd: read for the last time in front of the barrier above)
}

// Backward reduce: g = RELU ? (out > 0 ? gout : 0) : gout;  partial = sum(g), sum(g*y).
// RELU = 1: the mask is read from the forward output; RELU = 2 (no residual add): it is rebuilt from y with the forward's own
// float32 coefficients, out = fma(y, scale, shift) -- bit-identical, and one tensor less to read in both backward passes.
template <int RELU>
__global__ __launch_bounds__(NT) void bn_bwd_stats_kernel(const float* __restrict__ gout, const float* __restrict__ y,
                                                          const float* __restrict__ out, const float* __restrict__ mscale,
                                                          const float* __restrict__ mshift, float* __restrict__ partial, int B, int C,
                                                          long long S, int nsplit, unsigned* __restrict__ zero) {
  __shared__ float sh[2 * NW];
  const int c = blockIdx.y, split = blockIdx.x;
  const int Bg = B / gridDim.z, b0 = blockIdx.z * Bg;
  const long long prow = (long long)blockIdx.z * C + c;
  // (the buffer of the apply pass's tracked maximum, if there is one, is zeroed here: a launch of its own costs 5 us in front of every pass)
  if (zero && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < MODE_BN_ABSMAX_FLOATS; i += NT) zero[i] = 0u;
  const float msc = RELU == 2 ? mscale[prow] : 0.f, msh = RELU == 2 ? mshift[prow] : 0.f;
  const long long S4 = (S & 3) ? 0 : (S >> 2);  // rows that are not multiples of 16 bytes: the scalar loop below takes the whole row
  float s0 = 0.f, s1 = 0.f;
  for (int b = b0; b < b0 + Bg; ++b) {
    const long long base = ((long long)b * C + c) * S;
    const float4* gp = reinterpret_cast<const float4*>(gout + base);
    const float4* yp = reinterpret_cast<const float4*>(y + base);
    const float4* op = reinterpret_cast<const float4*>(out + base);
    // (four independent loads of every operand per iteration; sums in element order)
    auto acc1 = [&](float4 g, const float4& v, const float4& o) {
      if (RELU == 1) {
        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
      } else if (RELU == 2) {
        g.x = __builtin_fmaf(v.x, msc, msh) > 0.f ? g.x : 0.f;
        g.y = __builtin_fmaf(v.y, msc, msh) > 0.f ? g.y : 0.f;
        g.z = __builtin_fmaf(v.z, msc, msh) > 0.f ? g.z : 0.f;
        g.w = __builtin_fmaf(v.w, msc, msh) > 0.f ? g.w : 0.f;
      }
      s0 += (g.x + g.y) + (g.z + g.w);
      s1 += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
    };
    const long long st = (long long)nsplit * NT;  // interleaved 4 KB pieces per block, see bn_stats_kernel
    long long i = (long long)split * NT + threadIdx.x;
    for (; i + 3 * st < S4; i += 4 * st) {
      const float4 g0 = gp[i], g1 = gp[i + st], g2 = gp[i + 2 * st], g3 = gp[i + 3 * st];
      const float4 v0 = yp[i], v1 = yp[i + st], v2 = yp[i + 2 * st], v3 = yp[i + 3 * st];
      float4 o0 = v0, o1 = v1, o2 = v2, o3 = v3;
      if (RELU == 1) {
        o0 = op[i]; o1 = op[i + st]; o2 = op[i + 2 * st]; o3 = op[i + 3 * st];
      }
      acc1(g0, v0, o0);
      acc1(g1, v1, o1);
      acc1(g2, v2, o2);
      acc1(g3, v3, o3);
    }
    for (; i < S4; i += st) acc1(gp[i], yp[i], RELU == 1 ? op[i] : yp[i]);
    for (long long i = (S4 << 2) + (long long)split * NT + threadIdx.x; i < S; i += (long long)nsplit * NT) {  // scalar path (S % 4 != 0)
        float g = gout[base + i];
        if (RELU == 1) g = out[base + i] > 0.f ? g : 0.f;
        if (RELU == 2) g = __builtin_fmaf(y[base + i], msc, msh) > 0.f ? g : 0.f;
        s0 += g;
        s1 += g * y[base + i];
      }
  }
  block_sum2(s0, s1, sh);
  if (threadIdx.x == 0) {
    partial[(prow * nsplit + split) * 2] = s0;
    partial[(prow * nsplit + split) * 2 + 1] = s1;
  }
}

// grid = (chunks, B*C): g = masked gout; gy = A*g + Bc*y + Cc; optionally gadd = g, with
//   dgamma = invstd * (sum(g*y) - mean*sum(g)), dbeta = sum(g), A = gamma*invstd, Bc = -A*invstd*dgamma/count,
//   Cc = A*(mean*invstd*dgamma - sum(g))/count
// derived by every block from the partial sums; the block (chunk 0, sample 0) of each channel stores (or, with
// `accumulate`, adds into) ggamma / gbeta.
template <int RELU, bool GADD, bool AMAX = false>
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const float* __restrict__ gout, const float* __restrict__ y,
                                                          const float* __restrict__ out, const float* __restrict__ mscale,
                                                          const float* __restrict__ mshift, const float* __restrict__ partial,
                                                          const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, float* __restrict__ ggamma,
                                                          float* __restrict__ gbeta, int accumulate, int nsplit, double count,
                                                          int groups, int Bg, float* __restrict__ gy, float* __restrict__ gadd, int C,
                                                          long long S, unsigned* __restrict__ amax) {
  __shared__ double shd[2 * NW];
  __shared__ float coef[3];
  unsigned mx = 0;  // (amax) the largest finite |gy| this thread writes: the scale source of the fp16 convolution gradients that read gy
  auto mag = [](float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f) & 0x7fffffffu;
    return u < 0x7f800000u ? u : 0u;
  };
  const int bc = blockIdx.y;
  const int c = bc % C;
  const int grp = (bc / C) / Bg;
  {
    double sg, sgy;
    reduce_partials(partial, grp * C + c, nsplit, sg, sgy, shd);
    if (threadIdx.x == 0) {
      const double mean = save_mean[grp * C + c], invstd = save_invstd[grp * C + c], gm = gamma[c];
      const double dgamma = invstd * (sgy - mean * sg);
      const double A = gm * invstd;
      coef[0] = (float)A;
      coef[1] = (float)(-A * invstd * dgamma / count);
      coef[2] = (float)(A * (mean * invstd * dgamma - sg) / count);
    }
    // affine gradients: one block per channel sums the groups in order
    if (blockIdx.x == 0 && bc < C) {
      double dg = 0.0, db = 0.0;
      for (int gg = 0; gg < groups; ++gg) {
        double tg, tgy;
        reduce_partials(partial, gg * C + c, nsplit, tg, tgy, shd);
        dg += (double)save_invstd[gg * C + c] * (tgy - (double)save_mean[gg * C + c] * tg);
        db += tg;
      }
      if (threadIdx.x == 0) {
        if (accumulate) {
          ggamma[c] += (float)dg;
          gbeta[c] += (float)db;
        } else {
          ggamma[c] = (float)dg;
          gbeta[c] = (float)db;
        }
      }
    }
    __syncthreads();
  }
  const float A = coef[0], Bc = coef[1], Cc = coef[2];
  const float msc = RELU == 2 ? mscale[grp * C + c] : 0.f, msh = RELU == 2 ? mshift[grp * C + c] : 0.f;
  const long long base = (long long)bc * S;
  const long long S4 = (S & 3) ? 0 : (S >> 2);  // rows that are not multiples of 16 bytes: the scalar loop below takes the whole row
  const float4* gp = reinterpret_cast<const float4*>(gout + base);
  const float4* yp = reinterpret_cast<const float4*>(y + base);
  const float4* op = reinterpret_cast<const float4*>(out + base);
  float4* gyp = reinterpret_cast<float4*>(gy + base);
  float4* gap = reinterpret_cast<float4*>(gadd + base);
  // (four independent loads of every operand per thread and iteration, see bn_apply_kernel)
  auto one = [&](long long i, float4 g, const float4& v, const float4& o) {
    if (RELU == 1) {
      g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
    } else if (RELU == 2) {
      g.x = __builtin_fmaf(v.x, msc, msh) > 0.f ? g.x : 0.f;
      g.y = __builtin_fmaf(v.y, msc, msh) > 0.f ? g.y : 0.f;
      g.z = __builtin_fmaf(v.z, msc, msh) > 0.f ? g.z : 0.f;
      g.w = __builtin_fmaf(v.w, msc, msh) > 0.f ? g.w : 0.f;
    }
    if (GADD) gap[i] = g;
    float4 r;
    r.x = A * g.x + Bc * v.x + Cc;
    r.y = A * g.y + Bc * v.y + Cc;
    r.z = A * g.z + Bc * v.z + Cc;
    r.w = A * g.w + Bc * v.w + Cc;
    if (AMAX) mx = max(max(mx, mag(r.x)), max(max(mag(r.y), mag(r.z)), mag(r.w)));
    gyp[i] = r;
  };
  const long long st = (long long)gridDim.x * NT;
  long long i = (long long)blockIdx.x * NT + threadIdx.x;
  for (; i + 3 * st < S4; i += 4 * st) {
    const float4 g0 = gp[i], g1 = gp[i + st], g2 = gp[i + 2 * st], g3 = gp[i + 3 * st];
    const float4 v0 = yp[i], v1 = yp[i + st], v2 = yp[i + 2 * st], v3 = yp[i + 3 * st];
    float4 o0 = v0, o1 = v1, o2 = v2, o3 = v3;
    if (RELU == 1) {
      o0 = op[i]; o1 = op[i + st]; o2 = op[i + 2 * st]; o3 = op[i + 3 * st];
    }
    one(i, g0, v0, o0);
    one(i + st, g1, v1, o1);
    one(i + 2 * st, g2, v2, o2);
    one(i + 3 * st, g3, v3, o3);
  }
  for (; i < S4; i += st) one(i, gp[i], yp[i], RELU == 1 ? op[i] : yp[i]);
  for (long long i = (S4 << 2) + (long long)blockIdx.x * NT + threadIdx.x; i < S; i += (long long)gridDim.x * NT) {  // scalar path (S % 4 != 0)
      float g = gout[base + i];
      if (RELU == 1) g = out[base + i] > 0.f ? g : 0.f;
      if (RELU == 2) g = __builtin_fmaf(y[base + i], msc, msh) > 0.f ? g : 0.f;
      if (GADD) gadd[base + i] = g;
      const float r = A * g + Bc * y[base + i] + Cc;
      if (AMAX) mx = max(mx, mag(r));
      gy[base + i] = r;
    }
  if (AMAX) mode::absmax_block_commit(mx, amax, reinterpret_cast<unsigned*>(shd));
}

int pick_nsplit(int C, long long S) {
  // Blocks of the two statistics passes.  Round 4 (tools/experiments/stream_bw.hip, profiles/r04_microbench_stream_bw.txt): a plain read
  // stream of 256-thread blocks reaches 5.2-5.6 TB/s with 4-16 blocks per CU and 6.2 TB/s with 32-64 -- so up to 64 blocks per CU
  // in total, as long as a thread still has at least 8 x 16 bytes of every image (small layers pay the per-block reduction instead:
  // measured, 16 per CU kept there by this floor)
  long long n = (64LL * kNumCU + C - 1) / C;
  const long long maxn = std::max<long long>(1, (S / 4) / (8 * NT));
  if (n > maxn) n = maxn;
  if (n < 1) n = 1;
  if (n > 1024) n = 1024;
  return (int)n;
}

int apply_chunks(int BC, long long S) {
  // blocks per row of the two apply passes: up to 64 per CU in total (a read + write stream: 4.3 TB/s at 8 blocks of 256 threads per
  // CU, 5.7 at 64), at least 8 x 16 bytes per thread
  long long n = (64LL * kNumCU + BC - 1) / BC;
  const long long maxn = std::max<long long>(1, (S / 4) / (8 * NT));
  if (n > maxn) n = maxn;
  return (int)std::max<long long>(1, n);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// tensors are read in 16-byte pieces when their rows are multiples of 16 bytes, element by element otherwise
bool aligned_rows(const void* p, long long S) { return (reinterpret_cast<uintptr_t>(p) & ((S & 3) ? 3 : 15)) == 0; }

int check_bn(int B, int C, long long S, const char* who) {
  MODE_REQUIRE(B >= 0 && C > 0 && S > 0, MODE_ERR_BAD_ARG, "%s: non-positive size", who);
  MODE_REQUIRE((long long)B * C < 65536, MODE_ERR_UNSUPPORTED, "%s: B*C = %lld exceeds the grid limit", who, (long long)B * C);
  return MODE_OK;
}

template <typename K, typename... Args>
int launch_apply(K kernel, int BC, long long S, hipStream_t st, const char* who, Args... args) {
  hipLaunchKernelGGL(kernel, dim3(apply_chunks(BC, S), BC), dim3(NT), 0, st, args...);
  return mode::check_launch(who);
}

}  // namespace

namespace mode {
int absmax_begin(float* amax, hipStream_t st, const char* who) { return fill_words(amax, 0u, MODE_BN_ABSMAX_FLOATS, st, who); }
}  // namespace mode

// workspace (floats): per-block partial sums, up to 1024 pairs per (group, channel); C = channels x groups
extern "C" size_t mode_bn_workspace_bytes(int C) { return C > 0 ? (size_t)C * 2048 * sizeof(float) : 0; }

// prestats > 0: the workspace already holds `prestats` partial pairs per channel followed by the C pivots (written by a convolution
// kernel's statistics epilogue, mode_conv3d_fwd_split_stats): no statistics pass.
static int bn_train_fwd_impl(const float* y, const float* add, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, float* out,
                             float* save_mean, float* save_invstd, float* save_scale, float* save_shift, float* workspace, int B,
                             int C, long long S, int groups, int prestats, float* amax, mode_stream_t stream);

// The `_amax` entries also leave the largest finite magnitude of the tensor they write in `out_absmax` / `gy_absmax` (a device buffer of
// MODE_BN_ABSMAX_FLOATS floats whose maximum is the value; zeroed by the call): the power-of-two scale of an fp16-arithmetic consumer
// (mode_conv3d_fwd_split_f16, ...) comes out of the pass that writes the operand, without a pass of its own over the tensor.  NULL:
// the plain call.  (Rounds 5: one-shot thread-local setters in front of the plain calls -- state between two calls that no binder
// other than the Python one could be expected to get right; ABI 30 made it a parameter.)
extern "C" int mode_bn_train_fwd(const float* y, const float* add, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, float* out,
                                 float* save_mean, float* save_invstd, float* save_scale, float* save_shift, float* workspace, int B,
                                 int C, long long S, int groups, mode_stream_t stream) {
  return bn_train_fwd_impl(y, add, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu, out, save_mean,
                           save_invstd, save_scale, save_shift, workspace, B, C, S, groups, 0, nullptr, stream);
}

extern "C" int mode_bn_train_fwd_amax(const float* y, const float* add, const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, float* out,
                                      float* save_mean, float* save_invstd, float* save_scale, float* save_shift, float* workspace, int B,
                                      int C, long long S, int groups, float* out_absmax, mode_stream_t stream) {
  return bn_train_fwd_impl(y, add, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu, out, save_mean,
                           save_invstd, save_scale, save_shift, workspace, B, C, S, groups, 0, out_absmax, stream);
}

static int check_nsplit(int nsplit) {
  // the C pivots sit behind the 2 * C * nsplit partial sums in a workspace of C * 2048 floats: at most 1023 pairs per channel
  MODE_REQUIRE(nsplit > 0 && nsplit <= 1023, MODE_ERR_BAD_ARG, "mode_bn_train_fwd_prestats: %d partial pairs per channel (1..1023)", nsplit);
  return MODE_OK;
}

extern "C" int mode_bn_train_fwd_prestats(const float* y, const float* add, const float* gamma, const float* beta, float* running_mean,
                                          float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu,
                                          float* out, float* save_mean, float* save_invstd, float* save_scale, float* save_shift,
                                          float* workspace, int nsplit, int B, int C, long long S, mode_stream_t stream) {
  int rc = check_nsplit(nsplit);
  if (rc != MODE_OK) return rc;
  return bn_train_fwd_impl(y, add, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu, out, save_mean,
                           save_invstd, save_scale, save_shift, workspace, B, C, S, 1, nsplit, nullptr, stream);
}

extern "C" int mode_bn_train_fwd_prestats_amax(const float* y, const float* add, const float* gamma, const float* beta, float* running_mean,
                                               float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu,
                                               float* out, float* save_mean, float* save_invstd, float* save_scale, float* save_shift,
                                               float* workspace, int nsplit, int B, int C, long long S, float* out_absmax,
                                               mode_stream_t stream) {
  int rc = check_nsplit(nsplit);
  if (rc != MODE_OK) return rc;
  return bn_train_fwd_impl(y, add, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu, out, save_mean,
                           save_invstd, save_scale, save_shift, workspace, B, C, S, 1, nsplit, out_absmax, stream);
}

static int bn_train_fwd_impl(const float* y, const float* add, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, float* out,
                             float* save_mean, float* save_invstd, float* save_scale, float* save_shift, float* workspace, int B,
                             int C, long long S, int groups, int prestats, float* amax, mode_stream_t stream) {
  int rc = check_bn(B, C, S, "mode_bn_train_fwd");
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(B > 0, MODE_ERR_BAD_ARG, "mode_bn_train_fwd: empty batch has no statistics");
  MODE_REQUIRE(y && gamma && beta && out && save_mean && save_invstd && workspace, MODE_ERR_BAD_ARG, "mode_bn_train_fwd: null pointer");
  MODE_REQUIRE(aligned_rows(y, S) && aligned_rows(out, S) && (!add || aligned_rows(add, S)) && aligned16(workspace), MODE_ERR_UNSUPPORTED,
               "mode_bn_train_fwd: unaligned buffer");
  MODE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), MODE_ERR_BAD_ARG, "mode_bn_train_fwd: running stats must come in pairs");
  MODE_REQUIRE((save_scale == nullptr) == (save_shift == nullptr), MODE_ERR_BAD_ARG, "mode_bn_train_fwd: save_scale / save_shift come in pairs");
  MODE_REQUIRE(groups >= 1 && B % groups == 0, MODE_ERR_BAD_ARG, "mode_bn_train_fwd: batch %d not divisible into %d groups", B, groups);
  // the apply pass re-reads the pivot of the shifted sums from y while other blocks of the same launch write `out`
  MODE_REQUIRE(out != y, MODE_ERR_BAD_ARG, "mode_bn_train_fwd: in-place operation (out == y) is not supported");
  hipStream_t st = mode::as_stream(stream);
  if (amax && prestats > 0) {  // (no statistics pass to zero the maximum's buffer on the way)
    rc = mode::absmax_begin(amax, st, "mode_bn_train_fwd");
    if (rc != MODE_OK) return rc;
  }
  const int nsplit = prestats > 0 ? prestats : pick_nsplit(C * groups, S);
  if (prestats <= 0)
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nsplit, C, groups), dim3(NT), 0, st, y, workspace, B, C, S, nsplit, reinterpret_cast<unsigned*>(amax));
  BnCoefArgs k{workspace, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_invstd, save_scale, save_shift,
               momentum, eps, nsplit,
               (double)(B / groups) * (double)S, groups, B / groups, prestats > 0 ? workspace + 2LL * C * nsplit : nullptr,
               reinterpret_cast<unsigned*>(amax)};
  const int BC = B * C;
  const char* who = "mode_bn_train_fwd";
  if (amax) {
    if (relu)
      rc = add ? launch_apply(bn_apply_kernel<true, true, true, true>, BC, S, st, who, y, add, k, out, C, S)
               : launch_apply(bn_apply_kernel<true, false, true, true>, BC, S, st, who, y, y, k, out, C, S);
    else
      rc = add ? launch_apply(bn_apply_kernel<false, true, true, true>, BC, S, st, who, y, add, k, out, C, S)
               : launch_apply(bn_apply_kernel<false, false, true, true>, BC, S, st, who, y, y, k, out, C, S);
    return rc;
  }
  if (relu) {
    if (add) return launch_apply(bn_apply_kernel<true, true, true>, BC, S, st, who, y, add, k, out, C, S);
    return launch_apply(bn_apply_kernel<true, false, true>, BC, S, st, who, y, y, k, out, C, S);
  }
  if (add) return launch_apply(bn_apply_kernel<false, true, true>, BC, S, st, who, y, add, k, out, C, S);
  return launch_apply(bn_apply_kernel<false, false, true>, BC, S, st, who, y, y, k, out, C, S);
}

// The TRAIN coefficients of bn_apply_kernel on their own, one block per channel (one statistics group): for consumers that apply the
// normalisation themselves while they stage their operand (csrc/classif_head.hip) -- same arithmetic, same order, same saved values.
namespace {
__global__ __launch_bounds__(NT) void bn_finalize_kernel(const float* __restrict__ y, BnCoefArgs k, int C, long long S) {
  __shared__ double shd[2 * NW];
  const int c = blockIdx.x;
  double s0, s1;
  reduce_partials(k.partial, c, k.nsplit, s0, s1, shd);
  if (threadIdx.x != 0) return;
  const double dm = s0 / k.count;  // mean of y - K, K = the pivot of bn_stats_kernel
  const double mean = (double)y[(long long)c * S] + dm;
  double var = s1 / k.count - dm * dm;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)k.eps);
  const double sc = (double)k.gamma[c] * invstd;
  k.save_mean[c] = (float)mean;
  k.save_invstd[c] = (float)invstd;
  k.save_scale[c] = (float)sc;
  k.save_shift[c] = (float)((double)k.beta[c] - mean * sc);
  if (k.running_mean) {
    const double unbiased = k.count > 1.0 ? var * k.count / (k.count - 1.0) : var;
    k.running_mean[c] = (float)((1.0 - k.momentum) * (double)k.running_mean[c] + k.momentum * mean);
    k.running_var[c] = (float)((1.0 - k.momentum) * (double)k.running_var[c] + k.momentum * unbiased);
  }
  if (c == 0 && k.num_batches_tracked) *k.num_batches_tracked += 1;
}
}  // namespace

int mode::bn_train_coefficients(const float* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                long long* num_batches_tracked, float momentum, float eps, float* save_mean, float* save_invstd,
                                float* save_scale, float* save_shift, float* workspace, int B, int C, long long S, hipStream_t st,
                                const char* who) {
  int rc = ::check_bn(B, C, S, who);  // (the one of this file, not mode::check_bn(const mode_bn_epilogue*, ...))
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(B > 0, MODE_ERR_BAD_ARG, "%s: empty batch has no statistics", who);
  MODE_REQUIRE(y && gamma && beta && save_mean && save_invstd && save_scale && save_shift && workspace, MODE_ERR_BAD_ARG, "%s: null pointer", who);
  MODE_REQUIRE(aligned_rows(y, S) && aligned16(workspace), MODE_ERR_UNSUPPORTED, "%s: unaligned buffer", who);
  MODE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), MODE_ERR_BAD_ARG, "%s: running stats must come in pairs", who);
  const int nsplit = pick_nsplit(C, S);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nsplit, C, 1), dim3(NT), 0, st, y, workspace, B, C, S, nsplit, (unsigned*)nullptr);
  BnCoefArgs k{workspace, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_invstd, save_scale, save_shift,
               momentum, eps, nsplit, (double)B * (double)S, 1, B, nullptr};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(NT), 0, st, y, k, C, S);
  return mode::check_launch(who);
}

extern "C" int mode_bn_eval_fwd(const float* y, const float* add, const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float eps, int relu, float* out, int B, int C, long long S,
                                mode_stream_t stream) {
  int rc = check_bn(B, C, S, "mode_bn_eval_fwd");
  if (rc != MODE_OK) return rc;
  if (B == 0) return MODE_OK;
  MODE_REQUIRE(y && gamma && beta && running_mean && running_var && out, MODE_ERR_BAD_ARG, "mode_bn_eval_fwd: null pointer");
  MODE_REQUIRE(aligned_rows(y, S) && aligned_rows(out, S) && (!add || aligned_rows(add, S)), MODE_ERR_UNSUPPORTED, "mode_bn_eval_fwd: unaligned buffer");
  hipStream_t st = mode::as_stream(stream);
  BnCoefArgs k{nullptr, gamma, beta, const_cast<float*>(running_mean), const_cast<float*>(running_var), nullptr, nullptr, nullptr, nullptr,
               nullptr, 0.f,
               eps, 0, 0.0, 1, 1, nullptr};
  const int BC = B * C;
  const char* who = "mode_bn_eval_fwd";
  if (relu) {
    if (add) return launch_apply(bn_apply_kernel<true, true, false>, BC, S, st, who, y, add, k, out, C, S);
    return launch_apply(bn_apply_kernel<true, false, false>, BC, S, st, who, y, y, k, out, C, S);
  }
  if (add) return launch_apply(bn_apply_kernel<false, true, false>, BC, S, st, who, y, add, k, out, C, S);
  return launch_apply(bn_apply_kernel<false, false, false>, BC, S, st, who, y, y, k, out, C, S);
}

extern "C" int mode_bn_train_bwd(const float* gout, const float* y, const float* out, const float* gamma, const float* save_mean,
                                 const float* save_invstd, const float* save_scale, const float* save_shift, int relu, float* gy,
                                 float* gadd, float* ggamma, float* gbeta, int accumulate, float* workspace, int B, int C, long long S,
                                 int groups, mode_stream_t stream) {
  return mode_bn_train_bwd_amax(gout, y, out, gamma, save_mean, save_invstd, save_scale, save_shift, relu, gy, gadd, ggamma, gbeta,
                                accumulate, workspace, B, C, S, groups, nullptr, stream);
}

extern "C" int mode_bn_train_bwd_amax(const float* gout, const float* y, const float* out, const float* gamma, const float* save_mean,
                                      const float* save_invstd, const float* save_scale, const float* save_shift, int relu, float* gy,
                                      float* gadd, float* ggamma, float* gbeta, int accumulate, float* workspace, int B, int C,
                                      long long S, int groups, float* gy_absmax, mode_stream_t stream) {
  float* amax = gy_absmax;
  int rc = check_bn(B, C, S, "mode_bn_train_bwd");
  if (rc != MODE_OK) return rc;
  MODE_REQUIRE(B > 0, MODE_ERR_BAD_ARG, "mode_bn_train_bwd: empty batch");
  MODE_REQUIRE(gout && y && gamma && save_mean && save_invstd && gy && ggamma && gbeta && workspace, MODE_ERR_BAD_ARG,
               "mode_bn_train_bwd: null pointer");
  MODE_REQUIRE((save_scale == nullptr) == (save_shift == nullptr), MODE_ERR_BAD_ARG, "mode_bn_train_bwd: save_scale / save_shift come in pairs");
  MODE_REQUIRE(!relu || out || save_scale, MODE_ERR_BAD_ARG,
               "mode_bn_train_bwd: the ReLU mask needs the forward output, or (no residual add) the forward's scale / shift");
  MODE_REQUIRE(aligned_rows(gout, S) && aligned_rows(y, S) && aligned_rows(gy, S) && (!out || aligned_rows(out, S)) &&
                   (!gadd || aligned_rows(gadd, S)) && aligned16(workspace),
               MODE_ERR_UNSUPPORTED, "mode_bn_train_bwd: unaligned buffer");
  MODE_REQUIRE(groups >= 1 && B % groups == 0, MODE_ERR_BAD_ARG, "mode_bn_train_bwd: batch %d not divisible into %d groups", B, groups);
  hipStream_t st = mode::as_stream(stream);
  const int nsplit = pick_nsplit(C * groups, S);
  float* partial = workspace;
  // mask source: the forward output when given (mandatory if a residual was added before the ReLU), else y and the coefficients
  const int mode = !relu ? 0 : (out ? 1 : 2);
  const float* o = out ? out : y;
  unsigned* zero = reinterpret_cast<unsigned*>(amax);  // (the maximum's buffer is zeroed by the statistics pass)
  if (mode == 0)
    hipLaunchKernelGGL(bn_bwd_stats_kernel<0>, dim3(nsplit, C, groups), dim3(NT), 0, st, gout, y, o, save_scale, save_shift, partial, B, C, S, nsplit, zero);
  else if (mode == 1)
    hipLaunchKernelGGL(bn_bwd_stats_kernel<1>, dim3(nsplit, C, groups), dim3(NT), 0, st, gout, y, o, save_scale, save_shift, partial, B, C, S, nsplit, zero);
  else
    hipLaunchKernelGGL(bn_bwd_stats_kernel<2>, dim3(nsplit, C, groups), dim3(NT), 0, st, gout, y, o, save_scale, save_shift, partial, B, C, S, nsplit, zero);
  const int BC = B * C;
  const char* who = "mode_bn_train_bwd";
  const double count = (double)(B / groups) * (double)S;
#define MODE_BN_BWD_APPLY(M, G)                                                                                                         \
  (amax ? launch_apply(bn_bwd_apply_kernel<M, G, true>, BC, S, st, who, gout, y, o, save_scale, save_shift, partial, gamma, save_mean,    \
                       save_invstd, ggamma, gbeta, accumulate, nsplit, count, groups, B / groups, gy, (G) ? gadd : gy, C, S,            \
                       reinterpret_cast<unsigned*>(amax))                                                                             \
        : launch_apply(bn_bwd_apply_kernel<M, G, false>, BC, S, st, who, gout, y, o, save_scale, save_shift, partial, gamma, save_mean,   \
                       save_invstd, ggamma, gbeta, accumulate, nsplit, count, groups, B / groups, gy, (G) ? gadd : gy, C, S,            \
                       (unsigned*)nullptr))
  if (mode == 0) return gadd ? MODE_BN_BWD_APPLY(0, true) : MODE_BN_BWD_APPLY(0, false);
  if (mode == 1) return gadd ? MODE_BN_BWD_APPLY(1, true) : MODE_BN_BWD_APPLY(1, false);
  return gadd ? MODE_BN_BWD_APPLY(2, true) : MODE_BN_BWD_APPLY(2, false);
#undef MODE_BN_BWD_APPLY
}

// ---------------------------------------------------------------------------------------------------------------------
// out = a + b [+ c [+ d]] in one pass (fixed association ((a + b) + c) + d): the gradient of a tensor with several consumers.  autograd
// accumulates such gradients pairwise -- n - 1 launches that each read two tensors and write one; for the 4 consumers of cost0
// (models/mode_disparity.py:119-125: the input of dres2 and the three residual adds) that is 3.6 GB of traffic per step instead of 2 GB.
namespace {
__global__ __launch_bounds__(NT) void sum_n_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                   const float* __restrict__ d, float* __restrict__ out, long long n) {
  const long long n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  const float4* c4 = reinterpret_cast<const float4*>(c);
  const float4* d4 = reinterpret_cast<const float4*>(d);
  float4* o4 = reinterpret_cast<float4*>(out);
  const long long st = (long long)gridDim.x * NT;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += st) {
    float4 v = a4[i];
    const float4 w = b4[i];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    if (c) {
      const float4 u = c4[i];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    if (d) {
      const float4 u = d4[i];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    o4[i] = v;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * NT + threadIdx.x; i < n; i += st) {
    float v = a[i] + b[i];
    if (c) v += c[i];
    if (d) v += d[i];
    out[i] = v;
  }
}
}  // namespace

extern "C" int mode_sum_n(const float* a, const float* b, const float* c, const float* d, float* out, long long n, mode_stream_t stream) {
  MODE_REQUIRE(n >= 0, MODE_ERR_BAD_ARG, "mode_sum_n: negative size");
  if (n == 0) return MODE_OK;
  MODE_REQUIRE(a && b && out && (c || !d), MODE_ERR_BAD_ARG, "mode_sum_n: null pointer (operands are a, b[, c[, d]])");
  MODE_REQUIRE(aligned16(a) && aligned16(b) && aligned16(out) && (!c || aligned16(c)) && (!d || aligned16(d)), MODE_ERR_UNSUPPORTED,
               "mode_sum_n: unaligned buffer");
  const long long blocks = (n / 4 + NT - 1) / NT;
  const int grid = (int)std::max<long long>(1, std::min<long long>(blocks, 16LL * kNumCU));
  hipLaunchKernelGGL(sum_n_kernel, dim3(grid), dim3(NT), 0, mode::as_stream(stream), a, b, c, d, out, n);
  return mode::check_launch("mode_sum_n");
}
