// EXPERIMENT (not part of libmode_hip.so): fp32 3x3x3 convolution on the bf16 matrix pipe with three-way operand splitting.
//
// DESIGN.md section 6 names "a 6-term bf16 split" as the only large lever left on the 45 ms of 3-D convolutions and says it was
// not taken because it changes the arithmetic type of the path.  This file measures what that lever is worth and what it does to
// the result, so that the decision can be taken on numbers:
//   a = a1 + a2 + a3 exactly (a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2): 3 x 8 significant bits = fp32's 24);
//   a * b ~= a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1   (the dropped terms are <= 2^-23 |a b|), each product exact in the
//   fp32 accumulator of v_mfma_f32_32x32x16_bf16 -- six bf16 MFMAs (6 x 32 cycles for K = 16) in place of eight fp32 MFMAs
//   (8 x 64 cycles): 2.67 x on paper.
// Layer: the benchmark's dominant shape, Conv3d(32 -> 32, k3 p1) on 2 x 32 x 48 x 256 x 128 (87 GFLOP per launch).
// The kernel keeps the production tile (2 x 8 rows x 32 columns, 4 waves x 4 rows); the input tile is split ONCE when it is staged
// (three bf16 images in LDS, [channel octet][row][w][8 channels] so that a B fragment is one ds_read_b128) and used by 27 taps.
//
//   hipcc --offload-arch=gfx950 -O3 -o conv3d_bf16x6 conv3d_bf16x6.hip && ./conv3d_bf16x6
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int B = 2, C = 32, D = 48, H = 256, W = 128;
constexpr int TD = 2, NT = 256, IW = 34;

__host__ __device__ inline uint32_t f2u(float f) {
  union { float f; uint32_t u; } v;
  v.f = f;
  return v.u;
}
__host__ __device__ inline float u2f(uint32_t u) {
  union { float f; uint32_t u; } v;
  v.u = u;
  return v.f;
}
// bf16 (as the upper half of a float) nearest to f, ties to even
__host__ __device__ inline float bf16_round(float f) {
  uint32_t u = f2u(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u2f(u & 0xffff0000u);
}
__host__ __device__ inline void split3(float v, uint16_t& p1, uint16_t& p2, uint16_t& p3) {
  const float a1 = bf16_round(v);
  const float r1 = v - a1;
  const float a2 = bf16_round(r1);
  const float a3 = bf16_round(r1 - a2);
  p1 = (uint16_t)(f2u(a1) >> 16);
  p2 = (uint16_t)(f2u(a2) >> 16);
  p3 = (uint16_t)(f2u(a3) >> 16);
}

// wp[step(2)][tap(27)][piece(3)][lane(64)] = 8 bf16: W[o = lane & 31][c = step*16 + (lane >> 5)*8 + j][tap]
// TH = 8: the production tile (2 x 8 rows; 130 KB of LDS: one workgroup per CU, nothing overlaps its staging);
// TH = 4: 2 x 4 rows, 78 KB: two workgroups per CU, one stages while the other multiplies.
template <int TH>
__global__ __launch_bounds__(NT) void conv3d_bf16x6_kernel(const float* __restrict__ x, const uint4* __restrict__ wp, float* __restrict__ y) {
  constexpr int ID = TD + 2, IH = TH + 2;
  constexpr int ROWS = ID * IH;         // haloed rows
  constexpr int PIECE = 2 * ROWS * IW;  // uint4 (8 bf16) entries per piece and 16-channel step: [khalf][row][w]
  constexpr int R = TD * TH / 4;        // output rows per wave
  extern __shared__ __attribute__((aligned(16))) uint4 sm[];  // [piece][khalf][row][w]
  constexpr int nWt = W / 32, nHt = H / TH, nDt = D / TD;
  int t = blockIdx.x;
  const int wt = t % nWt;
  t /= nWt;
  const int ht = t % nHt;
  t /= nHt;
  const int dt = t % nDt;
  const int b = t / nDt;
  const int w0 = wt * 32, h0 = ht * TH, d0 = dt * TD;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const long long HW = (long long)H * W, DHW = (long long)D * HW;
  const float* xb = x + (long long)b * C * DHW;

  f32x16 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = (f32x16){0};

  for (int step = 0; step < C / 16; ++step) {
    if (step) __syncthreads();
    // ---- staging: item = (khalf, row, w): 8 channel values -> three uint4 of bf16.  All loads of the step are issued first
    // (unconditional, from clamped addresses), then split and written: a load -> split -> write loop is one HBM round trip per item.
    constexpr int NITEM = 2 * ROWS * IW, KIT = (NITEM + NT - 1) / NT;
    float raw[KIT][8];
    bool okk[KIT];
#pragma unroll
    for (int k = 0; k < KIT; ++k) {
      const int item = min(tid + k * NT, NITEM - 1);
      const int wi = item % IW, rw = (item / IW) % ROWS, kh = item / (IW * ROWS);
      const int gd = d0 + rw / IH - 1, gh = h0 + rw % IH - 1, gw = w0 + wi - 1;
      okk[k] = gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W;
      const float* p = xb + (long long)(step * 16 + kh * 8) * DHW + (okk[k] ? gd * HW + (long long)gh * W + gw : 0);
#pragma unroll
      for (int c = 0; c < 8; ++c) raw[k][c] = p[(long long)c * DHW];
    }
#pragma unroll
    for (int k = 0; k < KIT; ++k) {
      const int item = tid + k * NT;
      uint32_t q1[4], q2[4], q3[4];
#pragma unroll
      for (int c2 = 0; c2 < 4; ++c2) {
        const float v0 = okk[k] ? raw[k][2 * c2] : 0.f, v1 = okk[k] ? raw[k][2 * c2 + 1] : 0.f;
        uint16_t a1, a2, a3, b1, b2, b3;
        split3(v0, a1, a2, a3);
        split3(v1, b1, b2, b3);
        q1[c2] = a1 | ((uint32_t)b1 << 16);
        q2[c2] = a2 | ((uint32_t)b2 << 16);
        q3[c2] = a3 | ((uint32_t)b3 << 16);
      }
      if (item < NITEM) {
        sm[0 * PIECE + item] = make_uint4(q1[0], q1[1], q1[2], q1[3]);
        sm[1 * PIECE + item] = make_uint4(q2[0], q2[1], q2[2], q2[3]);
        sm[2 * PIECE + item] = make_uint4(q3[0], q3[1], q3[2], q3[3]);
      }
    }
    __syncthreads();
    const uint4* wq = wp + ((long long)step * 27) * 3 * 64 + lane;
    const uint4* bbase = sm + half * (ROWS * IW) + j;
    // weights one tap ahead (as in the library's kernels: requested right before their first use they cost an L2 round trip per tap)
    uint4 an0 = wq[0], an1 = wq[64], an2 = wq[128];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int toff = ((tap / 9) * IH + (tap / 3) % 3) * IW + tap % 3;
      const bf16x8 a1 = __builtin_bit_cast(bf16x8, an0), a2 = __builtin_bit_cast(bf16x8, an1), a3 = __builtin_bit_cast(bf16x8, an2);
      if (tap + 1 < 27) {
        an0 = wq[((tap + 1) * 3 + 0) * 64];
        an1 = wq[((tap + 1) * 3 + 1) * 64];
        an2 = wq[((tap + 1) * 3 + 2) * 64];
      }
      bf16x8 bq[R][3];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int row = wave * R + r;
        const int off = ((row / TH) * IH + row % TH) * IW + toff;
        bq[r][0] = __builtin_bit_cast(bf16x8, bbase[off]);
        bq[r][1] = __builtin_bit_cast(bf16x8, bbase[PIECE + off]);
        bq[r][2] = __builtin_bit_cast(bf16x8, bbase[2 * PIECE + off]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // smallest terms first; rows interleaved so that consecutive MFMAs never hit the same accumulator
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bq[r][0], acc[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[r][2], acc[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bq[r][1], acc[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bq[r][0], acc[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[r][1], acc[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[r][0], acc[r], 0, 0, 0);
    }
  }
  float* yb = y + (long long)b * C * DHW;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int row = wave * R + r;
    const long long sp = (long long)(d0 + row / TH) * HW + (long long)(h0 + row % TH) * W + w0 + j;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int o = (q & 3) + 8 * (q >> 2) + 4 * half;
      yb[o * DHW + sp] = acc[r][q];
    }
  }
}

// references at sampled output points: exact (double) and a plain sequential fp32 accumulation (what "an fp32 result" scatters by)
__global__ void sample_ref_kernel(const float* __restrict__ x, const float* __restrict__ w, const int* __restrict__ pts, int n, double* ref64,
                                  float* ref32) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = pts[5 * i], o = pts[5 * i + 1], d = pts[5 * i + 2], h = pts[5 * i + 3], ww = pts[5 * i + 4];
  double s = 0.0;
  float f = 0.f;
  for (int c = 0; c < C; ++c)
    for (int kd = 0; kd < 3; ++kd)
      for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
          const int gd = d + kd - 1, gh = h + kh - 1, gw = ww + kw - 1;
          if (gd < 0 || gd >= D || gh < 0 || gh >= H || gw < 0 || gw >= W) continue;
          const float xv = x[(((long long)b * C + c) * D + gd) * H * W + (long long)gh * W + gw];
          const float wv = w[((o * C + c) * 27) + kd * 9 + kh * 3 + kw];
          s += (double)xv * (double)wv;
          f = fmaf(xv, wv, f);
        }
  ref64[i] = s;
  ref32[i] = f;
}

int main() {
  const size_t n = (size_t)B * C * D * H * W;
  std::vector<float> hx(n), hw((size_t)C * C * 27);
  uint32_t st = 12345u;
  auto rnd = [&]() {  // uniform in [-1, 1)
    st = st * 1664525u + 1013904223u;
    return (float)((st >> 8) & 0xffffff) / 8388608.0f - 1.0f;
  };
  for (auto& v : hx) v = rnd() * 1.7f + 0.25f;
  for (auto& v : hw) v = rnd() * 0.07f;
  std::vector<uint16_t> hp((size_t)2 * 27 * 3 * 64 * 8);
  for (int step = 0; step < 2; ++step)
    for (int tap = 0; tap < 27; ++tap)
      for (int lane = 0; lane < 64; ++lane)
        for (int jj = 0; jj < 8; ++jj) {
          const int o = lane & 31, c = step * 16 + (lane >> 5) * 8 + jj;
          uint16_t p1, p2, p3;
          split3(hw[((size_t)o * C + c) * 27 + tap], p1, p2, p3);
          const size_t base = (((size_t)step * 27 + tap) * 3) * 64 * 8 + (size_t)lane * 8 + jj;
          hp[base] = p1;
          hp[base + 64 * 8] = p2;
          hp[base + 2 * 64 * 8] = p3;
        }
  float *dx, *dw, *dy;
  uint4* dp;
  hipMalloc(&dx, n * 4);
  hipMalloc(&dy, n * 4);
  hipMalloc(&dw, hw.size() * 4);
  hipMalloc(&dp, hp.size() * 2);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dp, hp.data(), hp.size() * 2, hipMemcpyHostToDevice);
  const double flop = 2.0 * 27 * C * C * (double)B * D * H * W;
  auto run = [&](auto kernel, int TH, const char* what) {
    const size_t lds = 3 * (size_t)(2 * (TD + 2) * (TH + 2) * IW) * 16;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      fprintf(stderr, "cannot get %zu B of LDS\n", lds);
      exit(1);
    }
    const int grid = B * (D / TD) * (H / TH) * (W / 32);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(NT), lds, 0, dx, dp, dy);
    hipEventRecord(e0);
    const int iters = 10;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kernel, dim3(grid), dim3(NT), lds, 0, dx, dp, dy);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) {
      fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
      exit(1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    printf("conv3d 32->32 @48x256x128 B=2 on bf16 x 6, %s (%zu KB LDS): %.3f ms per launch = %.1f TFLOP/s fp32-equivalent\n", what, lds / 1024, ms,
           flop / ms * 1e-9);
  };
  printf("(fp32 MFMA kernel of the library, same layer: 1.36 ms, 128 TFLOP/s)\n");
  run(conv3d_bf16x6_kernel<8>, 8, "tile 2 x 8 rows, one workgroup per CU");
  run(conv3d_bf16x6_kernel<4>, 4, "tile 2 x 4 rows, two workgroups per CU");

  const int NS = 1 << 15;
  std::vector<int> pts(5 * NS);
  for (int i = 0; i < NS; ++i) {
    st = st * 1664525u + 1013904223u; pts[5 * i] = (st >> 16) % B;
    st = st * 1664525u + 1013904223u; pts[5 * i + 1] = (st >> 16) % C;
    st = st * 1664525u + 1013904223u; pts[5 * i + 2] = (i % 7 == 0) ? ((st >> 16) & 1) * (D - 1) : (st >> 16) % D;
    st = st * 1664525u + 1013904223u; pts[5 * i + 3] = (i % 5 == 0) ? ((st >> 16) & 1) * (H - 1) : (st >> 16) % H;
    st = st * 1664525u + 1013904223u; pts[5 * i + 4] = (i % 3 == 0) ? ((st >> 16) & 1) * (W - 1) : (st >> 16) % W;
  }
  int* dpts;
  double* dr64;
  float* dr32;
  hipMalloc(&dpts, pts.size() * 4);
  hipMalloc(&dr64, NS * 8);
  hipMalloc(&dr32, NS * 4);
  hipMemcpy(dpts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(sample_ref_kernel, dim3(NS / 256), dim3(256), 0, 0, dx, dw, dpts, NS, dr64, dr32);
  std::vector<double> r64(NS);
  std::vector<float> r32(NS), hy(n);
  hipMemcpy(r64.data(), dr64, NS * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r32.data(), dr32, NS * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
  double e_split = 0, e_f32 = 0, s_split = 0, s_f32 = 0, scale = 0;
  for (int i = 0; i < NS; ++i) {
    const size_t idx = ((((size_t)pts[5 * i] * C + pts[5 * i + 1]) * D + pts[5 * i + 2]) * H + pts[5 * i + 3]) * W + pts[5 * i + 4];
    const double a = std::fabs((double)hy[idx] - r64[i]), bb = std::fabs((double)r32[i] - r64[i]);
    e_split = std::fmax(e_split, a);
    e_f32 = std::fmax(e_f32, bb);
    s_split += a * a;
    s_f32 += bb * bb;
    scale = std::fmax(scale, std::fabs(r64[i]));
  }
  printf("error against an exact (double) evaluation at %d sampled outputs (|y| up to %.2f):\n", NS, scale);
  printf("  bf16 x 6 split      : max %.3e   rms %.3e\n", e_split, std::sqrt(s_split / NS));
  printf("  sequential fp32 fma : max %.3e   rms %.3e\n", e_f32, std::sqrt(s_f32 / NS));
  return 0;
}
