// How much plain vector / LDS work hides under v_mfma_f32_32x32x16_bf16 on gfx950?  (the spherical kernels run their
// matrix pipe 27-35 % busy with 4-6 vector instructions per MFMA; the 3-D kernels 70 % busy with 2.6.)
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/experiments/mfma_valu_overlap.hip && /tmp/mvo
//
// One workgroup per CU, 4 or 8 waves (1 or 2 per SIMD).  Loop body = 24 MFMAs on 4 independent accumulators (the order of a tap of
// sphere_fwd_split_kernel); after each MFMA: NV independent v_fmac_f32 (8 chains) and, every other MFMA, NL ds_read_b128.  Everything
// is `asm volatile`, so the listing is the program order.  Printed: core-clock cycles per loop iteration (s_memtime around the loop,
// averaged over the waves) against the 24 x 32 = 768 cycles the matrix pipe needs per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float sp2 __attribute__((ext_vector_type(2)));

#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define VALU(R) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(R) : "v"(s0), "v"(s1))
#define VPK(R) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(R) : "v"(p0), "v"(p1))

template <int NV, int NL, int BAR, int PK>
__global__ __launch_bounds__(512) void overlap_kernel(float* out, long long* cyc, int iters) {
  __shared__ uint4 lds[4096];
  f32x16 acc0, acc1, acc2, acc3;
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.f;
  const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0, threadIdx.x));
  const bf16x8 b = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0, threadIdx.x, 0));
  float v0 = 0.f, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
  sp2 q0 = {0.f, 1.f}, q1 = {1.f, 1.f}, q2 = {2.f, 1.f}, q3 = {3.f, 1.f};
  const float s0 = 1.0001f, s1 = 1e-6f * threadIdx.x;
  const sp2 p0 = {1.0001f, 0.5f}, p1 = {1e-6f, 1e-7f};
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_uint4(i, 0, 0, 0);
  __syncthreads();
  const unsigned laddr = (unsigned)(threadIdx.x & 63) * 16u + (threadIdx.x >> 6) * 4096u;
  uint4 ld0 = make_uint4(0, 0, 0, 0), ld1 = ld0, ld2 = ld0, ld3 = ld0;
#define VGROUP(n)                                                        \
  if (NV > n) {                                                           \
    if (PK) { if ((n) % 4 == 0) VPK(q0); else if ((n) % 4 == 1) VPK(q1); else if ((n) % 4 == 2) VPK(q2); else VPK(q3); } \
    else { if ((n) % 8 == 0) VALU(v0); else if ((n) % 8 == 1) VALU(v1); else if ((n) % 8 == 2) VALU(v2); else if ((n) % 8 == 3) VALU(v3); \
           else if ((n) % 8 == 4) VALU(v4); else if ((n) % 8 == 5) VALU(v5); else if ((n) % 8 == 6) VALU(v6); else VALU(v7); } \
  }
#define LREAD(R) asm volatile("ds_read_b128 %0, %1" : "=v"(R) : "v"(laddr))
#define STEP(ACC, EVEN)                                                                                                   \
  MFMA(ACC);                                                                                                              \
  VGROUP(0) VGROUP(1) VGROUP(2) VGROUP(3) VGROUP(4) VGROUP(5) VGROUP(6) VGROUP(7)                                          \
  if (NL > 0 && EVEN) LREAD(ld0);                                                                                         \
  if (NL > 1 && EVEN) LREAD(ld1);                                                                                         \
  if (NL > 2 && EVEN) LREAD(ld2);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      STEP(acc0, 1) STEP(acc1, 0) STEP(acc2, 1) STEP(acc3, 0)
    }
    if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (BAR) asm volatile("s_barrier" ::: "memory");
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
  s += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + q0[0] + q1[1] + q2[0] + q3[1];
  s += (float)(ld0.x + ld1.y + ld2.x + ld3.y);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NV, int NL, int BAR, int PK>
void run(int threads, float* out, long long* cyc) {
  const int iters = 400, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  overlap_kernel<NV, NL, BAR, PK><<<blocks, threads>>>(out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  overlap_kernel<NV, NL, BAR, PK><<<blocks, threads>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const int nw = blocks * threads / 64;
  std::vector<long long> h(nw);
  hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
  double sum = 0;
  for (long long c : h) sum += (double)c;
  const double per_iter = sum / nw / iters;  // s_memtime ticks (100 MHz on gfx9: converted with the wall time below)
  const double wall_cycles = ms * 1e-3 / iters;  // seconds per iteration
  const int wps = threads / 256;
  const double tflops = 2.0 * 32 * 32 * 16 * 24 * (double)nw / wall_cycles * 1e-12;
  printf("waves/SIMD %d  VALU/MFMA %d%s  ds_read_b128 per 2 MFMA %d  barrier %d :  %.3f us per iteration, counter %.1f per iteration, %.0f TFLOP/s bf16 (%.0f%% of 2500), MFMA-only time would be %.3f us at 2.4 GHz\n",
         wps, NV, PK ? " (v_pk_fma_f32)" : "", NL, BAR, wall_cycles * 1e6, per_iter, tflops, tflops / 25.0, wps * 768.0 / 2.4e3);
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&cyc, 256 * 8 * sizeof(long long));
  for (int threads : {256, 512}) {
    run<0, 0, 0, 0>(threads, out, cyc);
    run<1, 0, 0, 0>(threads, out, cyc);
    run<2, 0, 0, 0>(threads, out, cyc);
    run<3, 0, 0, 0>(threads, out, cyc);
    run<4, 0, 0, 0>(threads, out, cyc);
    run<5, 0, 0, 0>(threads, out, cyc);
    run<6, 0, 0, 0>(threads, out, cyc);
    run<7, 0, 0, 0>(threads, out, cyc);
    run<8, 0, 0, 0>(threads, out, cyc);
    run<2, 0, 0, 1>(threads, out, cyc);
    run<4, 0, 0, 1>(threads, out, cyc);
    run<0, 1, 0, 0>(threads, out, cyc);
    run<0, 3, 0, 0>(threads, out, cyc);
    run<4, 1, 0, 0>(threads, out, cyc);
    run<4, 3, 0, 0>(threads, out, cyc);
    run<4, 3, 1, 0>(threads, out, cyc);
    run<2, 1, 1, 0>(threads, out, cyc);
  }
  return 0;
}
