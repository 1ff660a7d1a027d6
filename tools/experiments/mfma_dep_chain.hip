// Does it matter how many accumulators a run of v_mfma_f32_32x32x16_bf16 cycles through?  (The compiler likes to cluster the six
// terms of one accumulator back to back.)   hipcc --offload-arch=gfx950 -O3 -o /tmp/mdc tools/experiments/mfma_dep_chain.hip && /tmp/mdc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))

template <int NACC, int RUN>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
  const bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0, threadIdx.x, 0));
  const bf16x8 b = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0, 0, threadIdx.x));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      const int w = NACC == 1 ? 0 : (RUN > 1 ? (i / RUN) % NACC : i % NACC);
      if (w == 0) MFMA(c0);
      if (w == 1) MFMA(c1);
      if (w == 2) MFMA(c2);
      if (w == 3) MFMA(c3);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int RUN>
void run(float* out, int threads, const char* what) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<NACC, RUN><<<256, threads>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<NACC, RUN><<<256, threads>>>(out, 400);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%d wave(s)/SIMD  %-58s %.3f us per 24 MFMAs per wave\n", threads / 256, what, ms * 1e3 / 400);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  for (int threads : {256, 512}) {
    run<4, 1>(out, threads, "4 accumulators round robin");
    run<2, 1>(out, threads, "2 accumulators alternating");
    run<1, 1>(out, threads, "1 accumulator (every MFMA depends on the one before)");
    run<4, 6>(out, threads, "4 accumulators, 6 in a row on each");
    run<4, 2>(out, threads, "4 accumulators, 2 in a row on each");
  }
  return 0;
}
