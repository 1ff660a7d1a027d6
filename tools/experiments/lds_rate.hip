// LDS instruction throughput per CU on gfx950, conflict-free patterns, 8 waves per workgroup, one workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsr tools/experiments/lds_rate.hip && /tmp/ldsr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

enum { R_B32, R2_B32, R2_B32_FAR, R_B64, R2_B64, R_B128, W_B32, W_B64, W_B128, R2_B32_HALVES, NK };
static const char* names[] = {"ds_read_b32 (lane -> dword)", "ds_read2_b32 offset1:81 (lane -> dword)", "ds_read2_b32 offset1:1", "ds_read_b64 (lane -> 8 B)",
                              "ds_read2_b64 offset1:1", "ds_read_b128 (lane -> 16 B)", "ds_write_b32", "ds_write_b64", "ds_write_b128",
                              "ds_read2_b32 offset1:81, lanes 32..63 at +8*649 dwords (the kernel's pattern)"};
static const int bytes[] = {256, 512, 512, 512, 1024, 1024, 256, 512, 1024, 512};

template <int K>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters) {
  __shared__ unsigned lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned a4 = lane * 4 + wave * 4096, a8 = lane * 8 + wave * 4096, a16 = lane * 16 + wave * 4096;
  unsigned ah = ((lane & 31) + (lane >> 5) * 8 * 649) * 4 + wave * 512;
  unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  u32x2 d0 = {0, 0}, d1 = d0, d2 = d0, d3 = d0;
  u32x4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
#define ONE(R, D, Q)                                                                                        \
  if (K == R_B32) asm volatile("ds_read_b32 %0, %1" : "=v"(R) : "v"(a4));                                    \
  if (K == R2_B32) asm volatile("ds_read2_b32 %0, %1 offset1:81" : "=v"(D) : "v"(a4));                        \
  if (K == R2_B32_FAR) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(D) : "v"(a4));                     \
  if (K == R_B64) asm volatile("ds_read_b64 %0, %1" : "=v"(D) : "v"(a8));                                    \
  if (K == R2_B64) asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(Q) : "v"(a16));                        \
  if (K == R_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(Q) : "v"(a16));                                 \
  if (K == W_B32) asm volatile("ds_write_b32 %0, %1" : : "v"(a4), "v"(R));                                    \
  if (K == W_B64) asm volatile("ds_write_b64 %0, %1" : : "v"(a8), "v"(D));                                    \
  if (K == W_B128) asm volatile("ds_write_b128 %0, %1" : : "v"(a16), "v"(Q));                                 \
  if (K == R2_B32_HALVES) asm volatile("ds_read2_b32 %0, %1 offset1:81" : "=v"(D) : "v"(ah));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ONE(r0, d0, q0) ONE(r1, d1, q1) ONE(r2, d2, q2) ONE(r3, d3, q3)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  out[blockIdx.x * 512 + threadIdx.x] = (float)(r0 + r1 + r2 + r3 + d0[0] + d1[1] + d2[0] + d3[1] + q0[0] + q1[1] + q2[2] + q3[3]);
}

template <int K>
void run(float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  rate_kernel<K><<<256, 512>>>(out, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  rate_kernel<K><<<256, 512>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_instr = ms * 1e6 / iters / (32.0 * 8);  // per wave-instruction, per CU
  printf("%-82s %.2f ns per wave-instruction per CU = %.1f cycles at 2.1 GHz, %.0f B/clk\n", names[K], ns_per_instr, ns_per_instr * 2.1,
         bytes[K] / (ns_per_instr * 2.1));
}

template <int K>
void all(float* out) {
  run<K>(out);
  if constexpr (K + 1 < NK) all<K + 1>(out);
}

int main() {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * 4);
  all<0>(out);
  return 0;
}
