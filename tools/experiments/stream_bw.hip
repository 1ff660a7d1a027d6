// What read / read+write bandwidth does a plain streaming kernel reach on this chip, as a function of loads in flight and grid shape?
// (BatchNorm passes run at 5.2-5.7 TB/s = 66-71 % of 8 TB/s.)   hipcc --offload-arch=gfx950 -O3 -o /tmp/sbw tools/experiments/stream_bw.hip && /tmp/sbw
#include <hip/hip_runtime.h>
#include <cstdio>

template <int U, int NTH>
__global__ __launch_bounds__(NTH) void read_sum(const float4* __restrict__ p, float* __restrict__ out, long long n4) {
  float s = 0.f;
  const long long stride = (long long)gridDim.x * NTH;
  long long i = (long long)blockIdx.x * NTH + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  for (; i < n4; i += stride) {
    const float4 v = p[i];
    s += (v.x + v.y) + (v.z + v.w);
  }
  if (s == 12345.678f) out[0] = s;
}

template <int U, int NTH>
__global__ __launch_bounds__(NTH) void scale_copy(const float4* __restrict__ p, float4* __restrict__ q, long long n4) {
  const long long stride = (long long)gridDim.x * NTH;
  long long i = (long long)blockIdx.x * NTH + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) q[i + u * stride] = make_float4(v[u].x * 1.5f + 1.f, v[u].y * 1.5f + 1.f, v[u].z * 1.5f + 1.f, v[u].w * 1.5f + 1.f);
  }
  for (; i < n4; i += stride) {
    const float4 v = p[i];
    q[i] = make_float4(v.x * 1.5f + 1.f, v.y * 1.5f + 1.f, v.z * 1.5f + 1.f, v.w * 1.5f + 1.f);
  }
}

template <typename F>
double timed(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) f();
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}

template <int U, int NTH>
void row(const float4* a, float4* b, float* out, long long n4) {
  for (int mult : {4, 8, 16, 32, 64}) {
    const int grid = 256 * mult;
    const double tr = timed([&] { read_sum<U, NTH><<<grid, NTH>>>(a, out, n4); });
    const double tc = timed([&] { scale_copy<U, NTH><<<grid, NTH>>>(a, b, n4); });
    printf("loads in flight %2d  block %4d  grid %5d (%2d per CU):  read %.3f ms = %.2f TB/s    read+write %.3f ms = %.2f TB/s\n", U, NTH, grid, mult, tr,
           n4 * 16.0 / tr * 1e-9, tc, 2 * n4 * 16.0 / tc * 1e-9);
  }
}

int main() {
  const long long n4 = 2LL * 32 * 48 * 256 * 128 / 4;  // the 403 MB tensor of the 3-D stage
  float4 *a, *b;
  float* out;
  (void)hipMalloc(&a, n4 * 16);
  (void)hipMalloc(&b, n4 * 16);
  (void)hipMalloc(&out, 16);
  (void)hipMemset(a, 0, n4 * 16);
  row<4, 256>(a, b, out, n4);
  row<8, 256>(a, b, out, n4);
  row<16, 256>(a, b, out, n4);
  row<4, 512>(a, b, out, n4);
  row<8, 512>(a, b, out, n4);
  row<2, 1024>(a, b, out, n4);
  row<4, 1024>(a, b, out, n4);
  return 0;
}
