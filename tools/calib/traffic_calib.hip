// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, per access width (VERDICT r1 item 8;
// MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Streams a buffer far larger than the 256 MiB Infinity Cache with coalesced loads / stores of 4, 8 and 16 bytes per lane:
//   read_b32 / read_b64 / read_b128     N bytes read, 4 KiB written          (kernel name = access width)
//   write_b32 / write_b64 / write_b128  N bytes written, nothing read
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, tools/calib/run.sh); the ratio
// counter * 1024 / N per kernel is the factor profiles/traffic.json divides by.
//   hipcc --offload-arch=gfx950 -O3 -o traffic_calib traffic_calib.hip && ./traffic_calib [MiB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

template <typename T>
__global__ __launch_bounds__(256) void read_kernel(const T* __restrict__ src, float* __restrict__ sink, size_t n) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = src[i];
    const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) acc += f[k];
  }
  if (acc == 123.456f) sink[threadIdx.x] = acc;  // never true: keeps the loads alive without a store stream
}

template <typename T>
__global__ __launch_bounds__(256) void write_kernel(T* __restrict__ dst, size_t n, float v) {
  T t;
  float* f = reinterpret_cast<float*>(&t);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(T) / 4); ++k) f[k] = v + k;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = t;
}


int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2048;
  const size_t bytes = mib << 20;
  char* buf = nullptr;
  float* sink = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) {
    fprintf(stderr, "hipMalloc failed\n");
    return 1;
  }
  (void)hipMemset(buf, 0, bytes);
  (void)hipDeviceSynchronize();
  const int grid = 256 * 16;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_kernel<float>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float*>(buf), sink, bytes / 4);
    hipLaunchKernelGGL(read_kernel<float2>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float2*>(buf), sink, bytes / 8);
    hipLaunchKernelGGL(read_kernel<float4>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const float4*>(buf), sink, bytes / 16);
    hipLaunchKernelGGL(write_kernel<float>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<float*>(buf), bytes / 4, 1.f);
    hipLaunchKernelGGL(write_kernel<float2>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<float2*>(buf), bytes / 8, 2.f);
    hipLaunchKernelGGL(write_kernel<float4>, dim3(grid), dim3(256), 0, 0, reinterpret_cast<float4*>(buf), bytes / 16, 3.f);
  }
  if (hipDeviceSynchronize() != hipSuccess) {
    fprintf(stderr, "kernel failed\n");
    return 1;
  }
  printf("streamed %zu MiB per kernel, 3 repetitions\n", mib);
  return 0;
}
