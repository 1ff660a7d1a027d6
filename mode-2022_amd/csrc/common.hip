// Error string + ABI version for libmode_hip.so.
#include "common.h"

namespace mode {

static thread_local char g_err[512] = "";
static thread_local int g_pack_reuse = 0;

bool pack_needed() { return g_pack_reuse == 0; }

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
__global__ __launch_bounds__(256) void fill_words_kernel(unsigned* __restrict__ dst, unsigned pattern, size_t nwords) {
  // 16-byte stores over the aligned middle, single words at the two ends
  const size_t head = ((16 - ((size_t)dst & 15)) & 15) / 4;
  const size_t h = head < nwords ? head : nwords;
  const size_t n4 = (nwords - h) / 4;
  uint4* mid = reinterpret_cast<uint4*>(dst + h);
  const uint4 v = make_uint4(pattern, pattern, pattern, pattern);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) mid[i] = v;
  if (blockIdx.x == 0) {
    if (threadIdx.x < h) dst[threadIdx.x] = pattern;
    const size_t tail0 = h + 4 * n4;
    if (tail0 + threadIdx.x < nwords && threadIdx.x < 4) dst[tail0 + threadIdx.x] = pattern;
  }
}
}  // namespace

int fill_words(void* dst, unsigned pattern, size_t nwords, hipStream_t st, const char* what) {
  if (nwords == 0) return MODE_OK;
  MODE_REQUIRE(dst && ((size_t)dst & 3) == 0, MODE_ERR_BAD_ARG, "%s: fill target must be a 4-byte aligned device pointer", what);
  const size_t blocks = (nwords / 4 + 255) / 256;
  const int grid = (int)(blocks < 1 ? 1 : (blocks > (size_t)16 * 256 ? (size_t)16 * 256 : blocks));
  hipLaunchKernelGGL(fill_words_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<unsigned*>(dst), pattern, nwords);
  return check_launch(what);
}

}  // namespace mode

extern "C" int mode_hip_abi_version(void) { return MODE_HIP_ABI_VERSION; }

extern "C" const char* mode_last_error(void) { return mode::g_err; }

extern "C" int mode_weight_pack_reuse(int on) {
  const int prev = mode::g_pack_reuse;
  mode::g_pack_reuse = on ? 1 : 0;
  return prev;
}
