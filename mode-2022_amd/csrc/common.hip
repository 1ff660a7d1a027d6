// Error string + ABI version for libmode_hip.so.
#include "common.h"

namespace mode {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

}  // namespace mode

extern "C" int mode_hip_abi_version(void) { return MODE_HIP_ABI_VERSION; }

extern "C" const char* mode_last_error(void) { return mode::g_err; }
