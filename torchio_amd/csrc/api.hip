// api.hip — error reporting and introspection for libtio_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.hpp"

namespace tio {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

int fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return status;
}

// Launch errors only (hipGetLastError); never synchronises.
int check_launch(const char* what) {
  const hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(TIO_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(err));
  return TIO_OK;
}

}  // namespace tio

extern "C" int tio_abi_version(void) { return TIO_ABI_VERSION; }

extern "C" const char* tio_last_error(void) { return tio::g_error; }

extern "C" int tio_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
