// runtime.hip -- error reporting / device check for libmpv_hip.so (no global mutable state
// other than the thread-local error string).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "mpv_common.h"
#include "mpv_kernels.h"

static thread_local char g_err[512] = "";

void mpv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mpv_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mpv_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return MPV_E_HIP;
  }
  return MPV_OK;
}

extern "C" const char* mpv_last_error(void) { return g_err; }
extern "C" int mpv_version(void) { return 100; }

extern "C" int mpv_check_device(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    mpv_set_error("mpv_check_device: no HIP device");
    return MPV_E_HIP;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    mpv_set_error("mpv_check_device: hipGetDeviceProperties failed");
    return MPV_E_HIP;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    mpv_set_error("mpv_check_device: kernels are built for gfx950 only, device is %s", prop.gcnArchName);
    return MPV_E_ARCH;
  }
  return MPV_OK;
}
