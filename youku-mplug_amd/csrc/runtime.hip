// runtime.hip -- error reporting / device check for libmpv_hip.so (no global mutable state
// other than the thread-local error string).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "mpv_common.h"
#include "../../include/mpv.h"

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
  return mpv_check_arch_name(prop.gcnArchName);
}

// the comparison mpv_check_device makes, on a caller-supplied gcnArchName ("gfx950:sramecc+:xnack-"): lets a host decide before it
// touches a device, and lets the error path be exercised on a gfx950 box
extern "C" int mpv_check_arch_name(const char* gcn_arch_name) {
  if (!gcn_arch_name) {
    mpv_set_error("mpv_check_arch_name: null name");
    return MPV_E_ARG;
  }
  if (strncmp(gcn_arch_name, "gfx950", 6) != 0 || (gcn_arch_name[6] != '\0' && gcn_arch_name[6] != ':')) {
    mpv_set_error("mpv_check_device: kernels are built for gfx950 only, device is %s", gcn_arch_name);
    return MPV_E_ARCH;
  }
  return MPV_OK;
}
