// Library-level entry points: version, error string, device check.
#include "lcv_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void lcv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int lcv_version(void) { return 1; }

extern "C" const char* lcv_last_error(void) { return g_err; }

extern "C" int lcv_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    lcv_set_error("device_check: no HIP device");
    return LCV_EDEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    lcv_set_error("device_check: hipGetDeviceProperties failed");
    return LCV_EDEVICE;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    lcv_set_error("device_check: device is %s, this library is built for gfx950 only", prop.gcnArchName);
    return LCV_EDEVICE;
  }
  return LCV_OK;
}
