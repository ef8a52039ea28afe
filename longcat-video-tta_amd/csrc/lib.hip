// Library-level entry points: version, error string, device check.
#include "lcv_common.h"
#include <mutex>
#include <stdlib.h>
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

// ---- A/B knobs: one table, read once (include/lcv_hip.h documents every entry) ----
namespace {
struct Knob { const char* name; char value[32]; bool set; };
Knob g_knobs[] = {
    {"LCV_GEMM_TILE"}, {"LCV_GEMM_GROUP_M"}, {"LCV_GEMM_FAST_EPI"}, {"LCV_GEMM_SPLITK_TAIL"},
    {"LCV_CONV_8P"}, {"LCV_CONV_N192"}, {"LCV_CONV_ROWS"}, {"LCV_CONV_ROWS_GRID"}, {"LCV_CONV_ROWS_ORDER"},
    {"LCV_ATTN_FWD_W64"}, {"LCV_ATTN_XCD"},
    {"LCV_ATTN_BWD_VAR"}, {"LCV_ATTN_BWD_DKV_WAVES"}, {"LCV_ATTN_BWD_DQ_WAVES"}, {"LCV_ATTN_BWD_PIPE"}, {"LCV_ATTN_BWD_STAGGER"},
    {"LCV_ATTN_BWD_XCD"},
};
std::mutex g_knob_mutex;
bool g_knobs_read = false;
void knobs_read_locked() {
  for (Knob& k : g_knobs) {
    const char* e = getenv(k.name);
    k.set = e != nullptr;
    k.value[0] = 0;
    if (e) { strncpy(k.value, e, sizeof(k.value) - 1); k.value[sizeof(k.value) - 1] = 0; }
  }
  g_knobs_read = true;
}
}  // namespace

const char* lcv_knob(const char* name) {
  std::lock_guard<std::mutex> g(g_knob_mutex);
  if (!g_knobs_read) knobs_read_locked();
  for (const Knob& k : g_knobs)
    if (strcmp(k.name, name) == 0) return k.set ? k.value : nullptr;
  return nullptr;   // not a knob of this library
}

extern "C" int lcv_knobs_reload(void) {
  std::lock_guard<std::mutex> g(g_knob_mutex);
  knobs_read_locked();
  int n = 0;
  for (const Knob& k : g_knobs) n += k.set ? 1 : 0;
  return n;           // how many knobs are set (>= 0)
}

extern "C" const char* lcv_knobs_list(void) {
  static char buf[1024];
  static std::once_flag once;
  std::call_once(once, [] {
    buf[0] = 0;
    for (const Knob& k : g_knobs) { strcat(buf, k.name); strcat(buf, " "); }
  });
  return buf;
}
