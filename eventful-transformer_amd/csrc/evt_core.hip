// evt_core.hip -- version / error plumbing of libevt_hip.so.
#include "evt_common.h"

static thread_local char g_err[512] = "";

char* evt_err_buf() { return g_err; }

int evt_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int evt_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return evt_fail(EVT_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return EVT_OK;
}

extern "C" int evt_version(void) { return EVT_ABI_VERSION; }
extern "C" const char* evt_last_error_string(void) { return g_err; }
extern "C" const char* evt_target_arch(void) { return "gfx950"; }
