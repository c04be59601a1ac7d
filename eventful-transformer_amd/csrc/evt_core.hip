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

static thread_local char g_note[256] = "";

void evt_note_launch_problem(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_note, sizeof(g_note), fmt, ap);
  va_end(ap);
}

int evt_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char note[256];
    snprintf(note, sizeof(note), "%s", g_note);
    g_note[0] = 0;
    return note[0] ? evt_fail(EVT_ERR_HIP, "%s: %s (%s)", what, hipGetErrorString(e), note)
                   : evt_fail(EVT_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  }
  g_note[0] = 0;
  return EVT_OK;
}

extern "C" int evt_version(void) { return EVT_ABI_VERSION; }
extern "C" const char* evt_last_error_string(void) { return g_err; }
extern "C" const char* evt_target_arch(void) { return "gfx950"; }
