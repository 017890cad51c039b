// Error plumbing + version of the C ABI (no exceptions cross the boundary; text is thread-local).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mgdt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mgdt_last_error(void) { return g_err; }
extern "C" const char* mgdt_version(void) { return "mgdt-hip 0.1 (gfx950)"; }
