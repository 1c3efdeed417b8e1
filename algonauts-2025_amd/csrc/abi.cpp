// Error plumbing and version of the C ABI (include/tribe_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/tribe_hip.h"

static thread_local char g_err[512] = "";

void tribe_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int tribe_version(void) { return TRIBE_ABI_VERSION; }
extern "C" const char* tribe_last_error(void) { return g_err; }
