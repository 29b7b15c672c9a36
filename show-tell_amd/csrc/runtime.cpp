// Error channel and version of libshowtell_hip.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void st_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* st_last_error(void) { return g_err; }
extern "C" int st_version(void) { return 1; }
