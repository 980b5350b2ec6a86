// Version / error plumbing of libmi355conv (host only).
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mi355conv.h"

static thread_local char g_err[512] = "";

void mi355_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int mi355_version(void) { return 100; }
extern "C" const char* mi355_last_error(void) { return g_err; }
