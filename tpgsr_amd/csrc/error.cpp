// thread-local last-error string for the C ABI
#include <stdarg.h>
#include <stdio.h>
#include "../../include/tpgsr_hip.h"

static thread_local char g_err[512] = "";

void tpgsr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* tpgsr_last_error(void) { return g_err; }
extern "C" int tpgsr_version(void) { return 1; }
