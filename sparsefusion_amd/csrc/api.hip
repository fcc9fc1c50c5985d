// Library-wide pieces of the C ABI: version and per-thread error text.
#include <stdarg.h>

#include "sf_common.h"

namespace {
thread_local char g_err[512] = "";
thread_local char g_kernel[160] = "";
}

void sf_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

void sf_note_kernel(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}

extern "C" const char *sf_last_conv_kernel(void) { return g_kernel; }
extern "C" int sf_version(void) { return 100; }  // 0.1.0
extern "C" const char *sf_last_error(void) { return g_err; }
