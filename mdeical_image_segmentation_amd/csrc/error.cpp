// Thread-local error string of the C ABI (include/misamd.h: mis_last_error).
#include <stdarg.h>
#include <stdio.h>

#include "misamd.h"

static thread_local char g_err[512] = "";

void mis_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mis_last_error(void) { return g_err; }
extern "C" int mis_version(void) { return 1; }
