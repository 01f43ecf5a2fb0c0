// Error plumbing + version of the C ABI (include/flair_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void flair_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* flair_last_error(void) { return g_err; }
extern "C" int flair_abi_version(void) { return 3; }
