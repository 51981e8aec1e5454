// ABI bookkeeping: version, target arch, thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "dctr_common.h"

static thread_local char g_err[512] = "";

void dctr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int dctr_abi_version(void) { return DCTR_ABI_VERSION; }
extern "C" const char* dctr_last_error(void) { return g_err; }
extern "C" const char* dctr_target_arch(void) { return "gfx950"; }
