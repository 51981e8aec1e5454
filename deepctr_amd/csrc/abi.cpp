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

// ---- kernel-duration probe ------------------------------------------------------------------------
static thread_local bool g_armed = false;
static thread_local hipEvent_t g_e0 = nullptr, g_e1 = nullptr;
static thread_local bool g_pending = false;

bool dctr_profile_take(hipEvent_t* start, hipEvent_t* stop) {
    if (!g_armed) return false;
    g_armed = false;
    if (g_e0 == nullptr) {
        if (hipEventCreate(&g_e0) != hipSuccess || hipEventCreate(&g_e1) != hipSuccess) return false;
    }
    *start = g_e0;
    *stop = g_e1;
    g_pending = true;
    return true;
}

extern "C" int dctr_profile_next_launch(void) {
    g_armed = true;
    return DCTR_OK;
}

extern "C" float dctr_profile_last_ms(void) {
    if (!g_pending) return -1.f;
    g_pending = false;
    if (hipEventSynchronize(g_e1) != hipSuccess) return -1.f;
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, g_e0, g_e1) != hipSuccess) return -1.f;
    return ms;
}
