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
// dctr_profile_next_launch(): the next launch of this host thread is timed (start/stop event pair of that dispatch).
// dctr_profile_arm(n): the next n launches are, into a pool of event pairs — for launches that are in flight TOGETHER
// (several streams), whose durations differ from an isolated launch's; dctr_profile_collect() waits and reads them.
constexpr int PROBE_POOL = 256;
static thread_local int g_armed = 0;         // launches still to be timed
static thread_local int g_taken = 0;         // pairs handed out since the last arm
static thread_local hipEvent_t g_e0[PROBE_POOL] = {}, g_e1[PROBE_POOL] = {};

bool dctr_profile_take(hipEvent_t* start, hipEvent_t* stop) {
    if (g_armed <= 0 || g_taken >= PROBE_POOL) return false;
    const int i = g_taken;
    if (g_e0[i] == nullptr) {
        if (hipEventCreate(&g_e0[i]) != hipSuccess || hipEventCreate(&g_e1[i]) != hipSuccess) return false;
    }
    *start = g_e0[i];
    *stop = g_e1[i];
    --g_armed;
    ++g_taken;
    return true;
}

extern "C" int dctr_profile_arm(int32_t n) {
    DCTR_REQUIRE(n >= 0 && n <= PROBE_POOL, DCTR_E_DIM, "profile_arm: n outside [0, %d]", PROBE_POOL);
    g_armed = n;
    g_taken = 0;
    return DCTR_OK;
}

extern "C" int dctr_profile_next_launch(void) { return dctr_profile_arm(1); }

extern "C" int dctr_profile_collect(float* ms, int32_t n) {
    DCTR_REQUIRE(ms != nullptr && n >= 0, DCTR_E_NULL, "profile_collect: null / negative");
    const int got = g_taken < n ? g_taken : n;
    for (int i = 0; i < got; ++i) {
        ms[i] = -1.f;
        if (hipEventSynchronize(g_e1[i]) != hipSuccess) continue;
        if (hipEventElapsedTime(&ms[i], g_e0[i], g_e1[i]) != hipSuccess) ms[i] = -1.f;
    }
    g_armed = 0;
    g_taken = 0;
    return got;
}

extern "C" float dctr_profile_last_ms(void) {
    float ms = -1.f;
    return dctr_profile_collect(&ms, 1) == 1 ? ms : -1.f;
}

extern "C" int dctr_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return -1;
    return khz;
}
