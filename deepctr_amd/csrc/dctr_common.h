// Shared host-side helpers for the libdctr_hip.so translation units (gfx950 only; no CUDA paths).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "dctr.h"
#include <stdlib.h>

// A/B switches of the lab scripts (scripts/gemm_lab.sh, opt_lab.py, bench_train.py): read from the process environment ONLY in a
// lab build (python -m deepctr_amd.build with DCTR_BUILD_LAB=1 -> -DDCTR_LAB).  The shipped library ignores the environment: its
// kernel choice, workspace sizing and summation order do not depend on ambient state.
static inline const char* dctr_lab_env(const char* name) {
#ifdef DCTR_LAB
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

void dctr_set_error(const char* fmt, ...);

#define DCTR_REQUIRE(cond, code, ...)  \
    do {                               \
        if (!(cond)) {                 \
            dctr_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

static inline int dctr_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        dctr_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return DCTR_OK;
}

static inline bool dctr_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

static inline int64_t dctr_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

#define DCTR_WAVE 64

// Per-DEVICE host caches (a process may drive several GPUs from one thread: engine.on_model_device switches the current
// device per model).  dctr_cur_device(): the current device clamped into the cache arrays.
#define DCTR_MAX_DEVICES 32
static inline int dctr_cur_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < DCTR_MAX_DEVICES ? dev : DCTR_MAX_DEVICES - 1;
}
// compute units of the current device (256 on an MI355X)
static inline int dctr_n_cus() {
    static int n[DCTR_MAX_DEVICES] = {0};
    const int dev = dctr_cur_device();
    if (n[dev] == 0) {
        hipDeviceProp_t prop;
        int v = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) v = prop.multiProcessorCount;
        n[dev] = v > 0 ? v : 256;
    }
    return n[dev];
}
// Raises a kernel's dynamic-LDS limit once per (kernel instantiation, device, size): `granted` is the caller's
// static thread_local size_t[DCTR_MAX_DEVICES] (the attribute call costs ~10 us).
static inline hipError_t dctr_grant_lds(const void* fn, size_t bytes, size_t* granted) {
    const int dev = dctr_cur_device();
    if (bytes <= granted[dev]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) granted[dev] = bytes;
    return e;
}

// Kernel-duration probe (bench / tests): when armed by dctr_profile_next_launch(), the NEXT launch on this
// host thread goes out through hipExtLaunchKernelGGL with a start/stop event pair, i.e. the GPU timestamps
// of that dispatch alone (what rocprofv3 --kernel-trace reports), not an event-to-event period that also
// contains launch gaps.  dctr_profile_last_ms() waits for the stop event and returns the duration.
bool dctr_profile_take(hipEvent_t* start, hipEvent_t* stop);

#define DCTR_LAUNCH(kernel, grid, block, lds, stream, ...)                                             \
    do {                                                                                               \
        hipEvent_t dctr_e0_, dctr_e1_;                                                                 \
        if (dctr_profile_take(&dctr_e0_, &dctr_e1_))                                                   \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, dctr_e0_, dctr_e1_, 0, __VA_ARGS__); \
        else                                                                                           \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                         \
    } while (0)
