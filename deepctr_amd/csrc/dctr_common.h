// Shared host-side helpers for the libdctr_hip.so translation units (gfx950 only; no CUDA paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dctr.h"

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
