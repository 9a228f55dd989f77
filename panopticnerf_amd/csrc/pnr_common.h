// Shared helpers for libpnr.so (gfx950 only).  See include/pnr.h for the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "pnr.h"

#define PNR_EXPORT extern "C" __attribute__((visibility("default")))

void pnr_set_error(const char* fmt, ...);

#define PNR_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            pnr_set_error(__VA_ARGS__);   \
            return PNR_EINVAL;            \
        }                                 \
    } while (0)

#define PNR_CHECK_LAUNCH(name)                                                    \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            pnr_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return PNR_EHIP;                                                      \
        }                                                                         \
    } while (0)

#define PNR_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e__ = (call);                                                   \
        if (e__ != hipSuccess) {                                                   \
            pnr_set_error("%s failed: %s", #call, hipGetErrorString(e__));         \
            return PNR_EHIP;                                                       \
        }                                                                          \
    } while (0)

// Compute units of the CURRENT device (256 on a full MI355X; partitioned / harvested parts report fewer),
// queried once per device and cached (pnr_api.cpp).
int pnr_cu_count(void);

// Memory-bound grids are capped at 8 workgroups of 256 threads per CU and grid-stride over
// the rest (cdna_hip_programming.md Guideline 11).
static inline int pnr_grid_cap(int64_t wanted, int per_cu = 8)
{
    const int64_t cap = pnr_cu_count() * (int64_t)per_cu;
    if (wanted < 1) wanted = 1;
    return (int)(wanted < cap ? wanted : cap);
}
