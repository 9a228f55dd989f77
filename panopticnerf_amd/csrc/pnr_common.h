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

// One 1 KiB LDS-DMA piece (device code): lane l copies the 16 bytes at src + 16 l to the LDS address dst + 16 l; src and dst are
// WAVE-UNIFORM.  Scalar-base form of the instruction -- an SGPR pair + this lane's constant 16 l as the 32-bit offset, the LDS
// target written to M0 by hand -- instead of what __builtin_amdgcn_global_load_lds makes of a per-lane pointer (a 64-bit VALU
// add per piece and a 64-bit address per lane): measured on the fused MLP launch, same box, outputs bit-identical:
// 11.50 -> 11.08 ms (-3.7 %, -7.7 % in cycles).  AUX: cache policy, 0 default, 1 sc0, 2 nt.  M0 is reserved in the
// backend (a clobber is rejected with a warning and ignored): tools/asm_lint.py
// -- run by `make all` on every object -- checks that these s_mov are the ONLY M0 accesses of the library.
// Also measured: the piece with NO vector register (buffer_load_dwordx4 off, srd, soffset lds through a resource with
// ADD_TID_ENABLE and stride 16): bit-identical outputs, same time as this form (11.36 vs 11.37 ms) -- not kept.
#ifdef __HIPCC__
template <int AUX = 0>
__device__ __forceinline__ void pnr_dma_piece(const void* src, void* dst, int lane16)
{
    // readfirstlane: a no-op on values the compiler already knows to be wave-uniform, and what keeps the "s" operands legal
    // where its divergence analysis cannot prove it (e.g. -DPNR_TRACE builds); the CALLER guarantees uniformity either way
    const uint32_t m0v = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)dst);
    const uint64_t sa = (uint64_t)(uintptr_t)src;
    src = (const void*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(sa >> 32)) << 32) |
                                   (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)sa));
#ifdef PNR_DEBUG_UNIFORM        /* debug builds: a caller that breaks wave-uniformity traps instead of copying lane 0's addresses */
    if ((uint64_t)(uintptr_t)src != sa || (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)dst != m0v) __builtin_trap();
#endif
    // no "m0" in the clobber list: hipcc answers it with "inline asm clobber list contains reserved registers: m0" (the backend
    // reserves M0 and ignores the declaration) -- the lint over EVERY object's assembly (tools/asm_lint.py, run by `make all`)
    // is the enforcement
    if constexpr (AUX == 2)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(lane16), "s"(src), "s"(m0v) : "memory");
    else if constexpr (AUX == 1)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc0" :: "v"(lane16), "s"(src), "s"(m0v) : "memory");
    else
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(lane16), "s"(src), "s"(m0v) : "memory");
}

#endif
