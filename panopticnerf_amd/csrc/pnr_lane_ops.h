// Cross-lane sums without LDS memory traffic (shared by the compositing kernels and the fused MLP + compositing epilogue).
#pragma once
#include <hip/hip_runtime.h>

// x + x[lane ^ D] without touching LDS where DPP can do it: quad_perm for D = 1, 2; row_half_mirror /
// row_mirror for D = 4, 8 (a lane is paired with a lane of the OTHER half of its 8 / 16 group, which is all
// a sum butterfly needs); ds_swizzle for D = 16; ds_bpermute (via __shfl_xor) only for D = 32.
template <int D>
__device__ __forceinline__ float xor_add(float x)
{
    const int xi = __float_as_int(x);
    int yi;
    if constexpr (D == 1) yi = __builtin_amdgcn_mov_dpp(xi, 0xB1, 0xF, 0xF, true);
    else if constexpr (D == 2) yi = __builtin_amdgcn_mov_dpp(xi, 0x4E, 0xF, 0xF, true);
    else if constexpr (D == 4) yi = __builtin_amdgcn_mov_dpp(xi, 0x141, 0xF, 0xF, true);
    else if constexpr (D == 8) yi = __builtin_amdgcn_mov_dpp(xi, 0x140, 0xF, 0xF, true);
    else if constexpr (D == 16) yi = __builtin_amdgcn_ds_swizzle(xi, 0x401F);
    else yi = __shfl_xor(xi, 32, 64);
    return x + __int_as_float(yi);
}

// max(x, x[lane ^ D]), same pairings.  Written as the instructions themselves: fmaxf() costs a canonicalising v_max per operand
// (IEEE sNaN quieting) and its v_mov_dpp is not folded into the v_max.  s_nop 1: a DPP operand written by the previous VALU
// instruction needs two wait states, and the compiler's hazard recogniser does not look inside an asm statement.
__device__ __forceinline__ float max_raw(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int D>
__device__ __forceinline__ float xor_max(float x)
{
    float r;
    if constexpr (D == 1) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r) : "v"(x));
    else if constexpr (D == 2) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r) : "v"(x));
    else if constexpr (D == 4) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r) : "v"(x));
    else if constexpr (D == 8) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(r) : "v"(x));
    else if constexpr (D == 16) {
        // v_permlane16_swap (gfx950) on two copies of x: one ends up with rows {0, 0, 2, 2}, the other with rows {1, 1, 3, 3} -- the
        // xor-16 exchange on the VALU.  ds_swizzle here put an LDS-crossbar round trip per register on the critical path: at 233
        // registers the compiler had one temporary for sixteen of them
        const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
        r = max_raw(__uint_as_float(p[0]), __uint_as_float(p[1]));
    } else r = max_raw(x, __int_as_float(__shfl_xor(__float_as_int(x), 32, 64)));
    return r;
}

// x + x[lane ^ 16] through v_permlane16_swap (see xor_max<16>)
__device__ __forceinline__ float xor16_add_swap(float x)
{
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}

// sum over the SUB-lane group of this lane (every lane of the group ends up with it)
template <int SUB>
__device__ __forceinline__ float group_sum(float x)
{
    if constexpr (SUB > 1) x = xor_add<1>(x);
    if constexpr (SUB > 2) x = xor_add<2>(x);
    if constexpr (SUB > 4) x = xor_add<4>(x);
    if constexpr (SUB > 8) x = xor_add<8>(x);
    if constexpr (SUB > 16) x = xor_add<16>(x);
    if constexpr (SUB > 32) x = xor_add<32>(x);
    return x;
}

template <int SUB>
__device__ __forceinline__ float group_max(float x)
{
    if constexpr (SUB > 1) x = xor_max<1>(x);
    if constexpr (SUB > 2) x = xor_max<2>(x);
    if constexpr (SUB > 4) x = xor_max<4>(x);
    if constexpr (SUB > 8) x = xor_max<8>(x);
    if constexpr (SUB > 16) x = xor_max<16>(x);
    if constexpr (SUB > 32) x = xor_max<32>(x);
    return x;
}

template <int SUB, int NB>
__device__ __forceinline__ void group_max_batch(float (&r)[NB])
{
#define PNR_STEP(D) if constexpr (SUB > D) { _Pragma("unroll") for (int j = 0; j < NB; ++j) r[j] = xor_max<D>(r[j]); }
    PNR_STEP(1) PNR_STEP(2) PNR_STEP(4) PNR_STEP(8) PNR_STEP(16) PNR_STEP(32)
#undef PNR_STEP
}

template <int SUB, int NB>
__device__ __forceinline__ void group_sum_batch(float (&r)[NB])
{
#define PNR_STEP(D) if constexpr (SUB > D) { _Pragma("unroll") for (int j = 0; j < NB; ++j) r[j] = xor_add<D>(r[j]); }
    PNR_STEP(1) PNR_STEP(2) PNR_STEP(4) PNR_STEP(8) PNR_STEP(16) PNR_STEP(32)
#undef PNR_STEP
}

// DPP move with an identity for lanes whose source is out of range (row_shr / row_shl / wave_shr ...): no LDS involved
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_or(float x, float identity)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(x), CTRL, ROW_MASK, 0xF, false));
}
