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
