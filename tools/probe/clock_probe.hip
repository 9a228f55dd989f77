// Diagnostic: effective shader clock under load.  s_memtime counts shader cycles, s_memrealtime a constant 100 MHz.
// (a) idle-ish kernel (one wave per CU spinning on VALU), (b) every SIMD saturated with bf16 MFMAs (2 waves/SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE>
__global__ __launch_bounds__(512) void k_clock(unsigned long long* out, int iters, float seed)
{
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    bf16x8 a, b, a2[4], b2[4];
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    // MODE 2: operands that CHANGE from MFMA to MFMA with pseudo-random bf16 contents in [-1, 1) -- the toggle rate of
    // real activations and weights (constant operands draw less power and clock higher)
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u; a2[k][i] = (__bf16)(((int)(h >> 9) & 0xffff) / 32768.0f - 1.0f);
            h = h * 1664525u + 1013904223u; b2[k][i] = (__bf16)(((int)(h >> 9) & 0xffff) / 32768.0f - 1.0f);
        }
    float x = seed;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[k & 3], b2[(k + 1) & 3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[(k + 1) & 3], b2[(k + 2) & 3], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[(k + 2) & 3], b2[(k + 3) & 3], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[(k + 3) & 3], b2[k & 3], acc3, 0, 0, 0);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 64; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] + x == 12345.678f) out[0] = 0;
}

int main()
{
    unsigned long long* out; hipMalloc(&out, 256 * 16);
    unsigned long long h[512];
    for (int rep = 0; rep < 2; ++rep) {
        for (int mode = 0; mode < 3; ++mode) {
            const int iters = mode ? 60000 : 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 2) hipLaunchKernelGGL(k_clock<2>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
            else if (mode) hipLaunchKernelGGL(k_clock<1>, dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
            else hipLaunchKernelGGL(k_clock<0>, dim3(256), dim3(64), 0, 0, out, iters, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
            const double mhz = 100.0 * (double)h[0] / (double)h[1];
            const double tf = mode ? (double)iters * 64 * 2.0 * 32 * 32 * 16 * 8 * 256 / (ms * 1e-3) / 1e12 : 0.0;
            printf("%s: %.2f ms, s_memtime/s_memrealtime -> %.0f MHz%s", mode == 2 ? "MFMA-saturated, RANDOM changing operands" : mode ? "MFMA-saturated, constant operands      " : "light VALU (1 wave/CU)                 ", ms, mhz, mode ? "" : "\n");
            if (mode) printf(", %.0f TFLOP/s = %.1f %% of 2.5 PF; cycles per MFMA per SIMD %.2f\n", tf, tf / 25.0, (double)h[0] / ((double)iters * 64 * 2));
        }
    }
    return 0;
}
