// Microbenchmark (diagnostic): how many cycles does ONE wave need per back-to-back v_mfma_f32_32x32x16_bf16 when it is alone on
// its SIMD, vs two waves sharing the SIMD?  (The ping-pong MLP runs one wave per SIMD in its M phase and measures 36.5-38 cycles
// per MFMA there; the pipe's own rate is 32.)  Register-only loop, 2 or 4 independent accumulator chains, constant operands,
// optionally one ds_read_b128 + counted wait per MFMA as in the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int CHAINS, bool LDS>
__global__ __launch_bounds__(512) void k_issue(unsigned long long* out, int iters)
{
    __shared__ __attribute__((aligned(16))) char smem[65536];
    f32x16 acc[4] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * i); b[i] = (__bf16)(0.02f * i); }
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (threadIdx.x & 63) * 16;
    u32x4 f[4];
    for (int k = 0; k < 4; ++k) f[k] = u32x4{1, 2, 3, 4};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (LDS) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[k & 3]));
            bf16x8 av = LDS ? __builtin_bit_cast(bf16x8, f[k & 3]) : a;
            acc[k % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b, acc[k % CHAINS], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (LDS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[(k + 3) & 3]) : "v"(addr), "n"(1024));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = c1 - c0;
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 123.4f) out[1] = 1;
}

int main()
{
    unsigned long long* out; hipMalloc(&out, 64);
    const int iters = 2000;
    for (int threads : {256, 512}) {          // 4 waves = one per SIMD; 8 waves = two per SIMD
        for (int v = 0; v < 4; ++v) {
            for (int rep = 0; rep < 2; ++rep) {
                if (v == 0) hipLaunchKernelGGL((k_issue<2, false>), dim3(256), dim3(threads), 0, 0, out, iters);
                if (v == 1) hipLaunchKernelGGL((k_issue<4, false>), dim3(256), dim3(threads), 0, 0, out, iters);
                if (v == 2) hipLaunchKernelGGL((k_issue<2, true>), dim3(256), dim3(threads), 0, 0, out, iters);
                if (v == 3) hipLaunchKernelGGL((k_issue<4, true>), dim3(256), dim3(threads), 0, 0, out, iters);
                hipDeviceSynchronize();
            }
            unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            const double per_wave = (double)h[0] / (iters * 32.0);
            printf("%d wave(s) per SIMD, %d accumulator chains, %s: %.1f cycles per MFMA per wave = %.1f per SIMD\n", threads / 256,
                   v & 1 ? 4 : 2, v >= 2 ? "ds_read_b128 + counted wait per MFMA" : "register operands only          ", per_wave, per_wave / (threads / 256));
        }
    }
    return 0;
}
