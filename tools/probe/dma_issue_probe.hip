// Microbenchmark (diagnostic): what does ISSUING an LDS-DMA piece (global_load_lds, 1 KiB per wave instruction) cost the
// issuing wave, and does the piece count in lgkmcnt?  One workgroup per CU, `waves` waves, every wave issues P pieces of
// an L2-resident image, then (a) reads s_memtime right away (the read itself is an lgkm op -> waits lgkmcnt(0)),
// (b) after a ~2000-cycle dependent VALU chain that needs no memory.  vmcnt(0) afterwards gives the completion time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void;

template <int P, int CHAIN>
__global__ __launch_bounds__(512) void k_probe(const uint8_t* img, unsigned long long* out, float seed)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    float x = seed + lane;
    unsigned long long best[3] = {~0ull, ~0ull, ~0ull};
    for (int it = 0; it < 20; ++it) {
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int f = (it * 64 + p * nw + wave) & 1023;
            __builtin_amdgcn_global_load_lds((const void*)(img + (size_t)f * 1024 + lane * 16), (lds_void*)(smem + (p * nw + wave) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < CHAIN; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (it >= 4) {
            if (t1 - t0 < best[0]) best[0] = t1 - t0;
            if (t2 - t0 < best[1]) best[1] = t2 - t0;
        }
    }
    if (blockIdx.x == 0 && lane == 0) { out[wave * 2] = best[0]; out[wave * 2 + 1] = best[1]; }
    if (x == 12345.f) out[63] = 1;
}

template <int P, int CHAIN>
static void run(const uint8_t* img, unsigned long long* out, int waves)
{
    hipMemset(out, 0, 64 * 8);
    hipLaunchKernelGGL((k_probe<P, CHAIN>), dim3(256), dim3(64 * waves), 64 * 1024 * 2, 0, img, out, 1.0f);
    unsigned long long h[64];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("waves=%d pieces/wave=%d valu_chain=%4d : wave0 issue+chain->memtime %6llu cyc, ->vmcnt(0) %6llu | last wave %6llu, %6llu\n",
           waves, P, CHAIN, h[0], h[1], h[(waves - 1) * 2], h[(waves - 1) * 2 + 1]);
}

int main()
{
    uint8_t* img; unsigned long long* out;
    hipMalloc(&img, 1024 * 1024); hipMalloc(&out, 64 * 8);
    hipMemset(img, 1, 1024 * 1024);
    hipFuncSetAttribute((const void*)k_probe<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
#define RUN(P, C) hipFuncSetAttribute((const void*)k_probe<P, C>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); run<P, C>(img, out, 1); run<P, C>(img, out, 4); run<P, C>(img, out, 8);
    RUN(0, 0) RUN(0, 512) RUN(1, 0) RUN(1, 512) RUN(4, 0) RUN(4, 512) RUN(8, 0) RUN(8, 512)
    return 0;
}
