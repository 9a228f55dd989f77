// Microbenchmark (diagnostic): per-CU rate of streaming an L2-resident 1.5 MB image into LDS,
// (A) by LDS-DMA (global_load_lds 16 B/lane), (B) by global_load_dwordx4 -> VGPR -> ds_write_b128,
// (C) global_load_dwordx4 only.  256 workgroups x 512 threads (one per CU), like k_mlp_fused.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_stream(const uint8_t* img, int n_frags, int chunk_frags, int iters, uint32_t* sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4 acc = {0, 0, 0, 0};
    int slot = 0;
    for (int it = 0; it < iters; ++it) {
        for (int c0 = 0; c0 + chunk_frags <= n_frags; c0 += chunk_frags) {
            char* dst = smem + slot * chunk_frags * 1024;
            const uint8_t* src = img + (size_t)c0 * 1024 + lane * 16;
            for (int f = wave; f < chunk_frags; f += 8) {
                if (MODE == 0) {
                    __builtin_amdgcn_global_load_lds((const void*)(src + (size_t)f * 1024), (lds_void*)(dst + f * 1024), 16, 0, 0);
                } else {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(src + (size_t)f * 1024);
                    if (MODE == 1) *reinterpret_cast<u32x4*>(dst + f * 1024 + lane * 16) = v;
                    else acc ^= v;
                }
            }
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (MODE != 2) acc ^= *reinterpret_cast<const u32x4*>(dst + ((lane * 7 + wave) & 1023) * 16);
            slot ^= 1;
        }
    }
    if (acc[0] == 0x12345678u) sink[threadIdx.x] = acc[1] ^ acc[2] ^ acc[3];
}

int main()
{
    const int n_frags = 1408, chunk = 32, iters = 200;
    uint8_t* img; uint32_t* sink;
    hipMalloc(&img, (size_t)n_frags * 1024); hipMalloc(&sink, 4096);
    hipMemset(img, 1, (size_t)n_frags * 1024);
    const size_t lds = 2 * chunk * 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"LDS-DMA (global_load_lds)", "global_load -> ds_write_b128", "global_load only"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k_stream<0>, dim3(256), dim3(512), lds, 0, img, n_frags, chunk, iters, sink);
            if (mode == 1) hipLaunchKernelGGL(k_stream<1>, dim3(256), dim3(512), lds, 0, img, n_frags, chunk, iters, sink);
            if (mode == 2) hipLaunchKernelGGL(k_stream<2>, dim3(256), dim3(512), lds, 0, img, n_frags, chunk, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)(n_frags / chunk) * chunk * 1024 * iters;
            if (rep) printf("%-32s %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip  (%.0f ns per 32 KiB chunk)\n", names[mode], ms,
                            bytes / ms / 1e6, bytes * 256 / ms / 1e9, ms * 1e6 / ((n_frags / chunk) * iters));
        }
    }
    return 0;
}
