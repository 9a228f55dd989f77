// Diagnostics for bench.py (not on the product path): what the matrix pipe of THIS device sustains, and at which clock.
//   pnr_probe_mfma_peak: a register-only v_mfma_f32_32x32x16_bf16 loop on every SIMD (8 waves per CU, 4 independent
//   accumulator chains per wave), either with constant operands or with pseudo-random operands that change from MFMA
//   to MFMA.  On MI355X the first sustains ~2.46 PFLOP/s at ~2.37 GHz, the second only ~1.83 PFLOP/s: with the toggle
//   rate of real data the chip lowers the shader clock to ~1.83 GHz (power).  The fused MLP's MFMA operands are real
//   activations and weights, so the second figure is the ceiling that applies to it; bench.py reports both next to the
//   datasheet peak.  s_memtime counts shader cycles, s_memrealtime a constant 100 MHz: their ratio is the clock.
#include <hip/hip_runtime.h>

#include "pnr_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <bool RANDOM>
__global__ __launch_bounds__(512) void k_mfma_peak(unsigned long long* out, int iters, float seed)
{
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    bf16x8 a[4], b[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            a[k][i] = RANDOM ? (__bf16)(((int)(h >> 9) & 0xffff) / 32768.0f - 1.0f) : (__bf16)(seed + i);
            h = h * 1664525u + 1013904223u;
            b[k][i] = RANDOM ? (__bf16)(((int)(h >> 9) & 0xffff) / 32768.0f - 1.0f) : (__bf16)(seed - i);
        }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k & 3], b[(k + 1) & 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 1) & 3], b[(k + 2) & 3], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 2) & 3], b[(k + 3) & 3], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(k + 3) & 3], b[k & 3], acc3, 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 12345.678f) out[2] = 1;
}

// tflops_out, mhz_out: host floats.  scratch: >= 32 bytes of device memory.  Synchronises the stream (diagnostic only).
PNR_EXPORT int pnr_probe_mfma_peak(int random_operands, int iters, void* scratch, float* tflops_out_host, float* mhz_out_host,
                                   void* stream)
{
    PNR_REQUIRE(iters >= 1 && scratch && tflops_out_host && mhz_out_host, "pnr_probe_mfma_peak: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 0;
    PNR_HIP(hipGetDevice(&dev));
    PNR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipEvent_t e0, e1;
    PNR_HIP(hipEventCreate(&e0));
    PNR_HIP(hipEventCreate(&e1));
    unsigned long long* out = (unsigned long long*)scratch;
    for (int rep = 0; rep < 2; ++rep) {          // first launch: warm-up (clock ramp, code load)
        PNR_HIP(hipEventRecord(e0, st));
        if (random_operands) hipLaunchKernelGGL(k_mfma_peak<true>, dim3(cus), dim3(512), 0, st, out, iters, 1.0f);
        else hipLaunchKernelGGL(k_mfma_peak<false>, dim3(cus), dim3(512), 0, st, out, iters, 1.0f);
        PNR_CHECK_LAUNCH("pnr_probe_mfma_peak");
        PNR_HIP(hipEventRecord(e1, st));
        PNR_HIP(hipEventSynchronize(e1));
    }
    float ms = 0.0f;
    PNR_HIP(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2] = {0, 1};
    PNR_HIP(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    *tflops_out_host = (float)((double)iters * 64.0 * 2.0 * 32 * 32 * 16 * 8 * cus / (ms * 1e-3) / 1e12);
    // block 0's first wave is the older wave of its SIMD: it owns the pipe and finishes in half the kernel time
    // (oldest-first MFMA arbitration), but the clock ratio of the two counters is what is wanted here
    *mhz_out_host = h[1] ? (float)(100.0 * (double)h[0] / (double)h[1]) : 0.0f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return PNR_OK;
}

// pnr_probe_raw_read: a pure read of a channel-major raw image in k_composite's own order (per wave: the 8 channel rows of
// a batch of one ray, N/4 lanes x 16 B each, 8 loads in flight, 8 waves per SIMD) -- what HBM delivers for this access
// pattern with no arithmetic at all.  bench.py quotes k_composite against it next to the 8 TB/s datasheet peak.
__global__ __launch_bounds__(256) void k_raw_read(const float* raw, int64_t sc, int64_t R, int N, int CH, float* sink)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const bool active = lane * 4 < N;
    float acc = 0.0f;
    for (int64_t ray = wave; ray < R; ray += n_waves) {
        const float* p = raw + ray * N + lane * 4;
        for (int c0 = 0; c0 < CH; c0 += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = (active && c0 + j < CH) ? *reinterpret_cast<const float4*>(p + (int64_t)(c0 + j) * sc) : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    }
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

// raw (n_channels, R*N) channel-major with channel stride raw_stride_c, N % 4 == 0, N <= 256.  gbs_out: host float.
PNR_EXPORT int pnr_probe_raw_read(const float* raw, int64_t raw_stride_c, int64_t n_rays, int n_samples, int n_channels,
                                  int iters, void* scratch, float* gbs_out_host, void* stream)
{
    PNR_REQUIRE(raw && scratch && gbs_out_host && iters >= 1 && n_rays >= 1, "pnr_probe_raw_read: bad arguments");
    PNR_REQUIRE(n_samples >= 4 && n_samples <= 256 && (n_samples % 4) == 0 && (raw_stride_c % 4) == 0, "pnr_probe_raw_read: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    PNR_HIP(hipEventCreate(&e0));
    PNR_HIP(hipEventCreate(&e1));
    const int grid = pnr_grid_cap((n_rays + 3) / 4, 8);
    hipLaunchKernelGGL(k_raw_read, dim3(grid), dim3(256), 0, st, raw, raw_stride_c, n_rays, n_samples, n_channels, (float*)scratch);
    PNR_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(k_raw_read, dim3(grid), dim3(256), 0, st, raw, raw_stride_c, n_rays, n_samples, n_channels, (float*)scratch);
    PNR_CHECK_LAUNCH("pnr_probe_raw_read");
    PNR_HIP(hipEventRecord(e1, st));
    PNR_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    PNR_HIP(hipEventElapsedTime(&ms, e0, e1));
    *gbs_out_host = (float)((double)n_rays * n_samples * 4.0 * n_channels * iters / (ms * 1e-3) / 1e9);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return PNR_OK;
}
