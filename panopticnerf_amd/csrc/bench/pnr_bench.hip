// libpnr_bench.so -- measurement helpers for bench.py and tools/ ONLY (include/pnr_bench.h).  Not part of the product: libpnr.so
// never allocates and never synchronises; everything here times launches with hipEvents and therefore SYNCHRONISES its stream.
// The product entry points it times are not linked: pnrb_bind() receives their addresses from the caller, which has loaded
// libpnr.so (or an A/B build of it) itself -- the timed code is exactly the library the caller uses.
//   pnrb_time_mlp_forward        iters x pnr_mlp_forward between two events on `stream`, + mean shader clock of the last launch
//   pnrb_time_mlp_forward_tiles  the same for pnr_mlp_forward_tiles (the fused inference MLP launch, without the combine kernel)
//   pnrb_probe_mfma_peak         what the matrix pipe of THIS device sustains, and at which clock (see below)
//   pnrb_probe_raw_read          what HBM delivers for k_composite's own access pattern with no arithmetic
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "pnr.h"
#include "pnr_bench.h"

#define PNRB_EXPORT extern "C" __attribute__((visibility("default")))
static thread_local char g_err[512] = "";
static void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
PNRB_EXPORT const char* pnrb_last_error(void) { return g_err; }
// (also for other translation units of the bench library)
void pnrb_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
#define PNR_EXPORT PNRB_EXPORT
#define PNR_REQUIRE(cond, ...) do { if (!(cond)) { set_error(__VA_ARGS__); return PNR_EINVAL; } } while (0)
#define PNR_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e__)); return PNR_EHIP; } } while (0)
#define PNR_CHECK_LAUNCH(name) do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) { set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); return PNR_EHIP; } } while (0)
static int cu_count(void)
{
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    return n;
}
static inline int pnr_grid_cap(int64_t wanted, int per_cu = 8)
{
    const int64_t cap = cu_count() * (int64_t)per_cu;
    if (wanted < 1) wanted = 1;
    return (int)(wanted < cap ? wanted : cap);
}

// ---- the product entry points under test (addresses from the caller)
typedef int (*fn_mlp_forward)(const pnr_mlp_desc*, const void*, const float*, const float*, int64_t, int, float*, int64_t, int64_t, void*);
typedef int (*fn_mlp_forward_tiles)(const pnr_mlp_desc*, const void*, const float*, const float*, int64_t, int, void*, void*);
static fn_mlp_forward g_fwd = nullptr;
static fn_mlp_forward_tiles g_tiles = nullptr;

PNRB_EXPORT int pnrb_bind(void* mlp_forward, void* mlp_forward_tiles)
{
    PNR_REQUIRE(mlp_forward && mlp_forward_tiles, "pnrb_bind: null entry point");
    g_fwd = (fn_mlp_forward)mlp_forward;
    g_tiles = (fn_mlp_forward_tiles)mlp_forward_tiles;
    return PNR_OK;
}
// the caller's descriptor with the clock probe armed (pnr_mlp_desc.clk_probe: {shader cycles, 100 MHz ticks} -> scratch)
static pnr_mlp_desc with_probe(const pnr_mlp_desc* desc, void* scratch)
{
    pnr_mlp_desc d = *desc;
    d.clk_probe[0] = (int32_t)(uint32_t)((uintptr_t)scratch & 0xffffffffu);
    d.clk_probe[1] = (int32_t)(uint32_t)((uint64_t)(uintptr_t)scratch >> 32);
    return d;
}

// iters launches of `launch` between two events on `st`; the clock probe (s_memtime / s_memrealtime of workgroup 0's first wave,
// written by the kernel to `scratch`, >= 16 device bytes: armed by the caller through with_probe) is read after the last launch
template <class F>
static int time_launches(F&& launch, int iters, void* scratch, float* ms_out, float* mhz_out, hipStream_t st)
{
    PNR_REQUIRE(g_fwd, "libpnr_bench: pnrb_bind was not called");
    PNR_REQUIRE(iters >= 1 && ms_out && mhz_out && scratch, "pnrb_time_*: bad arguments");
    hipEvent_t e0 = nullptr, e1 = nullptr;
    PNR_HIP(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); set_error("hipEventCreate failed"); return PNR_EHIP; }
    int rc = PNR_OK;
    hipError_t he = hipEventRecord(e0, st);
    for (int i = 0; i < iters && rc == PNR_OK && he == hipSuccess; ++i) rc = launch();
    float ms = 0.0f;
    unsigned long long h[2] = {0, 1};
    if (rc == PNR_OK && he == hipSuccess) he = hipEventRecord(e1, st);
    if (rc == PNR_OK && he == hipSuccess) he = hipEventSynchronize(e1);
    if (rc == PNR_OK && he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    if (rc == PNR_OK && he == hipSuccess) he = hipMemcpy(h, scratch, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipEventDestroy(e0);               // on every path
    (void)hipEventDestroy(e1);
    if (rc != PNR_OK) { set_error("the timed entry point failed (code %d): see pnr_last_error()", rc); return rc; }
    if (he != hipSuccess) { set_error("HIP call failed while timing: %s", hipGetErrorString(he)); return PNR_EHIP; }
    *ms_out = ms / (float)iters;
    *mhz_out = h[1] ? (float)(100.0 * (double)h[0] / (double)h[1]) : 0.0f;
    return PNR_OK;
}

PNRB_EXPORT int pnrb_time_mlp_forward(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z, int64_t n_rays,
                                      int n_samples, float* raw, int64_t raw_stride_s, int64_t raw_stride_c, int iters, void* scratch,
                                      float* ms_out_host, float* mhz_out_host, void* stream)
{
    PNR_REQUIRE(desc && scratch, "pnrb_time_mlp_forward: null desc / scratch");
    const pnr_mlp_desc d = with_probe(desc, scratch);
    return time_launches([&]() { return g_fwd(&d, packed, rays, z, n_rays, n_samples, raw, raw_stride_s, raw_stride_c, stream); },
                         iters, scratch, ms_out_host, mhz_out_host, (hipStream_t)stream);
}

PNRB_EXPORT int pnrb_time_mlp_forward_tiles(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z,
                                            int64_t n_rays, int n_samples, void* workspace, int iters, void* scratch,
                                            float* ms_out_host, float* mhz_out_host, void* stream)
{
    PNR_REQUIRE(desc && scratch, "pnrb_time_mlp_forward_tiles: null desc / scratch");
    const pnr_mlp_desc d = with_probe(desc, scratch);
    return time_launches([&]() { return g_tiles(&d, packed, rays, z, n_rays, n_samples, workspace, stream); },
                         iters, scratch, ms_out_host, mhz_out_host, (hipStream_t)stream);
}

// What the matrix pipe of THIS device sustains, and at which clock:
//   pnrb_probe_mfma_peak: a register-only v_mfma_f32_32x32x16_bf16 loop on every SIMD (8 waves per CU, 4 independent
//   accumulator chains per wave), either with constant operands or with pseudo-random operands that change from MFMA
//   to MFMA.  On MI355X the first sustains ~2.46 PFLOP/s at ~2.37 GHz, the second only ~1.83 PFLOP/s: with the toggle
//   rate of real data the chip lowers the shader clock to ~1.83 GHz (power).  The fused MLP's MFMA operands are real
//   activations and weights, so the second figure is the ceiling that applies to it; bench.py reports both next to the
//   datasheet peak.  s_memtime counts shader cycles, s_memrealtime a constant 100 MHz: their ratio is the clock.
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
PNR_EXPORT int pnrb_probe_mfma_peak(int random_operands, int iters, void* scratch, float* tflops_out_host, float* mhz_out_host,
                                   void* stream)
{
    PNR_REQUIRE(iters >= 1 && scratch && tflops_out_host && mhz_out_host, "pnrb_probe_mfma_peak: bad arguments");
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
        PNR_CHECK_LAUNCH("pnrb_probe_mfma_peak");
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

// pnrb_probe_mfma_order: the same register-only loop on random operands, with the ORDER in which the operands change from MFMA to
// MFMA as the variable (the part is power-limited and the power of an MFMA depends on what toggles at its inputs).  A k-step of a
// two-block x two-tile unit is four MFMAs on four accumulators, A = a weight fragment (block b), B = an activation slice (tile t):
//   0  every MFMA has a new A and a new B (k_mfma_peak<true>)
//   1  (b0,t0) (b0,t1) (b1,t0) (b1,t1): k_mlp_tt's order -- A changes twice, B four times per k-step
//   2  (b0,t0) (b0,t1) (b1,t1) (b1,t0): the "snake" -- A twice, B three times
//   3  A changes every MFMA, B never        4  B changes every MFMA, A never
template <int PATTERN>
__global__ __launch_bounds__(512) void k_mfma_order(unsigned long long* out, int iters)
{
    f32x16 acc[4] = {};
    bf16x8 a[8], b[8];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int k = 0; k < 8; ++k)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            a[k][i] = (__bf16)(((int)(h >> 9) & 0xffff) / 32768.0f - 1.0f);
            h = h * 1664525u + 1013904223u;
            b[k][i] = (__bf16)(((int)(h >> 9) & 0xffff) / 32768.0f - 1.0f);
        }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // (A index, B index) of the k-step's four MFMAs; accumulator = (block, tile) in patterns 1 and 2
            constexpr int P = PATTERN;
            const int a0 = 2 * k, a1 = 2 * k + 1, b0 = 2 * k, b1 = 2 * k + 1;
            if constexpr (P == 0) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a0], b[b0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a1], b[b1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(a0 + 3) & 7], b[(b0 + 5) & 7], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(a1 + 3) & 7], b[(b1 + 5) & 7], acc[3], 0, 0, 0);
            } else if constexpr (P == 1) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a0], b[b0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a0], b[b1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a1], b[b0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a1], b[b1], acc[3], 0, 0, 0);
            } else if constexpr (P == 2) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a0], b[b0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a0], b[b1], acc[1], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a1], b[b1], acc[3], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a1], b[b0], acc[2], 0, 0, 0);
            } else if constexpr (P == 3) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a0], b[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[a1], b[0], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(a0 + 3) & 7], b[0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(a1 + 3) & 7], b[0], acc[3], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[b0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[b1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[(b0 + 5) & 7], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[(b1 + 5) & 7], acc[3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    if (acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] == 12345.678f) out[2] = 1;
}

PNR_EXPORT int pnrb_probe_mfma_order(int pattern, int iters, void* scratch, float* tflops_out_host, float* mhz_out_host, void* stream)
{
    PNR_REQUIRE(iters >= 1 && scratch && tflops_out_host && mhz_out_host && pattern >= 0 && pattern <= 4, "pnrb_probe_mfma_order: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 0;
    PNR_HIP(hipGetDevice(&dev));
    PNR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    hipEvent_t e0, e1;
    PNR_HIP(hipEventCreate(&e0));
    PNR_HIP(hipEventCreate(&e1));
    unsigned long long* out = (unsigned long long*)scratch;
    for (int rep = 0; rep < 2; ++rep) {
        PNR_HIP(hipEventRecord(e0, st));
        switch (pattern) {
        case 0: hipLaunchKernelGGL(k_mfma_order<0>, dim3(cus), dim3(512), 0, st, out, iters); break;
        case 1: hipLaunchKernelGGL(k_mfma_order<1>, dim3(cus), dim3(512), 0, st, out, iters); break;
        case 2: hipLaunchKernelGGL(k_mfma_order<2>, dim3(cus), dim3(512), 0, st, out, iters); break;
        case 3: hipLaunchKernelGGL(k_mfma_order<3>, dim3(cus), dim3(512), 0, st, out, iters); break;
        default: hipLaunchKernelGGL(k_mfma_order<4>, dim3(cus), dim3(512), 0, st, out, iters); break;
        }
        PNR_CHECK_LAUNCH("pnrb_probe_mfma_order");
        PNR_HIP(hipEventRecord(e1, st));
        PNR_HIP(hipEventSynchronize(e1));
    }
    float ms = 0.0f;
    PNR_HIP(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2] = {0, 1};
    PNR_HIP(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    *tflops_out_host = (float)((double)iters * 16.0 * 2.0 * 32 * 32 * 16 * 8 * cus / (ms * 1e-3) / 1e12);
    *mhz_out_host = h[1] ? (float)(100.0 * (double)h[0] / (double)h[1]) : 0.0f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return PNR_OK;
}

// pnrb_probe_raw_read: a pure read of a channel-major raw image in k_composite's own order (per wave: the 8 channel rows of
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
// the same for k_composite2's mapping at 32 < N <= 64: 8 lanes per ray x 8 consecutive samples (two float4 loads per lane and channel
// row), 8 rays per wave, 8 rows in flight
__global__ __launch_bounds__(256) void k_raw_read2(const float* raw, int64_t sc, int64_t R, int N, int CH, float* sink)
{
    const int lane = threadIdx.x & 63, q = lane & 7, g = lane >> 3;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float acc = 0.0f;
    for (int64_t grp = wave; grp < (R + 7) / 8; grp += n_waves) {
        const int64_t ray = grp * 8 + g;
        const bool a0 = ray < R && q * 8 < N, a1 = ray < R && q * 8 + 4 < N;
        const float* p = raw + (ray < R ? ray : R - 1) * N + (a0 ? q * 8 : 0);
        for (int c0 = 0; c0 < CH; c0 += 8) {
            float4 v[8][2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = c0 + j < CH;
                v[j][0] = (a0 && in) ? *reinterpret_cast<const float4*>(p + (int64_t)(c0 + j) * sc) : make_float4(0, 0, 0, 0);
                v[j][1] = (a1 && in) ? *reinterpret_cast<const float4*>(p + (int64_t)(c0 + j) * sc + 4) : make_float4(0, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += ((v[j][0].x + v[j][0].y) + (v[j][0].z + v[j][0].w)) + ((v[j][1].x + v[j][1].y) + (v[j][1].z + v[j][1].w));
        }
    }
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

// The candidate mappings of a ray's samples onto lanes (round-5 verdict item 6: "change the layout, not the kernel" for the N = 64
// level): L lanes per ray x N / L consecutive samples per lane (N / L / 4 float4 loads per lane and channel row), 64 / L rays per wave,
// J channel rows in flight.  Rays are consecutive in a channel row, so a wave-load covers 64 / L rays x N x 4 B contiguous bytes.
// L = 8: k_composite2's mapping at N = 64; L = 16: four samples per lane (k_composite's mapping) with four rays per wave.
template <int L, int J>
__global__ __launch_bounds__(256) void k_raw_read_pattern(const float* raw, int64_t sc, int64_t R, int N, int CH, float* sink)
{
    constexpr int RPW = 64 / L;
    const int lane = threadIdx.x & 63, q = lane % L, g = lane / L;
    const int per = N / L;                  // samples per lane: a multiple of 4
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float acc = 0.0f;
    for (int64_t grp = wave; grp < (R + RPW - 1) / RPW; grp += n_waves) {
        const int64_t ray = grp * RPW + g;
        const bool in_ray = ray < R;
        const float* p = raw + (in_ray ? ray : R - 1) * N + q * per;
        for (int c0 = 0; c0 < CH; c0 += J) {
            for (int k = 0; k < per; k += 4) {
                float4 v[J];
#pragma unroll
                for (int j = 0; j < J; ++j)
                    v[j] = (in_ray && c0 + j < CH) ? *reinterpret_cast<const float4*>(p + (int64_t)(c0 + j) * sc + k) : make_float4(0, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < J; ++j) acc += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            }
        }
    }
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

PNR_EXPORT int pnrb_probe_raw_read_pattern(const float* raw, int64_t raw_stride_c, int64_t n_rays, int n_samples, int n_channels,
                                          int lanes_per_ray, int rows_in_flight, int waves_per_simd, int iters, void* scratch,
                                          float* gbs_out_host, void* stream)
{
    PNR_REQUIRE(raw && scratch && gbs_out_host && iters >= 1 && n_rays >= 1, "pnrb_probe_raw_read_pattern: bad arguments");
    PNR_REQUIRE((lanes_per_ray == 8 || lanes_per_ray == 16 || lanes_per_ray == 32 || lanes_per_ray == 64) &&
                (rows_in_flight == 4 || rows_in_flight == 8 || rows_in_flight == 16) && n_samples % (4 * lanes_per_ray) == 0 &&
                (raw_stride_c % 4) == 0 && waves_per_simd >= 1 && waves_per_simd <= 8, "pnrb_probe_raw_read_pattern: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    PNR_HIP(hipEventCreate(&e0));
    PNR_HIP(hipEventCreate(&e1));
    const int rpw = 64 / lanes_per_ray;
    const int grid = pnr_grid_cap((n_rays + 4 * rpw - 1) / (4 * rpw), waves_per_simd);
    void (*kern)(const float*, int64_t, int64_t, int, int, float*) = nullptr;
#define PNRB_PAT(LL, JJ) if (lanes_per_ray == LL && rows_in_flight == JJ) kern = k_raw_read_pattern<LL, JJ>;
    PNRB_PAT(8, 4) PNRB_PAT(8, 8) PNRB_PAT(8, 16) PNRB_PAT(16, 4) PNRB_PAT(16, 8) PNRB_PAT(16, 16)
    PNRB_PAT(32, 4) PNRB_PAT(32, 8) PNRB_PAT(32, 16) PNRB_PAT(64, 4) PNRB_PAT(64, 8) PNRB_PAT(64, 16)
#undef PNRB_PAT
    for (int i = 0; i <= iters; ++i) {
        if (i == 1) PNR_HIP(hipEventRecord(e0, st));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, raw, raw_stride_c, n_rays, n_samples, n_channels, (float*)scratch);
    }
    PNR_CHECK_LAUNCH("pnrb_probe_raw_read_pattern");
    PNR_HIP(hipEventRecord(e1, st));
    PNR_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    PNR_HIP(hipEventElapsedTime(&ms, e0, e1));
    *gbs_out_host = (float)((double)n_rays * n_samples * 4.0 * n_channels * iters / (ms * 1e-3) / 1e9);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return PNR_OK;
}

PNR_EXPORT int pnrb_probe_raw_read(const float* raw, int64_t raw_stride_c, int64_t n_rays, int n_samples, int n_channels,
                                  int iters, void* scratch, float* gbs_out_host, void* stream)
{
    PNR_REQUIRE(raw && scratch && gbs_out_host && iters >= 1 && n_rays >= 1, "pnrb_probe_raw_read: bad arguments");
    PNR_REQUIRE(n_samples >= 4 && n_samples <= 256 && (n_samples % 4) == 0 && (raw_stride_c % 4) == 0, "pnrb_probe_raw_read: bad shape");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    PNR_HIP(hipEventCreate(&e0));
    PNR_HIP(hipEventCreate(&e1));
    // the mapping pnr_composite uses at this N: 8 lanes x 8 samples for 32 < N <= 64 (k_composite2), else 4 samples per lane
    const bool second = n_samples > 32 && n_samples <= 64 && (n_samples % 8) == 0;
    const int grid = pnr_grid_cap(second ? (n_rays + 31) / 32 : (n_rays + 3) / 4, 8);
    auto kern = second ? k_raw_read2 : k_raw_read;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, raw, raw_stride_c, n_rays, n_samples, n_channels, (float*)scratch);
    PNR_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, raw, raw_stride_c, n_rays, n_samples, n_channels, (float*)scratch);
    PNR_CHECK_LAUNCH("pnrb_probe_raw_read");
    PNR_HIP(hipEventRecord(e1, st));
    PNR_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    PNR_HIP(hipEventElapsedTime(&ms, e0, e1));
    *gbs_out_host = (float)((double)n_rays * n_samples * 4.0 * n_channels * iters / (ms * 1e-3) / 1e9);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return PNR_OK;
}
