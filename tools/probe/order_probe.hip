// Diagnostic: does the ORDER in which the compositing kernel walks the channel-major raw image matter to HBM?
// Same bytes, same instruction count, 48 of 64 lanes x 16 B per load (a 192-sample ray row = 768 B), 8 loads in flight:
//   A  "8 channel rows of one ray"      (today's k_composite batch: eight 768 B pieces, one per channel row)
//   B  "8 consecutive rays of one row"  (one contiguous 6 KB piece per wave, 24 KB per workgroup)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void k_read(const float* raw, int64_t sc, int R, int N, int CH, float* sink)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const bool active = lane * 4 < N;
    float acc = 0.0f;
    if (MODE == 0) {
        for (int64_t ray = wave; ray < R; ray += n_waves) {
            const float* p = raw + ray * N + lane * 4;
            for (int c0 = 0; c0 < CH; c0 += 8) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (active && c0 + j < CH) ? *reinterpret_cast<const float4*>(p + (int64_t)(c0 + j) * sc) : make_float4(0, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            }
        }
    } else {
        for (int64_t r0 = wave * 8; r0 < R; r0 += n_waves * 8) {
            const float* p = raw + r0 * N + lane * 4;
            for (int c = 0; c < CH; ++c) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (active && r0 + j < R) ? *reinterpret_cast<const float4*>(p + (int64_t)c * sc + (int64_t)j * N) : make_float4(0, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += (v[j].x + v[j].y) + (v[j].z + v[j].w);
            }
        }
    }
    if (acc == 12345.678f) sink[threadIdx.x] = acc;
}

int main()
{
    const int R = 65536, N = 192, CH = 81;
    const int64_t S = (int64_t)R * N, sc = S + 64;
    float *raw, *sink;
    hipMalloc(&raw, sizeof(float) * sc * CH); hipMalloc(&sink, 4096);
    hipMemset(raw, 0x3c, sizeof(float) * sc * CH);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)R * N * 4 * CH;
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 2; ++mode)
            for (int wpc = 8; wpc <= 8; wpc += 4) {
                const int grid = 256 * wpc;     // wpc workgroups of 4 waves per CU
                hipEventRecord(e0);
                for (int it = 0; it < 5; ++it) {
                    if (mode == 0) hipLaunchKernelGGL(k_read<0>, dim3(grid), dim3(256), 0, 0, raw, sc, R, N, CH, sink);
                    else hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, 0, raw, sc, R, N, CH, sink);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
                if (rep) printf("%s: %.3f ms  %.2f TB/s\n", mode ? "B: 8 rays of one channel row   " : "A: 8 channel rows of one ray   ", ms, bytes / ms / 1e9);
            }
    return 0;
}
