// Ground-truth probe of MFMA operand/accumulator layouts on gfx950 (diagnostic tool, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ void k_f32(const float* A /*32x2*/, const float* B /*2x32*/, float* D /*64 lanes x 16*/, int mode)
{
    int l = threadIdx.x;
    float a, b;
    if (mode == 0) { a = A[(l & 31) * 2 + (l >> 5)]; b = B[(l >> 5) * 32 + (l & 31)]; }   // assumed layout
    else { a = A[(l & 31) * 2 + (l >> 5)]; b = B[(l >> 5) * 32 + (l & 31)]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[l * 16 + i] = acc[i];
}
__global__ void k_f32_chain(const float* A /*32x8*/, const float* B /*8x32*/, float* D)
{
    // 4 chained MFMAs as kstep<FP32> does: lane holds a[j] = A[i][2j+hi], b[j] = B[2j+hi][n]
    int l = threadIdx.x, i = l & 31, hi = l >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * 8 + 2 * j + hi], B[(2 * j + hi) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = acc[r];
}
int main()
{
    float hA[64], hB[64], hD[1024], *dA, *dB, *dD;
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 2; ++k) hA[i * 2 + k] = (float)(i + 1) + 0.001f * k;
    for (int k = 0; k < 2; ++k) for (int n = 0; n < 32; ++n) hB[k * 32 + n] = (k == 0 ? 1.0f : 100.0f) * (n + 1);
    hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    k_f32<<<1, 64>>>(dA, dB, dD, 0);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        int n = l & 31, hi = l >> 5, row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float ref = hA[row * 2 + 0] * hB[0 * 32 + n] + hA[row * 2 + 1] * hB[1 * 32 + n];
        if (fabsf(ref - hD[l * 16 + r]) > 1e-3f * fabsf(ref)) { if (bad < 8) printf("f32 mismatch lane %d reg %d: got %g want %g\n", l, r, hD[l * 16 + r], ref); ++bad; }
    }
    printf("f32 32x32x2 single: %d mismatches (lane0 regs: %g %g %g %g; lane32: %g %g)\n", bad, hD[0], hD[1], hD[2], hD[4], hD[32 * 16], hD[32 * 16 + 1]);
    float hA8[256], hB8[256];
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 8; ++k) hA8[i * 8 + k] = sinf(i * 0.37f + k * 1.3f);
    for (int k = 0; k < 8; ++k) for (int n = 0; n < 32; ++n) hB8[k * 32 + n] = cosf(n * 0.11f + k * 0.7f);
    hipMemcpy(dA, hA8, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB8, 1024, hipMemcpyHostToDevice);
    k_f32_chain<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        int n = l & 31, hi = l >> 5, row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float ref = 0; for (int k = 0; k < 8; ++k) ref += hA8[row * 8 + k] * hB8[k * 32 + n];
        if (fabsf(ref - hD[l * 16 + r]) > 1e-4f) { if (bad < 8) printf("chain mismatch lane %d reg %d: got %g want %g\n", l, r, hD[l * 16 + r], ref); ++bad; }
    }
    printf("f32 chain of 4: %d mismatches\n", bad);
    return 0;
}
