#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short i16x4;
__global__ void k(const short* in, short* out)
{
    __shared__ __attribute__((aligned(16))) short s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x;
    // lane a of each 16-lane group loads row (a>>2), cols 4(a&3).. of a [4][16] block; group g -> block g
    i16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(s + (l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main()
{
    short h[1024], *d, *o, r[256];
    for (int i = 0; i < 1024; ++i) h[i] = i;
    hipMalloc(&d, 2048); hipMalloc(&o, 512); hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(r, o, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) { if (l < 20 || l % 16 == 0) printf("lane %2d: %4d %4d %4d %4d\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]); }
    return 0;
}
