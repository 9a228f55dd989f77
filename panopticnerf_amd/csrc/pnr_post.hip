// Label-map post-processing and evaluator counters (the step after the path in the reference's
// `run.py evaluate` -- argmax semantic, instance assignment, panoptic merge, PSNR / mIoU).  The reference's evaluator
// is not in the mount; conventions below are this build's and are listed in DESIGN.md 8:
//   semantic label = argmax_c of the composited semantic map (lowest index on ties, -1 if the row is all -inf / NaN-free
//                    maps never are);
//   instance label = argmax_k of the composited instance map where the semantic class is a "thing" (is_thing[c] != 0),
//                    -1 on "stuff";
//   panoptic id    = class * 1000 + instance on things, class on stuff (the KITTI-360 / Cityscapes id convention).
// Bit-exact with oracle/np_oracle.py (integer work; fp32 comparisons only).
#include <hip/hip_runtime.h>
#include <math.h>

#include "pnr_common.h"

// 16 lanes per ray (4 rays per wave): a ray's row of the map is contiguous, so a lane group reads it as coalesced 64-byte
// pieces (one thread per ray walked the rows with a stride of C floats: 0.58 ms per 529,408-ray frame at 45 / 32, now ~0.1).
// Each lane scans its columns c = l, l + 16, ... in increasing order with a strict >, the butterfly keeps the larger value and,
// on equal values, the lower index: the first maximum of the row, as the sequential scan finds it.  A NaN logit is read as
// -inf (a NaN kept as a lane's running maximum would shadow every later column of that lane -- `v > NaN` is never true -- and
// the lanes of a group could then disagree on the winner): the result is the first maximum of the row's non-NaN values,
// index 0 for a row without any, identical on every lane of the group.
__device__ __forceinline__ void pnr_argmax16(const float* __restrict__ row, int n, int l, float& bv, int& bi)
{
    bv = -INFINITY;
    bi = 0x7fffffff;
    for (int c = l; c < n; c += 16) {
        float v = row[c];
        v = (v == v) ? v : -INFINITY;
        if (v > bv || bi == 0x7fffffff) { bv = v; bi = c; }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        const float ov = __shfl_xor(bv, m, 16);
        const int oi = __shfl_xor(bi, m, 16);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > bv || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
}

__global__ __launch_bounds__(256) void k_panoptic_labels(const float* __restrict__ sem, const float* __restrict__ inst,
                                                         const int32_t* __restrict__ is_thing, int64_t R, int C, int K,
                                                         int32_t* __restrict__ sem_label, int32_t* __restrict__ inst_label,
                                                         int32_t* __restrict__ panoptic)
{
    const int l = threadIdx.x & 15;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, gs = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int64_t Rw = (R + 3) & ~(int64_t)3;        // whole waves run the butterflies: rays past R are computed on row R-1, not stored
    for (int64_t g = g0; g < Rw; g += gs) {
        const int64_t r = g < R ? g : R - 1;
        float bv, iv;
        int best, ib = -1;
        pnr_argmax16(sem + r * C, C, l, bv, best);
        best = __shfl(best, 0, 16);                  // one winner per lane group, whatever the butterfly left in the other lanes
        if (inst && K > 0 && (!is_thing || is_thing[best] != 0)) pnr_argmax16(inst + r * K, K, l, iv, ib);
        if (l == 0 && g < R) {
            if (sem_label) sem_label[r] = best;
            if (inst_label) inst_label[r] = ib;
            if (panoptic) panoptic[r] = ib >= 0 ? best * 1000 + ib : best;
        }
    }
}

PNR_EXPORT int pnr_panoptic_labels(const float* sem, const float* inst, const int32_t* is_thing, int64_t n_rays, int n_sem,
                                   int n_inst, int32_t* sem_label, int32_t* inst_label, int32_t* panoptic, void* stream)
{
    PNR_REQUIRE(n_rays >= 0 && n_sem >= 1 && n_inst >= 0, "pnr_panoptic_labels: bad size");
    if (n_rays == 0) return PNR_OK;
    PNR_REQUIRE(sem, "pnr_panoptic_labels: null semantic map");
    hipLaunchKernelGGL(k_panoptic_labels, dim3(pnr_grid_cap((n_rays + 15) / 16)), dim3(256), 0, (hipStream_t)stream, sem,
                       n_inst > 0 ? inst : nullptr, is_thing, n_rays, n_sem, n_inst, sem_label, inst_label, panoptic);
    PNR_CHECK_LAUNCH("pnr_panoptic_labels");
    return PNR_OK;
}

// conf[gt * n_cls + pred] += 1 for pixels with 0 <= gt < n_cls and 0 <= pred < n_cls.  Per-block LDS histogram, then
// one global integer atomic per non-zero cell: order-independent, so the result is exact and deterministic.
__global__ __launch_bounds__(256) void k_confusion(const int32_t* __restrict__ pred, const int32_t* __restrict__ gt, int64_t R,
                                                   int n_cls, unsigned long long* __restrict__ conf)
{
    extern __shared__ unsigned int h[];
    const int cells = n_cls * n_cls;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        const int g = gt[r], p = pred[r];
        if (g >= 0 && g < n_cls && p >= 0 && p < n_cls) atomicAdd(&h[g * n_cls + p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += blockDim.x)
        if (h[i]) atomicAdd(&conf[i], (unsigned long long)h[i]);
}

// the same for label spaces too large for an LDS histogram (segment-pair counts of the panoptic-quality metric):
// one global integer atomic per pixel
__global__ __launch_bounds__(256) void k_confusion_big(const int32_t* __restrict__ pred, const int32_t* __restrict__ gt, int64_t R,
                                                       int n_cls, unsigned long long* __restrict__ conf)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        const int g = gt[r], p = pred[r];
        if (g >= 0 && g < n_cls && p >= 0 && p < n_cls) atomicAdd(&conf[(int64_t)g * n_cls + p], 1ull);
    }
}

PNR_EXPORT int pnr_confusion(const int32_t* pred, const int32_t* gt, int64_t n, int n_classes, int64_t* conf, void* stream)
{
    PNR_REQUIRE(n >= 0 && n_classes >= 1 && n_classes <= 8192, "pnr_confusion: bad size (n_classes <= 8192)");
    if (n == 0) return PNR_OK;
    PNR_REQUIRE(pred && gt && conf, "pnr_confusion: null pointer");
    if (n_classes > 128) {
        hipLaunchKernelGGL(k_confusion_big, dim3(pnr_grid_cap((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, gt, n,
                           n_classes, (unsigned long long*)conf);
        PNR_CHECK_LAUNCH("pnr_confusion");
        return PNR_OK;
    }
    int grid = (int)((n + 256 * 16 - 1) / (256 * 16));
    grid = grid < 1 ? 1 : grid > 1024 ? 1024 : grid;
    hipLaunchKernelGGL(k_confusion, dim3(grid), dim3(256), (size_t)n_classes * n_classes * sizeof(unsigned int), (hipStream_t)stream,
                       pred, gt, n, n_classes, (unsigned long long*)conf);
    PNR_CHECK_LAUNCH("pnr_confusion");
    return PNR_OK;
}
