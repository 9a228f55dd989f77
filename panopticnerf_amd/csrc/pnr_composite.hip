// K4: raw2outputs -- transmittance / alpha-compositing scan of RGB, depth, acc and the
// panoptic (semantic / instance) logit and fixed-field maps.  SURVEY.md 8a row a6; the
// reference function is raw2outputs in lib/networks/renderer (branch not in the mount, see
// include/pnr.h).  HBM-bound: every (sigma, rgb, logit) value is read exactly once.
//
// Mapping (wave64): a ray of N samples (N % 4 == 0, N <= 256) is owned by a group of
// SUB = pow2ceil(N/4) consecutive lanes, each lane holding 4 consecutive samples, so with
// the channel-major raw image written by the MLP kernel every channel row of a ray group is
// one 16 B-per-lane coalesced load (N=64: 4 rays per wave, 1 KiB per wave-load; N=192: one
// ray per wave, 768 B).  T(t) is a segmented (width SUB) wave-level inclusive product scan of
// the per-lane products of (1 - alpha + 1e-10); per-channel sums are width-SUB butterflies.
// Nothing is re-read, so the samples are staged in registers, not LDS: there is no reuse for
// LDS to serve (cdna_hip_programming.md common mistake 7).
#include "pnr_common.h"

struct CompositeArgs {
    const float* raw; int64_t stride_s, stride_c;
    const float* z; const float* rays; const float* noise;
    const int32_t* label_sem; const int32_t* label_inst;
    int64_t R; int N, C, K, sem_mode, white_bkgd;
    float *rgb, *depth, *acc, *weights, *sem, *inst, *fix_sem, *fix_inst;
};

struct f4 { float v[4]; };

template <bool CH_MAJOR>
__device__ __forceinline__ f4 load4(const float* __restrict__ raw, int64_t ss, int64_t sc, int64_t s0, int c, bool active)
{
    f4 o;
    if (!active) { o.v[0] = o.v[1] = o.v[2] = o.v[3] = 0.0f; return o; }
    if (CH_MAJOR) {
        const float4 t = *reinterpret_cast<const float4*>(raw + (int64_t)c * sc + s0);
        o.v[0] = t.x; o.v[1] = t.y; o.v[2] = t.z; o.v[3] = t.w;
    } else {
        const float* p = raw + s0 * ss + (int64_t)c * sc;
#pragma unroll
        for (int k = 0; k < 4; ++k) o.v[k] = p[k * ss];
    }
    return o;
}

__device__ __forceinline__ float group_sum(float x, int SUB)
{
    for (int d = SUB >> 1; d > 0; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

__device__ __forceinline__ float dot4(const f4& w, const f4& v)
{
    return fmaf(w.v[3], v.v[3], fmaf(w.v[2], v.v[2], fmaf(w.v[1], v.v[1], w.v[0] * v.v[0])));
}

template <bool CH_MAJOR>
__global__ __launch_bounds__(256) void k_composite(CompositeArgs a)
{
    const int lane = threadIdx.x & 63;
    const int N = a.N, nq = N >> 2;
    int SUB = 1;
    while (SUB < nq) SUB <<= 1;
    const int rpw = 64 / SUB;            // rays per wave
    const int q = lane & (SUB - 1);      // lane's position in its ray group
    const int g = lane / SUB;            // which ray of the wave
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t n_groups = (a.R + rpw - 1) / rpw;
    const int CH_SEM = 4, CH_INST = 4 + a.C;

    for (int64_t grp = wave_global; grp < n_groups; grp += n_waves) {
        const int64_t ray = grp * rpw + g;
        const bool active = (ray < a.R) && (q < nq);
        const int64_t rayc = ray < a.R ? ray : a.R - 1;
        const int64_t s0 = rayc * N + (active ? 4 * q : 0);

        // ---- phase 1: weights
        f4 zz, sg;
        if (active) {
            const float4 t = *reinterpret_cast<const float4*>(a.z + s0);
            zz.v[0] = t.x; zz.v[1] = t.y; zz.v[2] = t.z; zz.v[3] = t.w;
        } else {
            zz.v[0] = zz.v[1] = zz.v[2] = zz.v[3] = 0.0f;
        }
        sg = load4<CH_MAJOR>(a.raw, a.stride_s, a.stride_c, s0, 3, active);
        if (a.noise && active) {
            const float4 t = *reinterpret_cast<const float4*>(a.noise + s0);
            sg.v[0] += t.x; sg.v[1] += t.y; sg.v[2] += t.z; sg.v[3] += t.w;
        }
        const float dx = a.rays[rayc * 8 + 3], dy = a.rays[rayc * 8 + 4], dz = a.rays[rayc * 8 + 5];
        const float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float znext = __shfl_down(zz.v[0], 1, 64);   // first sample of the next lane
        f4 w, tt;
        float P = 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * q + k;
            const float zn = (k < 3) ? zz.v[k + 1] : znext;
            float dist = (i + 1 < N) ? (zn - zz.v[k]) : 1e10f;
            dist *= dn;
            const float s = fmaxf(sg.v[k], 0.0f);
            const float alpha = 1.0f - expf(-(s * dist));
            w.v[k] = alpha * P;            // alpha * (product of this lane's earlier factors)
            tt.v[k] = (1.0f - alpha) + 1e-10f;
            P *= tt.v[k];
        }
        if (!active) P = 1.0f;
        // segmented inclusive product scan over the SUB lanes of the ray
        float x = P;
        for (int d = 1; d < SUB; d <<= 1) {
            const float y = __shfl_up(x, d, 64);
            if (q >= d) x *= y;
        }
        float excl = __shfl_up(x, 1, 64);
        if (q == 0) excl = 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) w.v[k] = active ? w.v[k] * excl : 0.0f;
        if (a.weights && active) {
            *reinterpret_cast<float4*>(a.weights + s0) = make_float4(w.v[0], w.v[1], w.v[2], w.v[3]);
        }

        // ---- phase 2: per-channel weighted sums
        const float accv = group_sum((w.v[0] + w.v[1]) + (w.v[2] + w.v[3]), SUB);
        const float depv = group_sum(dot4(w, zz), SUB);
        float rgbv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f4 v = load4<CH_MAJOR>(a.raw, a.stride_s, a.stride_c, s0, c, active);
#pragma unroll
            for (int k = 0; k < 4; ++k) v.v[k] = 1.0f / (1.0f + expf(-v.v[k]));
            rgbv[c] = group_sum(dot4(w, v), SUB);
            if (a.white_bkgd) rgbv[c] += 1.0f - accv;
        }
        const bool writer = active && q == 0;
        if (writer) {
            if (a.rgb) { a.rgb[ray * 3 + 0] = rgbv[0]; a.rgb[ray * 3 + 1] = rgbv[1]; a.rgb[ray * 3 + 2] = rgbv[2]; }
            if (a.depth) a.depth[ray] = depv;
            if (a.acc) a.acc[ray] = accv;
        }

        // learned fields: two passes (semantic, instance) over their channel ranges
#pragma unroll 1
        for (int field = 0; field < 2; ++field) {
            const int nch = field == 0 ? a.C : a.K;
            const int ch0 = field == 0 ? CH_SEM : CH_INST;
            float* outp = field == 0 ? a.sem : a.inst;
            if (nch == 0 || outp == nullptr) continue;
            f4 mx, den;
            if (a.sem_mode == 1) {   // softmax over the field's channels, per sample (online max/sum)
#pragma unroll
                for (int k = 0; k < 4; ++k) { mx.v[k] = -INFINITY; den.v[k] = 0.0f; }
                for (int c = 0; c < nch; ++c) {
                    const f4 v = load4<CH_MAJOR>(a.raw, a.stride_s, a.stride_c, s0, ch0 + c, active);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float m2 = fmaxf(mx.v[k], v.v[k]);
                        den.v[k] = den.v[k] * expf(mx.v[k] - m2) + expf(v.v[k] - m2);
                        mx.v[k] = m2;
                    }
                }
            }
            int c = 0;
            // 4 channel rows in flight per lane (>= 3 KiB per wave outstanding)
            for (; c + 4 <= nch; c += 4) {
                f4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = load4<CH_MAJOR>(a.raw, a.stride_s, a.stride_c, s0, ch0 + c + j, active);
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (a.sem_mode == 1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[j].v[k] = expf(v[j].v[k] - mx.v[k]) / den.v[k];
                    }
                    r[j] = group_sum(dot4(w, v[j]), SUB);
                }
                if (writer) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) outp[ray * nch + c + j] = r[j];
                }
            }
            for (; c < nch; ++c) {
                f4 v = load4<CH_MAJOR>(a.raw, a.stride_s, a.stride_c, s0, ch0 + c, active);
                if (a.sem_mode == 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v.v[k] = expf(v.v[k] - mx.v[k]) / den.v[k];
                }
                const float r = group_sum(dot4(w, v), SUB);
                if (writer) outp[ray * nch + c] = r;
            }
        }

        // fixed (bbox-prior) fields: weighted histogram of the per-sample labels
#pragma unroll 1
        for (int field = 0; field < 2; ++field) {
            const int nch = field == 0 ? a.C : a.K;
            const int32_t* lab = field == 0 ? a.label_sem : a.label_inst;
            float* outp = field == 0 ? a.fix_sem : a.fix_inst;
            if (nch == 0 || outp == nullptr || lab == nullptr) continue;
            int l[4] = {-1, -1, -1, -1};
            if (active) {
                const int4 t = *reinterpret_cast<const int4*>(lab + s0);
                l[0] = t.x; l[1] = t.y; l[2] = t.z; l[3] = t.w;
            }
            for (int c = 0; c < nch; ++c) {
                float p = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) p += (l[k] == c) ? w.v[k] : 0.0f;
                const float r = group_sum(p, SUB);
                if (writer) outp[ray * nch + c] = r;
            }
        }
    }
}

PNR_EXPORT int pnr_composite(const float* raw, int64_t raw_stride_s, int64_t raw_stride_c, const float* z,
                             const float* rays, const float* noise, const int32_t* label_sem,
                             const int32_t* label_inst, int64_t n_rays, int n_samples, int n_sem, int n_inst,
                             int sem_mode, int white_bkgd, float* rgb, float* depth, float* acc, float* weights,
                             float* sem, float* inst, float* fix_sem, float* fix_inst, void* stream)
{
    PNR_REQUIRE(n_rays <= 0 || (raw && z && rays), "pnr_composite: null pointer");
    PNR_REQUIRE(n_samples >= 4 && n_samples <= 256 && (n_samples % 4) == 0,
                "pnr_composite: n_samples=%d must be a multiple of 4 in [4,256]", n_samples);
    PNR_REQUIRE(n_sem >= 0 && n_inst >= 0 && (sem_mode == 0 || sem_mode == 1), "pnr_composite: bad field sizes");
    PNR_REQUIRE((((uintptr_t)z) & 15) == 0, "pnr_composite: z must be 16-byte aligned");
    PNR_REQUIRE(!weights || (((uintptr_t)weights) & 15) == 0, "pnr_composite: weights must be 16-byte aligned");
    PNR_REQUIRE((((uintptr_t)noise) & 15) == 0 && (((uintptr_t)label_sem) & 15) == 0 && (((uintptr_t)label_inst) & 15) == 0,
                "pnr_composite: noise / label arrays must be 16-byte aligned");
    if (n_rays <= 0) return PNR_OK;
    CompositeArgs a;
    a.raw = raw; a.stride_s = raw_stride_s; a.stride_c = raw_stride_c; a.z = z; a.rays = rays; a.noise = noise;
    a.label_sem = label_sem; a.label_inst = label_inst; a.R = n_rays; a.N = n_samples; a.C = n_sem; a.K = n_inst;
    a.sem_mode = sem_mode; a.white_bkgd = white_bkgd; a.rgb = rgb; a.depth = depth; a.acc = acc; a.weights = weights;
    a.sem = sem; a.inst = inst; a.fix_sem = fix_sem; a.fix_inst = fix_inst;
    int sub = 1;
    while (sub < n_samples / 4) sub <<= 1;
    const int rpw = 64 / sub;
    const int64_t n_groups = (n_rays + rpw - 1) / rpw;
    const int grid = pnr_grid_cap((n_groups + 3) / 4, 8);
    const bool ch_major = raw_stride_s == 1 && (raw_stride_c % 4) == 0 && (((uintptr_t)raw) & 15) == 0;
    if (ch_major)
        hipLaunchKernelGGL(k_composite<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_composite<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    PNR_CHECK_LAUNCH("pnr_composite");
    return PNR_OK;
}
