// K4: raw2outputs -- transmittance / alpha-compositing scan of RGB, depth, acc and the
// panoptic (semantic / instance) logit and fixed-field maps.  SURVEY.md 8a row a6; the
// reference function is raw2outputs in lib/networks/renderer (branch not in the mount, see
// include/pnr.h).  HBM-bound: every (sigma, rgb, logit) value is read exactly once.
//
// Mapping (wave64): a ray of N samples (N % 4 == 0, N <= 256) is owned by a group of
// SUB = pow2ceil(N/4) consecutive lanes (compile-time), each lane holding 4 consecutive samples,
// so with the channel-major raw image written by the MLP kernel every channel row of a ray group
// is one 16 B-per-lane coalesced load (N=64: 4 rays per wave, 1 KiB per wave-load; N=192: one ray
// per wave, 768 B).
//   * T(t): segmented (width SUB) wave-level inclusive product scan of the per-lane products of
//     (1 - alpha + 1e-10) -- the "wavefront-level prefix sum" of BASELINE.json's north_star.
//   * channel sums: 8 channel rows per batch; the NEXT batch's 8 loads are issued before the
//     current batch is reduced, so >= 6 KiB per wave stays in flight through the butterflies.
//   * fixed (bbox-prior) fields: a per-ray histogram of the sample labels in LDS, accumulated
//     in 2^-30 fixed point with integer ds_add (associative => bit-reproducible whatever the
//     order), instead of one compare-select reduction per class.
// Nothing is re-read, so samples are staged in registers, not LDS: there is no reuse for LDS
// to serve (cdna_hip_programming.md common mistake 7).
#include <stdlib.h>

#include "pnr_common.h"
#include "pnr_fuse_record.h"
#include "pnr_lane_ops.h"

struct CompositeArgs {
    const float* raw; int64_t stride_s, stride_c;
    const float* z; const float* rays; const float* noise;
    const int32_t* label_sem; const int32_t* label_inst;
    int64_t R; int N, C, K, sem_mode, white_bkgd;
    float *rgb, *depth, *acc, *weights, *sem, *inst, *fix_sem, *fix_inst;
    int use_hist;       // 1: LDS histogram for the fixed fields (fits in LDS), 0: reduction fallback
};

struct f4 { float v[4]; };
#ifndef CMP_GRID_PER_CU
#define CMP_GRID_PER_CU 8      /* resident 256-thread workgroups per CU the grid is capped at (grid-stride over the rest) */
#endif
#ifndef CMP_UB
#define CMP_UB 8        // channel rows per batch
#endif
#define CMP_FIX_SCALE 1073741824.0f   // 2^30
#ifndef CMP_PREFETCH
#define CMP_PREFETCH 0      /* 1 = z / sigma of the next ray group requested one group ahead.  SLOWER, so off: N = 192 0.782 -> 0.813 ms with
                               labels (5.54 -> 5.33 TB/s), 0.720 -> 0.729 without; N = 64 +-0 (round 4, same process, same buffers,
                               tools/composite_ab.py; registers 126 -> 160 cost a wave per SIMD) */
#endif

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

__device__ __forceinline__ float dot4(const f4& w, const f4& v)
{
    return fmaf(w.v[3], v.v[3], fmaf(w.v[2], v.v[2], fmaf(w.v[1], v.v[1], w.v[0] * v.v[0])));
}

template <bool CH_MAJOR, int SUB, bool SOFTMAX>
__global__ __launch_bounds__(256) void k_composite(CompositeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];   // [4 waves][RPW][C+K] when use_hist
    constexpr int RPW = 64 / SUB;          // rays per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = a.N, nq4 = N >> 2;
    const int q = lane & (SUB - 1);        // lane's position in its ray group
    const int g = lane / SUB;              // which ray of the wave
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t n_groups = (a.R + RPW - 1) / RPW;
    const int CK = a.C + a.K;
    const int nq = 3 + CK;                 // composited raw channels: rgb, semantic, instance (sigma excluded)
    uint32_t* hist = s_hist + ((size_t)wave * RPW + g) * CK;

    // z and sigma of a ray group: the head of the dependent chain (load -> exp -> scan -> weights).  With CMP_PREFETCH
    // they are requested one group ahead, while the previous group's channel rows are being reduced, so a wave's next
    // weights never wait for an HBM round trip.
    auto load_zs = [&](int64_t grp, f4& zz, f4& sg) {
        const int64_t ray = grp * RPW + g;
        const bool active = (ray < a.R) && (q < nq4) && (grp < n_groups);
        const int64_t rayc = ray < a.R ? ray : a.R - 1;
        const int64_t s0 = rayc * N + (active ? 4 * q : 0);
        if (active) {
            const float4 t = *reinterpret_cast<const float4*>(a.z + s0);
            zz.v[0] = t.x; zz.v[1] = t.y; zz.v[2] = t.z; zz.v[3] = t.w;
        } else {
            zz.v[0] = zz.v[1] = zz.v[2] = zz.v[3] = 0.0f;
        }
        sg = load4<CH_MAJOR>(a.raw, a.stride_s, a.stride_c, s0, 3, active);
    };
    f4 pz, ps;
    if (CMP_PREFETCH) load_zs(wave_global, pz, ps);

    for (int64_t grp = wave_global; grp < n_groups; grp += n_waves) {
        const int64_t ray = grp * RPW + g;
        const bool active = (ray < a.R) && (q < nq4);
        const int64_t rayc = ray < a.R ? ray : a.R - 1;
        const int64_t s0 = rayc * N + (active ? 4 * q : 0);
        const bool writer = active && q == 0;

        // ---- phase 1: weights
        f4 zz, sg;
        if (CMP_PREFETCH) { zz = pz; sg = ps; }
        else load_zs(grp, zz, sg);
        // first batch of channel rows: in flight while the weights are computed
        f4 nxt[CMP_UB];
#pragma unroll
        for (int j = 0; j < CMP_UB; ++j) nxt[j] = load4<CH_MAJOR>(a.raw, a.stride_s, a.stride_c, s0, j < 3 ? j : j + 1, active && j < nq);
        if (CMP_PREFETCH) load_zs(grp + n_waves, pz, ps);         // next group's chain head
        if (a.noise && active) {
            const float4 t = *reinterpret_cast<const float4*>(a.noise + s0);
            sg.v[0] += t.x; sg.v[1] += t.y; sg.v[2] += t.z; sg.v[3] += t.w;
        }
        const float dx = a.rays[rayc * 8 + 3], dy = a.rays[rayc * 8 + 4], dz = a.rays[rayc * 8 + 5];
        const float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float znext = __shfl_down(zz.v[0], 1, 64);   // first sample of the next lane
        f4 w;
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
            P *= (1.0f - alpha) + 1e-10f;
        }
        if (!active) P = 1.0f;
        // segmented inclusive product scan over the SUB lanes of the ray
        float x = P;
#pragma unroll
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
        const float accv = group_sum<SUB>((w.v[0] + w.v[1]) + (w.v[2] + w.v[3]));
        const float depv = group_sum<SUB>(dot4(w, zz));
        if (writer) {
            if (a.depth) a.depth[ray] = depv;
            if (a.acc) a.acc[ray] = accv;
        }

        // ---- fixed (bbox-prior) fields
        const bool want_fs = a.fix_sem && a.label_sem && a.C, want_fi = a.fix_inst && a.label_inst && a.K;
        if (want_fs || want_fi) {
            int ls[4] = {-1, -1, -1, -1}, li[4] = {-1, -1, -1, -1};
            if (active && want_fs) { const int4 t = *reinterpret_cast<const int4*>(a.label_sem + s0); ls[0] = t.x; ls[1] = t.y; ls[2] = t.z; ls[3] = t.w; }
            if (active && want_fi) { const int4 t = *reinterpret_cast<const int4*>(a.label_inst + s0); li[0] = t.x; li[1] = t.y; li[2] = t.z; li[3] = t.w; }
            if (a.use_hist) {
                // LDS ops of one wave execute in order and only this wave touches its histograms: no barrier
                for (int c = q; c < CK; c += SUB) hist[c] = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t fx = (uint32_t)(w.v[k] * CMP_FIX_SCALE + 0.5f);
                    if (want_fs && ls[k] >= 0 && ls[k] < a.C) atomicAdd(&hist[ls[k]], fx);
                    if (want_fi && li[k] >= 0 && li[k] < a.K) atomicAdd(&hist[a.C + li[k]], fx);
                }
                if (ray < a.R) {
                    if (want_fs) for (int c = q; c < a.C; c += SUB) a.fix_sem[ray * a.C + c] = (float)hist[c] * (1.0f / CMP_FIX_SCALE);
                    if (want_fi) for (int c = q; c < a.K; c += SUB) a.fix_inst[ray * a.K + c] = (float)hist[a.C + c] * (1.0f / CMP_FIX_SCALE);
                }
            } else {
                for (int c = 0; c < CK; ++c) {
                    const bool is_s = c < a.C;
                    if ((is_s && !want_fs) || (!is_s && !want_fi)) continue;
                    const int cc = is_s ? c : c - a.C;
                    float p = 0.0f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) p += ((is_s ? ls[k] : li[k]) == cc) ? w.v[k] : 0.0f;
                    const float r = group_sum<SUB>(p);
                    if (writer) (is_s ? a.fix_sem : a.fix_inst)[ray * (is_s ? a.C : a.K) + cc] = r;
                }
            }
        }

        // ---- softmax mode: per-sample max / denominator of each learned field (extra pass over its rows)
        f4 mx_s, den_s, mx_i, den_i;
        if constexpr (SOFTMAX) {
            auto field_stats = [&](int nch, int ch0, f4& mx, f4& den) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { mx.v[k] = -INFINITY; den.v[k] = 0.0f; }
                for (int c = 0; c < nch; ++c) {
                    const f4 v = load4<CH_MAJOR>(a.raw, a.stride_s, a.stride_c, s0, ch0 + c, active);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float d = v.v[k] - mx.v[k], e = expf(-fabsf(d));      // one expf per value (the other factor is expf(0))
                        den.v[k] = d <= 0.0f ? den.v[k] + e : den.v[k] * e + 1.0f;
                        mx.v[k] = fmaxf(mx.v[k], v.v[k]);
                    }
                }
            };
            field_stats(a.C, 4, mx_s, den_s);
            field_stats(a.K, 4 + a.C, mx_i, den_i);
        }

        // ---- phase 2: all composited channels, CMP_UB rows per batch.  Each row's weighted partial sum is
        // taken as soon as the row is used and its register is immediately re-armed with the row of the NEXT
        // batch, so 8 loads (6-8 KiB per wave) are in flight through the butterflies below.
#pragma unroll 1
        for (int q0 = 0; q0 < nq; q0 += CMP_UB) {
            float r[CMP_UB];
            const bool more = q0 + CMP_UB < nq;
#pragma unroll
            for (int j = 0; j < CMP_UB; ++j) {
                const int qq = q0 + j;
                f4 v = nxt[j];
                if (qq < 3) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v.v[k] = 1.0f / (1.0f + expf(-v.v[k]));
                } else if constexpr (SOFTMAX) {
                    const bool fs = (qq - 3) < a.C;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        v.v[k] = expf(v.v[k] - (fs ? mx_s.v[k] : mx_i.v[k])) / (fs ? den_s.v[k] : den_i.v[k]);
                }
                r[j] = dot4(w, v);
                const int q2 = qq + CMP_UB;               // >= 8 > 3: never an rgb row -> raw channel q2 + 1
                if (more) nxt[j] = load4<CH_MAJOR>(a.raw, a.stride_s, a.stride_c, s0, q2 + 1, active && q2 < nq);
            }
            group_sum_batch<SUB, CMP_UB>(r);
            if (writer) {
#pragma unroll
                for (int j = 0; j < CMP_UB; ++j) {
                    const int qq = q0 + j;
                    if (qq >= nq) continue;
                    if (qq < 3) { if (a.rgb) a.rgb[ray * 3 + qq] = a.white_bkgd ? r[j] + (1.0f - accv) : r[j]; }
                    else if (qq - 3 < a.C) { if (a.sem) a.sem[ray * a.C + (qq - 3)] = r[j]; }
                    else { if (a.inst) a.inst[ray * a.K + (qq - 3 - a.C)] = r[j]; }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------- second mapping (channel-major raw)
// L lanes per ray x M = 4*M4 CONSECUTIVE samples per lane (N <= L*M): the weighted partial sum is M FMAs per lane, and
// both the transmittance scan and the per-ray sums stay inside a 16-lane DPP row (row_shr / quad_perm / row_mirror: no
// LDS crossbar, no bpermute); CMP2_UB rows (x M4 loads) are in flight per wave through the reductions.
// Measured on MI355X against the 4-samples-per-lane mapping above, same process and buffers (tools/composite_ab.py,
// profiles/README.md): N = 64 as 8 lanes x 8 samples (8 rays per wave): +1..3 % -> used for 32 < N <= 64;
// N = 192 as 16 lanes x 12 samples (every lane busy, ~3x fewer issued instructions per byte): -1..-4 % at 4 or 8 rows
// in flight, -9 % at 2 -> NOT used there: at N = 192 the kernel is bound by what HBM delivers for 768-byte pieces, not
// by instruction issue.
#ifndef CMP2_UB
#define CMP2_UB 8
#endif

template <int L, int M4, bool SOFTMAX>
__global__ __launch_bounds__(256) void k_composite2(CompositeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_hist[];   // [4 waves][RPW][C+K] when use_hist
    constexpr int M = 4 * M4, RPW = 64 / L;
    static_assert(L == 8 || L == 16, "a ray must sit inside one 16-lane DPP row");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int N = a.N;
    const int q = lane & (L - 1), g = lane / L;
    const int i0 = q * M;                  // this lane's first sample
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t n_groups = (a.R + RPW - 1) / RPW;
    const int CK = a.C + a.K;
    const int nq = 3 + CK;
    uint32_t* hist = s_hist + ((size_t)wave * RPW + g) * CK;
    const int64_t sc = a.stride_c;

    for (int64_t grp = wave_global; grp < n_groups; grp += n_waves) {
        const int64_t ray = grp * RPW + g;
        const bool rvalid = ray < a.R;
        const int64_t rayc = rvalid ? ray : a.R - 1;
        bool act[M4];
#pragma unroll
        for (int j = 0; j < M4; ++j) act[j] = rvalid && (i0 + 4 * j < N);
        const int64_t s0 = rayc * N + (act[0] ? i0 : 0);
        const bool writer = rvalid && q == 0;
        auto row = [&](const float* base, float (&v)[M]) {       // one channel row's share of this lane
#pragma unroll
            for (int j = 0; j < M4; ++j) {
                float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (act[j]) t = *reinterpret_cast<const float4*>(base + s0 + 4 * j);
                v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
            }
        };

        // ---- phase 1: weights
        float zz[M], w[M];
        row(a.z, zz);
        row(a.raw + 3 * sc, w);                                   // sigma, becomes the weights below
        float nxt[CMP2_UB][M];
#pragma unroll
        for (int j = 0; j < CMP2_UB; ++j) {
            if (j < nq) row(a.raw + (int64_t)(j < 3 ? j : j + 1) * sc, nxt[j]);
        }
        if (a.noise) {
            float nz[M];
            row(a.noise, nz);
#pragma unroll
            for (int k = 0; k < M; ++k) w[k] += nz[k];
        }
        const float dx = a.rays[rayc * 8 + 3], dy = a.rays[rayc * 8 + 4], dz = a.rays[rayc * 8 + 5];
        const float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float znext = dpp_or<0x101>(zz[0], 0.0f);           // row_shl:1 -- first sample of the next lane
        float P = 1.0f;
#pragma unroll
        for (int k = 0; k < M; ++k) {
            const int i = i0 + k;
            const float zn = (k < M - 1) ? zz[k + 1] : znext;
            float dist = (i + 1 < N) ? (zn - zz[k]) : 1e10f;
            dist *= dn;
            const float sgm = fmaxf(w[k], 0.0f);
            const float alpha = 1.0f - expf(-(sgm * dist));
            w[k] = alpha * P;
            P *= (1.0f - alpha) + 1e-10f;
        }
        if (!act[0]) P = 1.0f;
        // segmented inclusive product scan over the L lanes of the ray, inside the DPP row (row_shr:d; a lane without a
        // source keeps the identity)
        float x = P;
        {
            float y = dpp_or<0x111>(x, 1.0f); x *= (L == 16 || q >= 1) ? y : 1.0f;
            y = dpp_or<0x112>(x, 1.0f); x *= (L == 16 || q >= 2) ? y : 1.0f;
            y = dpp_or<0x114>(x, 1.0f); x *= (L == 16 || q >= 4) ? y : 1.0f;
            if constexpr (L == 16) { y = dpp_or<0x118>(x, 1.0f); x *= y; }
        }
        float excl = dpp_or<0x111>(x, 1.0f);
        if (q == 0) excl = 1.0f;
        float accp = 0.0f, depp = 0.0f;
#pragma unroll
        for (int k = 0; k < M; ++k) {
            w[k] = act[k >> 2] ? w[k] * excl : 0.0f;
            accp += w[k];
            depp = fmaf(w[k], zz[k], depp);
        }
        if (a.weights) {
#pragma unroll
            for (int j = 0; j < M4; ++j)
                if (act[j]) *reinterpret_cast<float4*>(a.weights + s0 + 4 * j) = make_float4(w[4 * j], w[4 * j + 1], w[4 * j + 2], w[4 * j + 3]);
        }
        const float accv = group_sum<L>(accp);
        const float depv = group_sum<L>(depp);
        if (writer) {
            if (a.depth) a.depth[ray] = depv;
            if (a.acc) a.acc[ray] = accv;
        }

        // ---- fixed (bbox-prior) fields: per-ray LDS histogram in 2^-30 fixed point (integer adds: order-independent)
        const bool want_fs = a.fix_sem && a.label_sem && a.C, want_fi = a.fix_inst && a.label_inst && a.K;
        if (want_fs || want_fi) {
            for (int c = q; c < CK; c += L) hist[c] = 0;
#pragma unroll
            for (int j = 0; j < M4; ++j) {
                int4 ls = make_int4(-1, -1, -1, -1), li = make_int4(-1, -1, -1, -1);
                if (act[j] && want_fs) ls = *reinterpret_cast<const int4*>(a.label_sem + s0 + 4 * j);
                if (act[j] && want_fi) li = *reinterpret_cast<const int4*>(a.label_inst + s0 + 4 * j);
                const int lsv[4] = {ls.x, ls.y, ls.z, ls.w}, liv[4] = {li.x, li.y, li.z, li.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t fx = (uint32_t)(w[4 * j + k] * CMP_FIX_SCALE + 0.5f);
                    if (want_fs && lsv[k] >= 0 && lsv[k] < a.C) atomicAdd(&hist[lsv[k]], fx);
                    if (want_fi && liv[k] >= 0 && liv[k] < a.K) atomicAdd(&hist[a.C + liv[k]], fx);
                }
            }
            if (rvalid) {
                if (want_fs) for (int c = q; c < a.C; c += L) a.fix_sem[ray * a.C + c] = (float)hist[c] * (1.0f / CMP_FIX_SCALE);
                if (want_fi) for (int c = q; c < a.K; c += L) a.fix_inst[ray * a.K + c] = (float)hist[a.C + c] * (1.0f / CMP_FIX_SCALE);
            }
        }

        // ---- softmax mode: per-sample max / denominator of each learned field (extra pass over its rows)
        float mx_s[SOFTMAX ? M : 1], den_s[SOFTMAX ? M : 1], mx_i[SOFTMAX ? M : 1], den_i[SOFTMAX ? M : 1];
        if constexpr (SOFTMAX) {
            auto field_stats = [&](int nch, int ch0, float (&mx)[M], float (&den)[M]) {
#pragma unroll
                for (int k = 0; k < M; ++k) { mx[k] = -INFINITY; den[k] = 0.0f; }
                for (int c = 0; c < nch; ++c) {
                    float v[M];
                    row(a.raw + (int64_t)(ch0 + c) * sc, v);
#pragma unroll
                    for (int k = 0; k < M; ++k) {
                        const float d = v[k] - mx[k], e = expf(-fabsf(d));          // one expf per value (the other factor is expf(0))
                        den[k] = d <= 0.0f ? den[k] + e : den[k] * e + 1.0f;
                        mx[k] = fmaxf(mx[k], v[k]);
                    }
                }
            };
            field_stats(a.C, 4, mx_s, den_s);
            field_stats(a.K, 4 + a.C, mx_i, den_i);
        }

        // ---- phase 2: every composited channel, CMP2_UB rows per batch; a row's registers are re-armed with the row of
        // the next batch as soon as its partial sum is taken
#pragma unroll 1
        for (int q0 = 0; q0 < nq; q0 += CMP2_UB) {
            float r[CMP2_UB];
#pragma unroll
            for (int j = 0; j < CMP2_UB; ++j) {
                const int qq = q0 + j;
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    float v = nxt[j][k];
                    if (qq < 3) v = 1.0f / (1.0f + expf(-v));
                    else if constexpr (SOFTMAX) {
                        const bool fs = (qq - 3) < a.C;
                        v = expf(v - (fs ? mx_s[k] : mx_i[k])) / (fs ? den_s[k] : den_i[k]);
                    }
                    acc = fmaf(w[k], v, acc);
                }
                r[j] = acc;
                const int q2 = qq + CMP2_UB;                     // >= 4 > 3: never an rgb row -> raw channel q2 + 1
                if (q2 < nq) row(a.raw + (int64_t)(q2 + 1) * sc, nxt[j]);
            }
            group_sum_batch<L, CMP2_UB>(r);
            if (writer) {
#pragma unroll
                for (int j = 0; j < CMP2_UB; ++j) {
                    const int qq = q0 + j;
                    if (qq >= nq) continue;
                    if (qq < 3) { if (a.rgb) a.rgb[ray * 3 + qq] = a.white_bkgd ? r[j] + (1.0f - accv) : r[j]; }
                    else if (qq - 3 < a.C) { if (a.sem) a.sem[ray * a.C + (qq - 3)] = r[j]; }
                    else { if (a.inst) a.inst[ray * a.K + (qq - 3 - a.C)] = r[j]; }
                }
            }
        }
    }
}

template <bool SOFTMAX>
static void launch_composite2(int L, int m4, int grid, size_t lds, hipStream_t st, const CompositeArgs& a)
{
#define PNR_C2(LL, MM) hipLaunchKernelGGL((k_composite2<LL, MM, SOFTMAX>), dim3(grid), dim3(256), lds, st, a)
    if (L == 16) { if (m4 == 1) PNR_C2(16, 1); else if (m4 == 2) PNR_C2(16, 2); else if (m4 == 3) PNR_C2(16, 3); else PNR_C2(16, 4); }
    else { if (m4 == 1) PNR_C2(8, 1); else PNR_C2(8, 2); }
#undef PNR_C2
}

template <bool CH_MAJOR, bool SOFTMAX>
static void launch_composite(int sub, int grid, size_t lds, hipStream_t st, const CompositeArgs& a)
{
    switch (sub) {
    case 1: hipLaunchKernelGGL((k_composite<CH_MAJOR, 1, SOFTMAX>), dim3(grid), dim3(256), lds, st, a); break;
    case 2: hipLaunchKernelGGL((k_composite<CH_MAJOR, 2, SOFTMAX>), dim3(grid), dim3(256), lds, st, a); break;
    case 4: hipLaunchKernelGGL((k_composite<CH_MAJOR, 4, SOFTMAX>), dim3(grid), dim3(256), lds, st, a); break;
    case 8: hipLaunchKernelGGL((k_composite<CH_MAJOR, 8, SOFTMAX>), dim3(grid), dim3(256), lds, st, a); break;
    case 16: hipLaunchKernelGGL((k_composite<CH_MAJOR, 16, SOFTMAX>), dim3(grid), dim3(256), lds, st, a); break;
    case 32: hipLaunchKernelGGL((k_composite<CH_MAJOR, 32, SOFTMAX>), dim3(grid), dim3(256), lds, st, a); break;
    default: hipLaunchKernelGGL((k_composite<CH_MAJOR, 64, SOFTMAX>), dim3(grid), dim3(256), lds, st, a); break;
    }
}

// PNR_CMP_VARIANT (A/B tools): 0 = 4-samples-per-lane mapping only, 1 = second mapping for 32 < N <= 64 [default], 2 = for every N > 32
#ifndef PNR_CMP_DEFAULT_VARIANT
#define PNR_CMP_DEFAULT_VARIANT 1
#endif
static int composite_variant()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("PNR_CMP_VARIANT");
        v = e ? atoi(e) : PNR_CMP_DEFAULT_VARIANT;
    }
    return v;
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
    const size_t hist_bytes = (size_t)4 * rpw * (n_sem + n_inst) * sizeof(uint32_t);
    const bool want_fix = (fix_sem && label_sem && n_sem) || (fix_inst && label_inst && n_inst);
    a.use_hist = (want_fix && hist_bytes <= 48 * 1024) ? 1 : 0;
    const size_t lds = a.use_hist ? hist_bytes : 0;
    const bool ch_major = raw_stride_s == 1 && (raw_stride_c % 4) == 0 && (((uintptr_t)raw) & 15) == 0;
    hipStream_t st = (hipStream_t)stream;
    // second mapping (L lanes x 4*m4 consecutive samples per lane): channel-major images with 33..64 samples per ray
    // [variant 2, A/B only: every N > 32]
    const int variant = composite_variant();
    if (ch_major && n_samples > 32 && ((variant >= 1 && n_samples <= 64) || variant >= 2)) {
        const int L = n_samples > 64 ? 16 : 8;
        const int m4 = (n_samples + 4 * L - 1) / (4 * L);
        const int rpw2 = 64 / L;
        const size_t hb = (size_t)4 * rpw2 * (n_sem + n_inst) * sizeof(uint32_t);
        if (!want_fix || hb <= 48 * 1024) {
            const int64_t ng = (n_rays + rpw2 - 1) / rpw2;
            const int grid2 = pnr_grid_cap((ng + 3) / 4, CMP_GRID_PER_CU);
            a.use_hist = want_fix ? 1 : 0;
            if (sem_mode) launch_composite2<true>(L, m4, grid2, want_fix ? hb : 0, st, a); else launch_composite2<false>(L, m4, grid2, want_fix ? hb : 0, st, a);
            PNR_CHECK_LAUNCH("pnr_composite");
            return PNR_OK;
        }
    }
    const int64_t n_groups = (n_rays + rpw - 1) / rpw;
    const int grid = pnr_grid_cap((n_groups + 3) / 4, CMP_GRID_PER_CU);
    if (ch_major) { if (sem_mode) launch_composite<true, true>(sub, grid, lds, st, a); else launch_composite<true, false>(sub, grid, lds, st, a); }
    else { if (sem_mode) launch_composite<false, true>(sub, grid, lds, st, a); else launch_composite<false, false>(sub, grid, lds, st, a); }
    PNR_CHECK_LAUNCH("pnr_composite");
    return PNR_OK;
}


// ------------------------------------------------------------------------------- second half of the fused path
// k_composite_combine: finishes every ray from what the fused MLP epilogue wrote (pnr_mlp_fuse.h) -- the N / 32 per-tile
// records (Q, logit sums) and the N per-sample quadruples (lw, r, g, b):
//   T_k = prod_{k' < k} Q_k',  w_i = T_{i / 32} lw_i,
//   acc / depth / rgb = sum_i w_i {1, z_i, sigmoid(rgb_i)},  fix_x[c] = sum_i w_i [label_i == c],  logits_c = sum_k T_k S_c(k).
// One wave per ray, lane = sample (i = lane + 64 j) for the per-sample part and lane = column for the logits; 26 B per sample of
// traffic at 45 / 32 heads (+ z, + 4 B per labelled field) instead of the raw image's 324 B.  The fixed fields are a
// fixed-point histogram in LDS (integer adds: order-independent, deterministic).
struct CombineArgs {
    const float* rec; int rec_floats; const float4* ps; const float* z; const int32_t* lab_s; const int32_t* lab_i;
    int64_t R; int N, C, K, white_bkgd;
    float *rgb, *depth, *acc, *weights, *sem, *inst, *fix_sem, *fix_inst;
};

__global__ __launch_bounds__(256) void k_composite_combine(CombineArgs a)
{
    __shared__ uint32_t hist_all[4][128];       // C + K <= 128 (pnr_mlp_forward_composite)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t ray = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (ray >= a.R) return;                      // wave-uniform
    const int N = a.N, T = N >> 5, RF = a.rec_floats, C = a.C, K = a.K, CK = C + K;
    const float* __restrict__ rec = a.rec + ray * T * RF;
    const float4* __restrict__ ps = a.ps + ray * N;
    const float* __restrict__ zr = a.z + ray * N;
    const bool want_s = a.lab_s && a.fix_sem && C, want_i = a.lab_i && a.fix_inst && K;
    const int32_t* __restrict__ lsr = want_s ? a.lab_s + ray * N : nullptr;
    const int32_t* __restrict__ lir = want_i ? a.lab_i + ray * N : nullptr;
    // Everything this ray reads is requested up front -- the kernel waits for memory 84 % of its cycles (rocprofv3, round 3),
    // and the weights store below would otherwise fence every later load behind it (the pointers may alias for all hipcc knows)
    float q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = k < T ? rec[k * RF] : 1.0f;
    float4 pv[4];
    float zv[4];
    int lsv[4], liv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        const bool in = i < N;
        pv[j] = in ? ps[i] : float4{0.0f, 0.0f, 0.0f, 0.0f};
        zv[j] = in ? zr[i] : 0.0f;
        lsv[j] = (in && want_s) ? lsr[i] : -1;
        liv[j] = (in && want_i) ? lir[i] : -1;
    }
    float lg[2][8];                              // the logit sums of this lane's columns (c = lane, lane + 64), per tile
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 8; ++k) lg[h][k] = (k < T && lane + 64 * h < CK) ? rec[k * RF + PNR_FUSE_REC_LOGITS + lane + 64 * h] : 0.0f;
    float Tk[8];
    float t = 1.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        Tk[k] = t;
        if (k < T) t *= q[k];
    }
    uint32_t* hist = hist_all[wv];
    if (want_s || want_i)
        for (int c = lane; c < CK; c += 64) hist[c] = 0;
    float r5[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + 64 * j;
        if (i < N) {
            const float4 p = pv[j];
            float tk = Tk[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) tk = (i >> 5) == k ? Tk[k] : tk;
            const float w = tk * p.x;
            r5[0] += w;
            r5[1] = fmaf(w, zv[j], r5[1]);
            r5[2] = fmaf(w, 1.0f / (1.0f + expf(-p.y)), r5[2]);
            r5[3] = fmaf(w, 1.0f / (1.0f + expf(-p.z)), r5[3]);
            r5[4] = fmaf(w, 1.0f / (1.0f + expf(-p.w)), r5[4]);
            if (a.weights) a.weights[ray * N + i] = w;
            if (want_s || want_i) {
                const uint32_t fx = (uint32_t)(w * PNR_FUSE_FIX_SCALE + 0.5f);
                if (want_s) { const int l = lsv[j]; if (l >= 0 && l < C) atomicAdd(&hist[l], fx); }
                if (want_i) { const int l = liv[j]; if (l >= 0 && l < K) atomicAdd(&hist[C + l], fx); }
            }
        }
    }
    group_sum_batch<64, 5>(r5);
    if (lane == 0) {
        if (a.acc) a.acc[ray] = r5[0];
        if (a.depth) a.depth[ray] = r5[1];
    }
    if (lane < 3 && a.rgb) {
        const float v = lane == 0 ? r5[2] : lane == 1 ? r5[3] : r5[4];
        a.rgb[ray * 3 + lane] = a.white_bkgd ? v + (1.0f - r5[0]) : v;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = lane + 64 * h;
        if (c >= CK) continue;
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k < T) v = fmaf(Tk[k], lg[h][k], v);
        if (c < C) { if (a.sem) a.sem[ray * C + c] = v; }
        else if (a.inst) a.inst[ray * K + (c - C)] = v;
        // LDS operations of one wave execute in order and only this wave touches its histogram: no barrier
        if (c < C) { if (want_s) a.fix_sem[ray * C + c] = (float)hist[c] * (1.0f / PNR_FUSE_FIX_SCALE); }
        else if (want_i) a.fix_inst[ray * K + (c - C)] = (float)hist[c] * (1.0f / PNR_FUSE_FIX_SCALE);
    }
}

int pnr_composite_combine_launch(const float* rec, int rec_floats, const float4* ps, const float* z, const int32_t* label_sem,
                                 const int32_t* label_inst, int64_t R, int N, int C, int K, int white_bkgd, float* rgb, float* depth,
                                 float* acc, float* weights, float* sem, float* inst, float* fix_sem, float* fix_inst, hipStream_t st)
{
    CombineArgs a;
    a.rec = rec; a.rec_floats = rec_floats; a.ps = ps; a.z = z; a.lab_s = label_sem; a.lab_i = label_inst;
    a.R = R; a.N = N; a.C = C; a.K = K; a.white_bkgd = white_bkgd;
    a.rgb = rgb; a.depth = depth; a.acc = acc; a.weights = weights; a.sem = sem; a.inst = inst; a.fix_sem = fix_sem; a.fix_inst = fix_inst;
    const int64_t blocks = (R + 3) / 4;
    hipLaunchKernelGGL(k_composite_combine, dim3((unsigned)blocks), dim3(256), 0, st, a);
    PNR_CHECK_LAUNCH("pnr_mlp_forward_composite (combine)");
    return PNR_OK;
}
