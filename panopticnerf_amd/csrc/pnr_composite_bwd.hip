// K4-bwd: backward of raw2outputs (the reverse scan).  HBM-bound: reads raw
// once, writes d_raw once.  Same lane mapping as the forward kernel (pnr_composite.hip): a ray is
// owned by SUB = pow2ceil(N/4) lanes, 4 consecutive samples per lane; channel-major images.
//
//   w_i = alpha_i T_i,  T_i = prod_{j<i} (1 - alpha_j + 1e-10),  alpha_i = 1 - exp(-relu(sigma_i) delta_i)
//   out_c = sum_i w_i v_ci        (v = sigmoid(raw) for rgb, raw for the logit fields, z for depth, 1 for acc)
//   G_i   = dL/dw_i = sum_c g_c v_ci + g_depth z_i + g_acc + g_w_i
//   dL/dv_ci = w_i g_c
//   dL/dsigma_i = [sigma_i > 0] delta_i (1 - alpha_i) ( G_i T_i - S_i / (1 - alpha_i + 1e-10) ),
//                 S_i = sum_{k>i} G_k w_k      (segmented reverse scan over the ray)
// sample_pdf's output is detached in the reference, so no gradient flows into z.
// Fixed (bbox-prior) fields have no learnable input.  Softmax-composited fields (sem_mode 1) are
// not differentiated here (forward-only option); the caller is told so.
#include "pnr_common.h"

struct CompositeBwdArgs {
    const float* raw; int64_t sc;          // channel-major: element (s, c) at raw[c*sc + s]
    const float* z; const float* rays; const float* noise;
    int64_t R; int N, C, K;
    const float *g_rgb, *g_depth, *g_acc, *g_sem, *g_inst, *g_w;   // upstream grads (any may be null)
    float* d_raw;                          // (4+C+K, R*N) channel-major
    // fixed (bbox-prior) fields: fix_x[c] = sum_i w_i [label_i == c]  =>  dL/dw_i += g_fix_x[label_i]
    const int32_t *label_sem, *label_inst; // (R,N), -1 = none; needed by g_fix_* and ce_*
    const float *g_fix_sem, *g_fix_inst;   // (R,C), (R,K) or null
    // per-sample 3D cross-entropy of the learned logits against the bbox labels (pnr_ce3d is its forward):
    // d_raw[c][s] += *ce_x * (softmax_c(raw_x[:, s]) - [c == label_s])  for label_s >= 0
    const float *ce_sem, *ce_inst;         // device scalars (upstream gradient * weight / count) or null
    int sem_mode;                          // 0: logits are composited; 1: softmax(logits) per sample is composited
};

struct f4 { float v[4]; };
// one row's arithmetic at a time: without the fence hipcc interleaves the expf chains of a whole batch of rows (8 rows x 4 samples)
// for instruction-level parallelism and needs ~175 registers for it -- at the 128 the launch bound grants it spilled ~45 of them
#ifndef BWD_FENCE
#define BWD_FENCE 1
#endif
#if BWD_FENCE
#define BWD_ROW_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define BWD_ROW_FENCE ((void)0)
#endif
// Per ray-group width (SUB lanes per ray): waves per SIMD the launch bound asks for, logit rows per pipelined batch (two batches
// resident), rows in flight in the log-sum-exp / dot passes.  One ray per wave (N > 64: SUB >= 32) means R waves in all -- 4096 at the
// training batch, 16 per CU -- so the kernel is capped at 128 VGPRs (4 waves per SIMD: the launch is resident at once; at its
// natural 149 registers it ran in two rounds) and keeps little in flight; several rays per wave (N <= 64) means few waves and
// nothing to gain from the cap, so those instances keep 3 waves per SIMD and deeper batches.  Same box, 4096 rays, 45 + 32 fields,
// tools/composite_bwd_time.py (d_raw bit-identical throughout): N = 192  229 us before -> 168 (cap 4, 2 + 4 rows) / 176-179 (cap 4,
// 4 + 4 / 4 + 8 rows) / 192 (no cap, pipelined only);  N = 64  95 -> 77 (no cap, 4 + 8) / 84-86 (capped).
#ifndef BWD_WAVES
#define BWD_WAVES(SUB) ((SUB) >= 32 ? 4 : 3)
#endif
#ifndef BWD_PB_OF
#define BWD_PB_OF(SUB) ((SUB) >= 32 ? 2 : 4)
#endif
#ifndef BWD_CB_OF
#define BWD_CB_OF(SUB) ((SUB) >= 32 ? 4 : 8)
#endif

__device__ __forceinline__ f4 ld4(const float* p, bool active)
{
    f4 o;
    if (!active) { o.v[0] = o.v[1] = o.v[2] = o.v[3] = 0.0f; return o; }
    const float4 t = *reinterpret_cast<const float4*>(p);
    o.v[0] = t.x; o.v[1] = t.y; o.v[2] = t.z; o.v[3] = t.w;
    return o;
}
// Row access as (wave-uniform row pointer) + (32-bit per-lane byte offset): hipcc then selects the scalar-base form of
// global_load / global_store (SGPR pair + one shared VGPR offset) instead of materialising a 64-bit VGPR address per row in
// flight -- two registers per row, which is what pushed this kernel past 128 VGPRs (3 waves per SIMD instead of 4).
__device__ __forceinline__ f4 ld4o(const float* row, uint32_t ob, bool active)
{
    return ld4(reinterpret_cast<const float*>(reinterpret_cast<const char*>(row) + ob), active);
}
__device__ __forceinline__ void st4(float* p, const f4& v, bool active);
__device__ __forceinline__ void st4o(float* row, uint32_t ob, const f4& v, bool active)
{
    st4(reinterpret_cast<float*>(reinterpret_cast<char*>(row) + ob), v, active);
}
__device__ __forceinline__ void st4(float* p, const f4& v, bool active)
{
    if (active) *reinterpret_cast<float4*>(p) = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]);
}

// SM: softmax compositing (sem_mode 1) -- a template parameter so that the default (logit compositing) instance carries neither
// the dot products nor their branches in its register budget
template <int SUB, bool SM>
__global__ __launch_bounds__(256, BWD_WAVES(SUB)) void k_composite_bwd(CompositeBwdArgs a)
{
    constexpr int RPW = 64 / SUB;
    constexpr int CB = BWD_CB_OF(SUB);      // channel rows in flight per lane (log-sum-exp / dot passes)
    constexpr int BWD_PB = BWD_PB_OF(SUB);  // logit rows per pipelined batch of the main pass
    const int lane = threadIdx.x & 63;
    const int N = a.N, nq4 = N >> 2;
    const int q = lane & (SUB - 1), g = lane / SUB;
    const int64_t wave_global = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t n_groups = (a.R + RPW - 1) / RPW;

    for (int64_t grp = wave_global; grp < n_groups; grp += n_waves) {
        const int64_t ray = grp * RPW + g;
        const bool active = (ray < a.R) && (q < nq4);
        const int64_t rayc = ray < a.R ? ray : a.R - 1;
        const int64_t s0 = rayc * N + (active ? 4 * q : 0);

        // ---- recompute alpha, T, w (as the forward kernel does)
        const uint32_t ob = (uint32_t)s0 * 4u;          // byte offset of the lane's first sample in a row (R * N < 2^30: the launcher checks)
        const f4 zz = ld4o(a.z, ob, active);
        f4 sg = ld4o(a.raw + 3 * a.sc, ob, active);
        if (a.noise) { const f4 nz = ld4o(a.noise, ob, active); for (int k = 0; k < 4; ++k) sg.v[k] += nz.v[k]; }
        const float dx = a.rays[rayc * 8 + 3], dy = a.rays[rayc * 8 + 4], dz = a.rays[rayc * 8 + 5];
        const float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float znext = __shfl_down(zz.v[0], 1, 64);
        f4 alpha, dist, Tl;            // Tl: product of this lane's earlier factors
        float P = 1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * q + k;
            const float zn = (k < 3) ? zz.v[k + 1] : znext;
            float d = (i + 1 < N) ? (zn - zz.v[k]) : 1e10f;
            d *= dn;
            dist.v[k] = d;
            alpha.v[k] = 1.0f - expf(-(fmaxf(sg.v[k], 0.0f) * d));
            Tl.v[k] = P;
            P *= (1.0f - alpha.v[k]) + 1e-10f;
        }
        if (!active) P = 1.0f;
        float x = P;
#pragma unroll
        for (int d = 1; d < SUB; d <<= 1) {
            const float y = __shfl_up(x, d, 64);
            if (q >= d) x *= y;
        }
        float excl = __shfl_up(x, 1, 64);
        if (q == 0) excl = 1.0f;
        f4 T, w;
#pragma unroll
        for (int k = 0; k < 4; ++k) { T.v[k] = Tl.v[k] * excl; w.v[k] = active ? alpha.v[k] * T.v[k] : 0.0f; }

        // ---- G_i and the per-channel value gradients (one pass over the channel rows)
        f4 G;
        {
            const float gd = a.g_depth ? a.g_depth[rayc] : 0.0f, ga = a.g_acc ? a.g_acc[rayc] : 0.0f;
            const f4 gw = a.g_w ? ld4o(a.g_w, ob, active) : f4{{0, 0, 0, 0}};
#pragma unroll
            for (int k = 0; k < 4; ++k) G.v[k] = fmaf(gd, zz.v[k], ga) + gw.v[k];
        }
        int ls[4] = {-1, -1, -1, -1}, li[4] = {-1, -1, -1, -1};
        if (a.label_sem && active) { const int4 t = *reinterpret_cast<const int4*>(a.label_sem + s0); ls[0] = t.x; ls[1] = t.y; ls[2] = t.z; ls[3] = t.w; }
        if (a.label_inst && active) { const int4 t = *reinterpret_cast<const int4*>(a.label_inst + s0); li[0] = t.x; li[1] = t.y; li[2] = t.z; li[3] = t.w; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (a.g_fix_sem && ls[k] >= 0 && ls[k] < a.C) G.v[k] += a.g_fix_sem[rayc * a.C + ls[k]];
            if (a.g_fix_inst && li[k] >= 0 && li[k] < a.K) G.v[k] += a.g_fix_inst[rayc * a.K + li[k]];
        }
        // 3D cross-entropy: per-sample log-sum-exp of each learned field (one extra pass over its channel rows)
        const float ces = (a.ce_sem && a.label_sem) ? *a.ce_sem : 0.0f, cei = (a.ce_inst && a.label_inst) ? *a.ce_inst : 0.0f;
        f4 mx_s, den_s, mx_i, den_i;
        auto lse = [&](int nch, int ch0, f4& mx, f4& den) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { mx.v[k] = -INFINITY; den.v[k] = 0.0f; }
            // CB channel rows requested before the first is consumed: with 4 waves per SIMD and one 1 KiB load in flight per
            // wave the pass ran at ~2 TB/s (4096 rays = 16 waves per CU)
            for (int c0 = 0; c0 < nch; c0 += CB) {
                f4 v[CB];
#pragma unroll
                for (int j = 0; j < CB; ++j) v[j] = ld4o(a.raw + (int64_t)(ch0 + (c0 + j < nch ? c0 + j : nch - 1)) * a.sc, ob, active);
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    if (c0 + j >= nch) break;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // online log-sum-exp with ONE expf per value: of the two factors expf(mx - m2), expf(v - m2) one is always
                        // expf(0) = 1 (same bits as the two-expf form)
                        const float d = v[j].v[k] - mx.v[k], e = expf(-fabsf(d));
                        den.v[k] = d <= 0.0f ? den.v[k] + e : den.v[k] * e + 1.0f;
                        mx.v[k] = fmaxf(mx.v[k], v[j].v[k]);
                    }
                    BWD_ROW_FENCE;
                }
            }
        };
        // The 3D term touches labelled samples only, and most rays cross no box at all: a wave none of whose samples carries a
        // label skips the pass (wave-uniform; mx / den are then never read -- the `lab < 0` guard below).  The pass is a read of
        // every logit row plus two expf per value: at 4096 rays (16 waves per CU, one ray group each) it was half the kernel.
        bool lab_s = false, lab_i = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) { lab_s |= ls[k] >= 0 && ls[k] < a.C; lab_i |= li[k] >= 0 && li[k] < a.K; }
        const bool wave_s = __builtin_amdgcn_ballot_w64(lab_s) != 0, wave_i = __builtin_amdgcn_ballot_w64(lab_i) != 0;
        if ((ces != 0.0f && wave_s) || (SM && a.g_sem)) lse(a.C, 4, mx_s, den_s);
        if ((cei != 0.0f && wave_i) || (SM && a.g_inst)) lse(a.K, 4 + a.C, mx_i, den_i);
        // softmax compositing: map_c = sum_i w_i s_{i,c}, s = softmax(x_i).  Needs dot_i = sum_c g_c s_{i,c}:
        //   dL/dw_i += dot_i,   d x_{i,c} = w_i s_{i,c} (g_c - dot_i)
        f4 dot_s = {{0, 0, 0, 0}}, dot_i = {{0, 0, 0, 0}};
        auto gdot = [&](int nch, int ch0, const float* gp, const f4& mx, const f4& den, f4& dot) {
            for (int c0 = 0; c0 < nch; c0 += CB) {
                f4 v[CB];
#pragma unroll
                for (int j = 0; j < CB; ++j) v[j] = ld4o(a.raw + (int64_t)(ch0 + (c0 + j < nch ? c0 + j : nch - 1)) * a.sc, ob, active);
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    if (c0 + j >= nch) break;
                    const float gc = gp[rayc * nch + c0 + j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) dot.v[k] = fmaf(gc, expf(v[j].v[k] - mx.v[k]) / den.v[k], dot.v[k]);
                }
            }
        };
        if (SM && a.g_sem) gdot(a.C, 4, a.g_sem, mx_s, den_s, dot_s);
        if (SM && a.g_inst) gdot(a.K, 4 + a.C, a.g_inst, mx_i, den_i, dot_i);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float gc = a.g_rgb ? a.g_rgb[rayc * 3 + c] : 0.0f;
            const f4 r = ld4o(a.raw + (int64_t)c * a.sc, ob, active);
            f4 dr;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = 1.0f / (1.0f + expf(-r.v[k]));
                G.v[k] = fmaf(gc, v, G.v[k]);
                dr.v[k] = w.v[k] * gc * v * (1.0f - v);
            }
            st4o(a.d_raw + (int64_t)c * a.sc, ob, dr, active);
        }
        const int CK = a.C + a.K;
        // The logit rows, BWD_PB rows per batch, software-pipelined: the next batch's loads are issued BEFORE this batch's
        // stores.  vmcnt is one in-order counter for loads and stores on this part, so a load issued behind a store is not seen
        // as landed until the store is acknowledged -- with the loads in front, the wait the compiler emits is a counted one
        // that leaves the stores in flight.  (One ray group per wave and 16 waves per CU at 4096 rays: the kernel is a chain of
        // memory round trips, not a bandwidth problem.)
        auto fetch_rows = [&](int c0, f4 (&rb)[BWD_PB], float (&gcb)[BWD_PB]) {
#pragma unroll
            for (int j = 0; j < BWD_PB; ++j) {
                const int c = c0 + j < CK ? c0 + j : CK - 1;
                const bool is_s = c < a.C;
                const float* gp = is_s ? a.g_sem : a.g_inst;
                gcb[j] = gp ? gp[rayc * (is_s ? a.C : a.K) + (is_s ? c : c - a.C)] : 0.0f;
                rb[j] = ld4o(a.raw + (int64_t)(4 + c) * a.sc, ob, active);
            }
        };
        auto do_rows = [&](int c0, const f4 (&rb)[BWD_PB], const float (&gcb)[BWD_PB]) {
#pragma unroll
            for (int j = 0; j < BWD_PB; ++j) {
                const int c = c0 + j;
                if (c >= CK) break;
                const bool is_s = c < a.C;
                const float* gp = is_s ? a.g_sem : a.g_inst;
                const float gc = gcb[j];
                const f4 r = rb[j];
                f4 dr;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (SM && gp) {
                        const float sc = expf(r.v[k] - (is_s ? mx_s.v[k] : mx_i.v[k])) / (is_s ? den_s.v[k] : den_i.v[k]);
                        dr.v[k] = w.v[k] * sc * (gc - (is_s ? dot_s.v[k] : dot_i.v[k]));
                    } else {
                        G.v[k] = fmaf(gc, r.v[k], G.v[k]);
                        dr.v[k] = w.v[k] * gc;
                    }
                }
                const float ce = is_s ? ces : cei;
                if (ce != 0.0f) {
                    const int cc = is_s ? c : c - a.C;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int lab = is_s ? ls[k] : li[k];
                        if (lab < 0 || lab >= (is_s ? a.C : a.K)) continue;
                        const float pc = expf(r.v[k] - (is_s ? mx_s.v[k] : mx_i.v[k])) / (is_s ? den_s.v[k] : den_i.v[k]);
                        dr.v[k] += ce * (pc - (lab == cc ? 1.0f : 0.0f));
                    }
                }
                st4o(a.d_raw + (int64_t)(4 + c) * a.sc, ob, dr, active);
                BWD_ROW_FENCE;
            }
        };
        {
            f4 rbA[BWD_PB], rbB[BWD_PB];
            float gA[BWD_PB], gB[BWD_PB];
            if (CK > 0) fetch_rows(0, rbA, gA);
            for (int c0 = 0; c0 < CK; c0 += 2 * BWD_PB) {
                if (c0 + BWD_PB < CK) fetch_rows(c0 + BWD_PB, rbB, gB);
                do_rows(c0, rbA, gA);
                if (c0 + BWD_PB >= CK) break;
                if (c0 + 2 * BWD_PB < CK) fetch_rows(c0 + 2 * BWD_PB, rbA, gA);
                do_rows(c0 + BWD_PB, rbB, gB);
            }
        }

        if (SM) {
#pragma unroll
            for (int k = 0; k < 4; ++k) G.v[k] += dot_s.v[k] + dot_i.v[k];
        }
        // ---- S_i = sum_{k>i} G_k w_k : lane-local suffix, then a segmented reverse (suffix) scan over lanes
        f4 gwk;
#pragma unroll
        for (int k = 0; k < 4; ++k) gwk.v[k] = active ? G.v[k] * w.v[k] : 0.0f;
        const float lane_tot = (gwk.v[0] + gwk.v[1]) + (gwk.v[2] + gwk.v[3]);
        float y = lane_tot;                 // inclusive suffix sum over lanes q..SUB-1 of the group
#pragma unroll
        for (int d = 1; d < SUB; d <<= 1) {
            const float t = __shfl_down(y, d, 64);
            if (q + d < SUB) y += t;
        }
        float after = __shfl_down(y, 1, 64);     // sum over the lanes after this one
        if (q == SUB - 1) after = 0.0f;
        f4 S;
        S.v[3] = after;
        S.v[2] = S.v[3] + gwk.v[3];
        S.v[1] = S.v[2] + gwk.v[2];
        S.v[0] = S.v[1] + gwk.v[1];
        f4 ds;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float om = 1.0f - alpha.v[k];
            const float dalpha = G.v[k] * T.v[k] - S.v[k] / (om + 1e-10f);
            ds.v[k] = (sg.v[k] > 0.0f) ? dist.v[k] * om * dalpha : 0.0f;
        }
        st4o(a.d_raw + 3 * a.sc, ob, ds, active);
    }
}

PNR_EXPORT int pnr_composite_backward2(const float* raw, int64_t raw_stride_c, const float* z, const float* rays,
                                       const float* noise, int64_t n_rays, int n_samples, int n_sem, int n_inst,
                                       const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_sem,
                                       const float* g_inst, const float* g_weights, const int32_t* label_sem,
                                       const int32_t* label_inst, const float* g_fix_sem, const float* g_fix_inst,
                                       const float* ce_sem, const float* ce_inst, float* d_raw, void* stream);

PNR_EXPORT int pnr_composite_backward(const float* raw, int64_t raw_stride_c, const float* z, const float* rays,
                                      const float* noise, int64_t n_rays, int n_samples, int n_sem, int n_inst,
                                      const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_sem,
                                      const float* g_inst, const float* g_weights, float* d_raw, void* stream)
{
    return pnr_composite_backward2(raw, raw_stride_c, z, rays, noise, n_rays, n_samples, n_sem, n_inst, g_rgb, g_depth, g_acc,
                                   g_sem, g_inst, g_weights, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_raw, stream);
}

static int composite_backward_impl(const float* raw, int64_t raw_stride_c, const float* z, const float* rays,
                                   const float* noise, int64_t n_rays, int n_samples, int n_sem, int n_inst, int sem_mode,
                                   const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_sem,
                                   const float* g_inst, const float* g_weights, const int32_t* label_sem,
                                   const int32_t* label_inst, const float* g_fix_sem, const float* g_fix_inst,
                                   const float* ce_sem, const float* ce_inst, float* d_raw, void* stream);

PNR_EXPORT int pnr_composite_backward2(const float* raw, int64_t raw_stride_c, const float* z, const float* rays,
                                       const float* noise, int64_t n_rays, int n_samples, int n_sem, int n_inst,
                                       const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_sem,
                                       const float* g_inst, const float* g_weights, const int32_t* label_sem,
                                       const int32_t* label_inst, const float* g_fix_sem, const float* g_fix_inst,
                                       const float* ce_sem, const float* ce_inst, float* d_raw, void* stream)
{
    return composite_backward_impl(raw, raw_stride_c, z, rays, noise, n_rays, n_samples, n_sem, n_inst, 0, g_rgb, g_depth, g_acc, g_sem,
                                   g_inst, g_weights, label_sem, label_inst, g_fix_sem, g_fix_inst, ce_sem, ce_inst, d_raw, stream);
}

PNR_EXPORT int pnr_composite_backward3(const float* raw, int64_t raw_stride_c, const float* z, const float* rays,
                                       const float* noise, int64_t n_rays, int n_samples, int n_sem, int n_inst, int sem_mode,
                                       const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_sem,
                                       const float* g_inst, const float* g_weights, const int32_t* label_sem,
                                       const int32_t* label_inst, const float* g_fix_sem, const float* g_fix_inst,
                                       const float* ce_sem, const float* ce_inst, float* d_raw, void* stream)
{
    PNR_REQUIRE(sem_mode == 0 || sem_mode == 1, "pnr_composite_backward: sem_mode must be 0 (logits) or 1 (softmax)");
    return composite_backward_impl(raw, raw_stride_c, z, rays, noise, n_rays, n_samples, n_sem, n_inst, sem_mode, g_rgb, g_depth, g_acc,
                                   g_sem, g_inst, g_weights, label_sem, label_inst, g_fix_sem, g_fix_inst, ce_sem, ce_inst, d_raw, stream);
}

static int composite_backward_impl(const float* raw, int64_t raw_stride_c, const float* z, const float* rays,
                                   const float* noise, int64_t n_rays, int n_samples, int n_sem, int n_inst, int sem_mode,
                                   const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_sem,
                                   const float* g_inst, const float* g_weights, const int32_t* label_sem,
                                   const int32_t* label_inst, const float* g_fix_sem, const float* g_fix_inst,
                                   const float* ce_sem, const float* ce_inst, float* d_raw, void* stream)
{
    PNR_REQUIRE(n_samples >= 4 && n_samples <= 256 && (n_samples % 4) == 0,
                "pnr_composite_backward: n_samples=%d must be a multiple of 4 in [4,256]", n_samples);
    if (n_rays <= 0) return PNR_OK;
    PNR_REQUIRE(n_rays * (int64_t)n_samples < ((int64_t)1 << 30), "pnr_composite_backward: R*N=%lld must stay below 2^30 (32-bit byte "
                "offsets inside a channel row): split the batch", (long long)(n_rays * (int64_t)n_samples));
    PNR_REQUIRE(raw && z && rays && d_raw, "pnr_composite_backward: null pointer");
    PNR_REQUIRE((raw_stride_c % 4) == 0 && (((uintptr_t)raw | (uintptr_t)d_raw | (uintptr_t)z | (uintptr_t)noise |
                                            (uintptr_t)g_weights | (uintptr_t)label_sem | (uintptr_t)label_inst) & 15) == 0,
                "pnr_composite_backward: channel-major, 16-byte aligned images required");
    PNR_REQUIRE((!g_fix_sem && !ce_sem) || label_sem, "pnr_composite_backward: g_fix_sem / ce_sem need label_sem");
    PNR_REQUIRE((!g_fix_inst && !ce_inst) || label_inst, "pnr_composite_backward: g_fix_inst / ce_inst need label_inst");
    CompositeBwdArgs a;
    a.raw = raw; a.sc = raw_stride_c; a.z = z; a.rays = rays; a.noise = noise; a.R = n_rays; a.N = n_samples;
    a.C = n_sem; a.K = n_inst; a.g_rgb = g_rgb; a.g_depth = g_depth; a.g_acc = g_acc; a.g_sem = g_sem; a.g_inst = g_inst;
    a.g_w = g_weights; a.d_raw = d_raw;
    a.label_sem = label_sem; a.label_inst = label_inst; a.g_fix_sem = g_fix_sem; a.g_fix_inst = g_fix_inst;
    a.ce_sem = ce_sem; a.ce_inst = ce_inst; a.sem_mode = sem_mode;
    int sub = 1;
    while (sub < n_samples / 4) sub <<= 1;
    const int rpw = 64 / sub;
    const int64_t n_groups = (n_rays + rpw - 1) / rpw;
    const int grid = pnr_grid_cap((n_groups + 3) / 4, 8);
    hipStream_t st = (hipStream_t)stream;
    switch (sub) {
    case 1: if (a.sem_mode) hipLaunchKernelGGL((k_composite_bwd<1, true>), dim3(grid), dim3(256), 0, st, a); else hipLaunchKernelGGL((k_composite_bwd<1, false>), dim3(grid), dim3(256), 0, st, a); break;
    case 2: if (a.sem_mode) hipLaunchKernelGGL((k_composite_bwd<2, true>), dim3(grid), dim3(256), 0, st, a); else hipLaunchKernelGGL((k_composite_bwd<2, false>), dim3(grid), dim3(256), 0, st, a); break;
    case 4: if (a.sem_mode) hipLaunchKernelGGL((k_composite_bwd<4, true>), dim3(grid), dim3(256), 0, st, a); else hipLaunchKernelGGL((k_composite_bwd<4, false>), dim3(grid), dim3(256), 0, st, a); break;
    case 8: if (a.sem_mode) hipLaunchKernelGGL((k_composite_bwd<8, true>), dim3(grid), dim3(256), 0, st, a); else hipLaunchKernelGGL((k_composite_bwd<8, false>), dim3(grid), dim3(256), 0, st, a); break;
    case 16: if (a.sem_mode) hipLaunchKernelGGL((k_composite_bwd<16, true>), dim3(grid), dim3(256), 0, st, a); else hipLaunchKernelGGL((k_composite_bwd<16, false>), dim3(grid), dim3(256), 0, st, a); break;
    case 32: if (a.sem_mode) hipLaunchKernelGGL((k_composite_bwd<32, true>), dim3(grid), dim3(256), 0, st, a); else hipLaunchKernelGGL((k_composite_bwd<32, false>), dim3(grid), dim3(256), 0, st, a); break;
    default: if (a.sem_mode) hipLaunchKernelGGL((k_composite_bwd<64, true>), dim3(grid), dim3(256), 0, st, a); else hipLaunchKernelGGL((k_composite_bwd<64, false>), dim3(grid), dim3(256), 0, st, a); break;
    }
    PNR_CHECK_LAUNCH("pnr_composite_backward");
    return PNR_OK;
}
