// K1 / K1b / K2 / K5: per-ray sample generation, Embedder, sample_pdf + merge, bbox prior.
// gfx950 only.  The sampler, sample_pdf and bbox kernels are BIT-EXACT restatements of
// oracle/pnr_oracle.c (pnro_stratified / pnro_points / pnro_sample_pdf / pnro_bbox_hits /
// pnro_sample_labels), which since round 3 pins TORCH'S OWN op order (linspace two-sided, sum in
// ATen's vector order, cumsum in double): same operations in the same order, one rounding each; the
// file is compiled with -ffp-contract=off and the pragma below so no mul+add pair is fused.
// Reference functions these replace (the reference source
// is not in the mount, include/pnr.h explains the citation form).
#include "pnr_common.h"
#include <stdlib.h>
#include <string.h>

#pragma clang fp contract(off)

// torch.linspace(0, 1, N)[i] as ATen's CPU kernel computes it (two-sided; the upper half from the end point with one
// rounding): oracle/pnr_oracle.c::pnro_linspace01, measured bit-exact against torch.  An explicit fmaf -- the file is built
// with contraction off.
__device__ __forceinline__ float pnr_linspace01(int i, int N)
{
    if (N <= 1) return 0.0f;
    const float step = 1.0f / (float)(N - 1);
    if (i < N / 2) return step * (float)i;
    return fmaf(-step, (float)(N - 1 - i), 1.0f);
}

// ------------------------------------------------------------------------------- a3
__device__ __forceinline__ float strat_z(float nr, float fr, int i, int N, int lindisp)
{
    const float t = pnr_linspace01(i, N);
    const float omt = 1.0f - t;
    if (!lindisp) {
        const float a = nr * omt, b = fr * t;
        return a + b;
    }
    const float a = (1.0f / nr) * omt, b = (1.0f / fr) * t;
    return 1.0f / (a + b);
}

// sample i of a ray's N: the stratified depth, jittered inside its bin by *t when t is given (k_stratified, k_ray_setup)
__device__ __forceinline__ float strat_sample(float nr, float fr, int i, int N, int lindisp, const float* t)
{
    const float zi = strat_z(nr, fr, i, N, lindisp);
    if (!t) return zi;
    const float z0 = strat_z(nr, fr, 0, N, lindisp), zl = strat_z(nr, fr, N - 1, N, lindisp);
    const float lo = (i == 0) ? z0 : 0.5f * (zi + strat_z(nr, fr, i - 1, N, lindisp));
    const float up = (i == N - 1) ? zl : 0.5f * (strat_z(nr, fr, i + 1, N, lindisp) + zi);
    const float w = up - lo;
    const float m = w * *t;
    return lo + m;
}

// Ray generation: one thread per ray, two float4 stores; write-bound (32 B/ray).
struct GenRaysArgs { float fx, fy, cx, cy; float c2w[12]; int width; float near_, far_; const int32_t* pix; int64_t R; float* rays; };
__global__ __launch_bounds__(256) void k_gen_rays(const GenRaysArgs a)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < a.R; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = a.pix ? (int64_t)a.pix[r] : r;
        const int j = (int)(p / a.width), i = (int)(p - (int64_t)j * a.width);
        const float x = ((float)i - a.cx) / a.fx;
        const float y = ((float)j - a.cy) / a.fy;
        float d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float u = a.c2w[k * 4 + 0] * x, v = a.c2w[k * 4 + 1] * y;
            d[k] = (u + v) + a.c2w[k * 4 + 2];
        }
        float4* o = reinterpret_cast<float4*>(a.rays + r * 8);
        o[0] = make_float4(a.c2w[3], a.c2w[7], a.c2w[11], d[0]);
        o[1] = make_float4(d[1], d[2], a.near_, a.far_);
    }
}

// One thread per sample; HBM-bound: reads 8 B/ray amortised (+4 B t_rand), writes 4 B.
__global__ __launch_bounds__(256) void k_stratified(const float* __restrict__ rays, int64_t R, int N,
                                                     int lindisp, const float* __restrict__ t_rand,
                                                     float* __restrict__ z_out)
{
    const int64_t total = R * N;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = s / N;
        const int i = (int)(s - r * N);
        z_out[s] = strat_sample(rays[r * 8 + 6], rays[r * 8 + 7], i, N, lindisp, t_rand ? t_rand + s : nullptr);
    }
}

__global__ __launch_bounds__(256) void k_points(const float* __restrict__ rays, const float* __restrict__ z,
                                                 int64_t R, int N, float* __restrict__ pts)
{
    const int64_t total = R * N * 3;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = e / 3;
        const int a = (int)(e - s * 3);
        const int64_t r = s / N;
        const float m = rays[r * 8 + 3 + a] * z[s];
        pts[e] = rays[r * 8 + a] + m;
    }
}

// ------------------------------------------------------------------------------- a4
// One thread per OUTPUT element so that the (n, 3+6L) sample-major image is written fully
// coalesced.  Trig-bound (cdna_hip_programming.md App. B): accurate sinf/cosf (ocml) because
// arguments reach 2^(L-1)*|x|.  The fused MLP kernel computes gamma() in registers instead.
__global__ __launch_bounds__(256) void k_embed(const float* __restrict__ x, int64_t n, int L,
                                                float* __restrict__ out)
{
    const int E = 3 + 6 * L;
    const int64_t total = n * E;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = e / E;
        const int c = (int)(e - s * E);
        float v;
        if (c < 3) {
            v = x[s * 3 + c];
        } else {
            const int k = (c - 3) / 6, j = (c - 3) - 6 * k;
            const int a = j % 3;
            const float arg = x[s * 3 + a] * ldexpf(1.0f, k);
            v = (j < 3) ? sinf(arg) : cosf(arg);
        }
        out[e] = v;
    }
}

// ------------------------------------------------------------------------------- a7
// One 64-lane workgroup (= one wavefront) per ray.  The CDF is accumulated by lane 0 in the
// oracle's sequential fp32 order (the bit-exact index requirement, SURVEY.md section 7 "hard
// parts"); everything else is lane-parallel: bins, pdf, one upper_bound per u, and a bitonic
// sort of the Nc+Nf union in LDS (a rank-based merge when the new samples are already ascending).  LDS: 7 KiB.
// the kept interval a sample at depth zz takes its label from: the one with the smallest t_in among those containing it (-1: none)
template <class F>
__device__ __forceinline__ int label_hit(float zz, int cnt, F interval)
{
    int best = -1;
    float bt = 0.0f;
    for (int h = 0; h < cnt; ++h) {
        float ti, to;
        interval(h, ti, to);
        if (ti <= zz && zz <= to && (best < 0 || ti < bt)) {
            best = h;
            bt = ti;
        }
    }
    return best;
}

// With lab.label_sem set the wave that merged a ray's samples also labels them (k_sample_labels' rule on the sorted z it still
// holds in LDS): the fine level's labels without a second pass over z (pnr_sample_pdf_labels; one launch less per chunk).
struct PdfLabelArgs {
    const float* hit_t; const int32_t* hit_box; const int32_t* hit_count; int max_hits; const int32_t* box_ids;
    int32_t* label_sem; int32_t* label_inst;
};
#define PDF_MAXC 256
#define PDF_MAXT 512
template <int MAXC, int MAXT>
__device__ __forceinline__ void sample_pdf_body(const float* __restrict__ z, const float* __restrict__ weights,
                                                    const float* __restrict__ u, int64_t R, int Nc, int Nf,
                                                    float* __restrict__ zs_out, int32_t* __restrict__ inds_out,
                                                    float* __restrict__ zfine_out, const PdfLabelArgs lab)
{
    // MAXC / MAXT: the LDS footprint sets how many rays a CU has in flight, and this kernel is all latency (dependent LDS searches,
    // a dozen barriers per ray): 9.4 KiB per ray held 17 waves per CU, the <64, 256> instance (64 + 128 samples: every BASELINE
    // config) 3.4 KiB holds the CU's 32
    __shared__ float s_z[MAXC], s_w[MAXC], s_pdf[MAXC], s_cdf[MAXC], s_bins[MAXC];
    __shared__ float s_sort[MAXT], s_out[MAXT];
    __shared__ float s_total, s_part[32];
    const int lane = threadIdx.x;
    const int nb = Nc - 1, nw = Nc - 2, Nt = Nc + Nf;
    int P = 1;
    while (P < Nt) P <<= 1;
    auto labels = [&](int64_t r, const float* sorted) {       // sorted: the ray's Nt merged depths in LDS (written before a barrier)
        if (!lab.label_sem) return;
        const int mh = lab.max_hits;
        const int cnt = lab.hit_count[r] < mh ? lab.hit_count[r] : mh;
        for (int i = lane; i < Nt; i += 64) {
            const int best = label_hit(sorted[i], cnt, [&](int h, float& ti, float& to) {
                ti = lab.hit_t[(r * mh + h) * 2];
                to = lab.hit_t[(r * mh + h) * 2 + 1];
            });
            int ls = -1, li = -1;
            if (best >= 0) {
                const int m = lab.hit_box[r * mh + best];
                ls = lab.box_ids[m * 2];
                li = lab.box_ids[m * 2 + 1];
            }
            lab.label_sem[r * Nt + i] = ls;
            lab.label_inst[r * Nt + i] = li;
        }
    };
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        for (int i = lane; i < Nc; i += 64) {
            s_z[i] = z[r * Nc + i];
            s_w[i] = weights[r * Nc + i];
        }
        __syncthreads();
        for (int k = lane; k < nb; k += 64) s_bins[k] = 0.5f * (s_z[k + 1] + s_z[k]);
        // total = torch.sum(w) in ATen's order (pnro_torch_sum): 8-lane vectors, four partial vectors interleaved -> lane
        // 8 k + l owns accumulator (k, l) and adds its elements in ascending order, exactly as the vector code does
        if (lane < 32) {
            const int k = lane >> 3, l = lane & 7, nv = nw / 8, groups = nv / 4;
            float pacc = 0.0f;
            for (int g = 0; g < groups; ++g) pacc = pacc + (s_w[(g * 4 + k) * 8 + l + 1] + 1e-5f);
            if (k == 0)
                for (int v = groups * 4; v < nv; ++v) pacc = pacc + (s_w[v * 8 + l + 1] + 1e-5f);
            s_part[lane] = pacc;
        }
        __syncthreads();
        if (lane == 0) {
            const int nv = nw / 8;
            float total = 0.0f;
            for (int j = nv * 8; j < nw; ++j) total = total + (s_w[j + 1] + 1e-5f);          // the scalar tail, from 0
            for (int l = 0; l < 8; ++l) {
                float p0 = s_part[l];
                p0 = p0 + s_part[8 + l]; p0 = p0 + s_part[16 + l]; p0 = p0 + s_part[24 + l];
                total = total + p0;
            }
            s_total = total;
        }
        __syncthreads();
        const float total = s_total;
        for (int j = lane; j < nw; j += 64) s_pdf[j] = (s_w[j + 1] + 1e-5f) / total;
        __syncthreads();
        // torch.cumsum: the running sum is kept in DOUBLE, every output rounded to fp32 -- a sequential chain of nw f64 adds
        // (with its LDS traffic, half of this kernel's issue cycles when one lane walks it).  The pdf values are fp32 and
        // non-negative: when every one of them is 0 or >= 2^-28, each is a multiple of 2^-51, and when their sum is < 2 so is
        // every partial sum of ANY subset -- 52 significant bits at most: no f64 add ever rounds, and a parallel prefix scan
        // returns the sequential chain's values bit for bit.  (pdf_j >= 1e-5 / total and sum(pdf) ~ 1: always the case for
        // compositing weights; the sum is checked on the scan's own result, which is exact whenever the true sum is < 2.)
        // Otherwise, and for more than 64 values, the sequential chain runs.
        const float pl = lane < nw ? s_pdf[lane < nw ? lane : 0] : 0.0f;
        double c = (double)pl;
        bool exact = nw <= 64 && __all(pl == 0.0f || pl >= 0x1p-28f);
        if (exact) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double o = __shfl_up(c, d, 64);
                if (lane >= d) c += o;
            }
            exact = __shfl(c, 63, 64) < 1.999;
        }
        if (exact) {
            if (lane == 0) s_cdf[0] = 0.0f;
            if (lane < nw) s_cdf[lane + 1] = (float)c;
        } else if (lane == 0) {
            double c = 0.0;
            s_cdf[0] = 0.0f;
            for (int j = 0; j < nw; ++j) {
                c += (double)s_pdf[j];
                s_cdf[j + 1] = (float)c;
            }
        }
        __syncthreads();
        for (int i = lane; i < Nf; i += 64) {
            const float uu = u ? u[r * Nf + i] : pnr_linspace01(i, Nf);
            int lo = 0, hi = nb;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_cdf[mid] <= uu) lo = mid + 1; else hi = mid;
            }
            const int inds = lo;
            const int below = inds - 1 > 0 ? inds - 1 : 0;
            const int above = inds < nb - 1 ? inds : nb - 1;
            float denom = s_cdf[above] - s_cdf[below];
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (uu - s_cdf[below]) / denom;
            const float span = s_bins[above] - s_bins[below];
            const float m = t * span;
            const float zs = s_bins[below] + m;
            if (zs_out) zs_out[r * Nf + i] = zs;
            if (inds_out) inds_out[r * Nf + i] = inds;
            s_sort[Nc + i] = zs;
        }
        if (zfine_out) {
            __syncthreads();
            // Are the new samples already ascending?  (Always with deterministic u: the inverse CDF is monotone.)  Then the
            // sorted union is a MERGE of two sorted lists: an element's final position is its own index plus the number
            // of elements of the OTHER list that precede it (binary searches; coarse samples go first on ties).  The
            // 36-stage LDS bitonic sort below was 2/3 of this kernel's time; it remains for random u.
            bool asc = true;
            for (int i = lane; i + 1 < Nf; i += 64) asc = asc && (s_sort[Nc + i] <= s_sort[Nc + i + 1]);
            for (int i = lane; i + 1 < Nc; i += 64) asc = asc && (s_z[i] <= s_z[i + 1]);
            if (__all(asc)) {
                for (int i = lane; i < Nc; i += 64) {          // coarse z[i]: + #samples strictly below it
                    const float v = s_z[i];
                    int lo = 0, hi = Nf;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_sort[Nc + mid] < v) lo = mid + 1; else hi = mid; }
                    s_out[i + lo] = v;
                }
                for (int j = lane; j < Nf; j += 64) {          // sample zs[j]: + #coarse z at or below it
                    const float v = s_sort[Nc + j];
                    int lo = 0, hi = Nc;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_z[mid] <= v) lo = mid + 1; else hi = mid; }
                    s_out[j + lo] = v;
                }
                __syncthreads();
                for (int i = lane; i < Nt; i += 64) zfine_out[r * Nt + i] = s_out[i];
                labels(r, s_out);
                __syncthreads();
                continue;
            }
            for (int i = lane; i < Nc; i += 64) s_sort[i] = s_z[i];
            for (int i = Nt + lane; i < P; i += 64) s_sort[i] = INFINITY;
            __syncthreads();
            for (int k = 2; k <= P; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = lane; t < (P >> 1); t += 64) {
                        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                        const int p = i | j;
                        const float a = s_sort[i], b = s_sort[p];
                        const bool asc = (i & k) == 0;
                        if ((a > b) == asc) {
                            s_sort[i] = b;
                            s_sort[p] = a;
                        }
                    }
                    __syncthreads();
                }
            }
            for (int i = lane; i < Nt; i += 64) zfine_out[r * Nt + i] = s_sort[i];
            labels(r, s_sort);
        }
        __syncthreads();
    }
}

// ---- the inference frame's instance (round 6): deterministic u, merged z_fine (+ labels) only, Nc <= 64, Nc + Nf <= 256.
// Same operations on the same values as sample_pdf_body -- bit for bit -- with a different instruction stream: the general body ran
// 8 waves per SIMD with SGPRs spilled to VGPR lanes (249 v_readlane / 147 v_writelane in its code) and VGPRs to scratch, and the
// kernel is bound by ISSUE, not latency.  Here: lane = coarse sample, z / w / pdf / bins in registers (neighbours by DPP); the
// searches are branch-free fixed-depth bisections over LDS (upper / lower bound of a SORTED list = the number of elements <= / <
// the key: identical to the general body's loops whenever the lists are sorted, and the merge path is only taken when they are);
// a ray's hit list is loaded once, one entry per lane, its ids looked up once per hit instead of once per sample, and broadcast by
// v_readlane with the hit loop outermost (the same sequence of comparisons per sample).  Unsorted lists (never with deterministic
// u and a monotone CDF) fall back to the bitonic sort in a function of its own.
__device__ __noinline__ void pdf_bitonic_fallback(float* s_sort, const float* s_z, const float* s_zs, int Nc, int Nf, int P, int lane)
{
    const int Nt = Nc + Nf;
    for (int i = lane; i < Nc; i += 64) s_sort[i] = s_z[i];
    for (int i = lane; i < Nf; i += 64) s_sort[Nc + i] = s_zs[i];
    for (int i = Nt + lane; i < P; i += 64) s_sort[i] = INFINITY;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (P >> 1); t += 64) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const float a = s_sort[i], b = s_sort[p];
                const bool asc = (i & k) == 0;
                if ((a > b) == asc) {
                    s_sort[i] = b;
                    s_sort[p] = a;
                }
            }
            __syncthreads();
        }
    }
}

#ifndef PDF_DET_WPE
#define PDF_DET_WPE 8           /* waves per SIMD the register allocation is held to.  The kernel is bound by LATENCY (a dozen dependent LDS round trips
                                   per ray, one ray per wave): at its natural 73 VGPRs (6 waves) it is SLOWER than the general body at 8 waves
                                   (1.06 against 0.94 ms per frame's 8 launches) although it issues a third of the instructions; held to 64
                                   registers 0.87 (profiles/r06/r06d, same box) */
#endif
#if PDF_DET_WPE
__attribute__((amdgpu_waves_per_eu(PDF_DET_WPE, PDF_DET_WPE)))
#endif
__global__ __launch_bounds__(64) void k_sample_pdf_det(const float* __restrict__ z, const float* __restrict__ weights, int64_t R, int Nc,
                                                        int Nf, float* __restrict__ zfine_out, const PdfLabelArgs lab)
{
    __shared__ float s_z[64], s_w[64], s_pdf[64], s_cdf[64], s_bins[64];
    __shared__ float s_zs[192], s_out[256];
    __shared__ float s_total, s_part[32];
    const int lane = threadIdx.x;
    const int nb = Nc - 1, nw = Nc - 2, Nt = Nc + Nf;
    int topc = 1, topf = 1;                      // the largest powers of two <= Nc, <= Nf: first strides of the bisections
    while (topc * 2 <= Nc) topc *= 2;
    while (topf * 2 <= Nf) topf *= 2;
    const int mh = lab.max_hits;
    for (int64_t r = blockIdx.x; r < R; r += gridDim.x) {
        const float* __restrict__ zr = z + r * Nc;
        const float* __restrict__ wr = weights + r * Nc;
        const float zl = lane < Nc ? zr[lane] : 0.0f, wl = lane < Nc ? wr[lane] : 0.0f;
        // the ray's hit list, one kept entry per lane (requested now, used at the end)
        int cnt = 0, h_ls = -1, h_li = -1;
        float h_ti = 0.0f, h_to = 0.0f;
        if (lab.label_sem) {
            const int c0 = lab.hit_count[r];
            cnt = c0 < mh ? c0 : mh;
            if (lane < cnt) {
                const float2 t2 = *reinterpret_cast<const float2*>(lab.hit_t + (r * mh + lane) * 2);
                h_ti = t2.x; h_to = t2.y;
                const int m = lab.hit_box[r * mh + lane];
                h_ls = lab.box_ids[m * 2];
                h_li = lab.box_ids[m * 2 + 1];
            }
        }
        s_z[lane] = zl;
        s_w[lane] = wl;
        const float zn = __shfl_down(zl, 1, 64), wn = __shfl_down(wl, 1, 64);
        s_bins[lane] = 0.5f * (zn + zl);             // bins[k] = 0.5 (z[k + 1] + z[k]), k < nb
        __syncthreads();
        // total = torch.sum(w[1:-1] + 1e-5) in ATen's order: sample_pdf_body's code
        if (lane < 32) {
            const int k = lane >> 3, l = lane & 7, nv = nw / 8, groups = nv / 4;
            float pacc = 0.0f;
            for (int g = 0; g < groups; ++g) pacc = pacc + (s_w[(g * 4 + k) * 8 + l + 1] + 1e-5f);
            if (k == 0)
                for (int v = groups * 4; v < nv; ++v) pacc = pacc + (s_w[v * 8 + l + 1] + 1e-5f);
            s_part[lane] = pacc;
        }
        __syncthreads();
        if (lane == 0) {
            const int nv = nw / 8;
            float total = 0.0f;
            for (int j = nv * 8; j < nw; ++j) total = total + (s_w[j + 1] + 1e-5f);
            for (int l = 0; l < 8; ++l) {
                float p0 = s_part[l];
                p0 = p0 + s_part[8 + l]; p0 = p0 + s_part[16 + l]; p0 = p0 + s_part[24 + l];
                total = total + p0;
            }
            s_total = total;
        }
        __syncthreads();
        const float total = s_total;
        const float pl = lane < nw ? (wn + 1e-5f) / total : 0.0f;      // pdf[j] = (w[j + 1] + 1e-5) / total
        s_pdf[lane] = pl;
        double c = (double)pl;
        bool exact = __all(pl == 0.0f || pl >= 0x1p-28f);
        if (exact) {
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const double o = __shfl_up(c, d, 64);
                if (lane >= d) c += o;
            }
            exact = __shfl(c, 63, 64) < 1.999;
        }
        if (exact) {
            if (lane == 0) s_cdf[0] = 0.0f;
            if (lane < nw) s_cdf[lane + 1] = (float)c;
        } else {
            __syncthreads();
            if (lane == 0) {
                double cc = 0.0;
                s_cdf[0] = 0.0f;
                for (int j = 0; j < nw; ++j) {
                    cc += (double)s_pdf[j];
                    s_cdf[j + 1] = (float)cc;
                }
            }
        }
        __syncthreads();
        // inverse CDF at the Nf deterministic u: upper bound over cdf[0 .. nb) by bisection (cdf is non-decreasing)
        bool asc = lane + 1 >= Nc || zl <= zn;
        for (int i = lane; i < Nf; i += 64) {
            const float uu = pnr_linspace01(i, Nf);
            int pos = 0;
            for (int st = topc; st >= 1; st >>= 1) {
                const int p = pos + st, q = p <= nb ? p : nb;
                pos = (p <= nb && s_cdf[q - 1] <= uu) ? p : pos;
            }
            const int inds = pos;
            const int below = inds - 1 > 0 ? inds - 1 : 0;
            const int above = inds < nb - 1 ? inds : nb - 1;
            const float cb = s_cdf[below];
            float denom = s_cdf[above] - cb;
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (uu - cb) / denom;
            const float bb = s_bins[below];
            const float span = s_bins[above] - bb;
            const float m = t * span;
            const float zs = bb + m;
            s_zs[i] = zs;
        }
        __syncthreads();
        for (int i = lane; i + 1 < Nf; i += 64) asc = asc && (s_zs[i] <= s_zs[i + 1]);
        if (__all(asc)) {
            // merge by rank: coarse z[i] goes to i + #{samples < z[i]}, sample zs[j] to j + #{coarse z <= zs[j]}
            if (lane < Nc) {
                int pos = 0;
                for (int st = topf; st >= 1; st >>= 1) {
                    const int p = pos + st, q = p <= Nf ? p : Nf;
                    pos = (p <= Nf && s_zs[q - 1] < zl) ? p : pos;
                }
                s_out[lane + pos] = zl;
            }
            for (int j = lane; j < Nf; j += 64) {
                const float v = s_zs[j];
                int pos = 0;
                for (int st = topc; st >= 1; st >>= 1) {
                    const int p = pos + st, q = p <= Nc ? p : Nc;
                    pos = (p <= Nc && s_z[q - 1] <= v) ? p : pos;
                }
                s_out[j + pos] = v;
            }
            __syncthreads();
        } else {
            int P = 1;
            while (P < Nt) P <<= 1;
            pdf_bitonic_fallback(s_out, s_z, s_zs, Nc, Nf, P, lane);
        }
        // the sorted union, and its labels (k_sample_labels' rule: the containing interval with the smallest t_in, lowest index first)
        float* __restrict__ zo = zfine_out + r * Nt;
        for (int i0 = 0; i0 < Nt; i0 += 256) {
            float zz[4];
            int best[4], ls[4], li[4];
            float bt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + lane + 64 * k;
                zz[k] = i < Nt ? s_out[i < Nt ? i : 0] : 0.0f;
                best[k] = -1; ls[k] = -1; li[k] = -1; bt[k] = 0.0f;
                if (i < Nt) zo[i] = zz[k];
            }
            if (lab.label_sem) {
                for (int h = 0; h < cnt; ++h) {
                    const float ti = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h_ti), h));
                    const float to = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(h_to), h));
                    const int hs = __builtin_amdgcn_readlane(h_ls, h), hi2 = __builtin_amdgcn_readlane(h_li, h);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool take = ti <= zz[k] && zz[k] <= to && (best[k] < 0 || ti < bt[k]);
                        best[k] = take ? h : best[k];
                        bt[k] = take ? ti : bt[k];
                        ls[k] = take ? hs : ls[k];
                        li[k] = take ? hi2 : li[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + lane + 64 * k;
                    if (i < Nt) {
                        lab.label_sem[r * Nt + i] = ls[k];
                        lab.label_inst[r * Nt + i] = li[k];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// two instances: <64, 256> (64 + 128 samples: every BASELINE config) at 64 registers and 3.4 KiB of LDS holds 8 waves per SIMD; the
// general one (up to 256 + 256 samples) is bound by its 9.1 KiB of LDS per ray
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8)))
void k_sample_pdf(const float* __restrict__ z, const float* __restrict__ weights, const float* __restrict__ u, int64_t R, int Nc, int Nf,
                  float* __restrict__ zs_out, int32_t* __restrict__ inds_out, float* __restrict__ zfine_out, const PdfLabelArgs lab)
{
    sample_pdf_body<64, 256>(z, weights, u, R, Nc, Nf, zs_out, inds_out, zfine_out, lab);
}
__global__ __launch_bounds__(64)
void k_sample_pdf_big(const float* __restrict__ z, const float* __restrict__ weights, const float* __restrict__ u, int64_t R, int Nc, int Nf,
                      float* __restrict__ zs_out, int32_t* __restrict__ inds_out, float* __restrict__ zfine_out, const PdfLabelArgs lab)
{
    sample_pdf_body<PDF_MAXC, PDF_MAXT>(z, weights, u, R, Nc, Nf, zs_out, inds_out, zfine_out, lab);
}

// ------------------------------------------------------------------------------- a8
// One thread per ray; the box table is wave-uniform (scalar loads, L2/K$ resident: M*60 B).
__global__ __launch_bounds__(256) void k_bbox_hits(const float* __restrict__ rays, int64_t R,
                                                    const float* __restrict__ box, int M, int max_hits,
                                                    float* __restrict__ hit_t, int32_t* __restrict__ hit_box,
                                                    int32_t* __restrict__ hit_count)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R;
         r += (int64_t)gridDim.x * blockDim.x) {
        const float o0 = rays[r * 8 + 0], o1 = rays[r * 8 + 1], o2 = rays[r * 8 + 2];
        const float d0 = rays[r * 8 + 3], d1 = rays[r * 8 + 4], d2 = rays[r * 8 + 5];
        const float nr = rays[r * 8 + 6], fr = rays[r * 8 + 7];
        for (int h = 0; h < max_hits; ++h) {
            hit_box[r * max_hits + h] = -1;
            hit_t[(r * max_hits + h) * 2 + 0] = 0.0f;
            hit_t[(r * max_hits + h) * 2 + 1] = 0.0f;
        }
        int cnt = 0;
        for (int m = 0; m < M; ++m) {
            const float* b = box + m * 15;
            const float p0 = o0 - b[0], p1 = o1 - b[1], p2 = o2 - b[2];
            float tmin = nr, tmax = fr;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float r0 = b[3 + 3 * a], r1 = b[4 + 3 * a], r2 = b[5 + 3 * a];
                const float ol = (r0 * p0 + r1 * p1) + r2 * p2;
                const float dl = (r0 * d0 + r1 * d1) + r2 * d2;
                const float inv = 1.0f / dl;
                const float e = b[12 + a];
                const float t1 = (-e - ol) * inv, t2 = (e - ol) * inv;
                tmin = fmaxf(tmin, fminf(t1, t2));
                tmax = fminf(tmax, fmaxf(t1, t2));
            }
            if (tmin <= tmax) {
                // keep the max_hits NEAREST intervals, ascending (t_in, box index): a street-scene ray crosses more boxes
                // than max_hits, and the near ones are the ones its samples fall in.  Insertion from the back; the
                // farthest entry falls off the end.  (Same op order as pnro_bbox_hits: bit-exact.)
                const int n = cnt < max_hits ? cnt : max_hits;
                int pos = n;
                while (pos > 0 && hit_t[(r * max_hits + pos - 1) * 2] > tmin) --pos;
                if (pos < max_hits) {
                    for (int k = (n < max_hits ? n : max_hits - 1); k > pos; --k) {
                        hit_t[(r * max_hits + k) * 2 + 0] = hit_t[(r * max_hits + k - 1) * 2 + 0];
                        hit_t[(r * max_hits + k) * 2 + 1] = hit_t[(r * max_hits + k - 1) * 2 + 1];
                        hit_box[r * max_hits + k] = hit_box[r * max_hits + k - 1];
                    }
                    hit_t[(r * max_hits + pos) * 2 + 0] = tmin;
                    hit_t[(r * max_hits + pos) * 2 + 1] = tmax;
                    hit_box[r * max_hits + pos] = m;
                }
                ++cnt;                      // TRUE number of intersected boxes: > max_hits reports the overflow
            }
        }
        hit_count[r] = cnt;
    }
}

// a8 + a3 + a8 in one launch (the coarse level's per-ray preamble: k_bbox_hits, k_restrict_rays' hull, k_stratified and
// k_sample_labels were four launches and three passes over the hit lists).  Phase 1, one thread per ray: the slab tests with the
// ray's kept intervals in LDS ([entry][thread]: conflict-free; k_bbox_hits keeps them in global memory) -- same operations in the
// same order, so the hit lists are k_bbox_hits' bit for bit.  Phase 2, lane = sample: z and the labels, written in whole rows.
// 64 rays per 4-wave workgroup: phase 1 occupies one wave (one thread per ray is all the parallelism it has; as one wave per SIMD
// of a 256-ray workgroup, phase 2 ran at 42 us against the separate kernels' 26), the other three are in the phase 2 of the CU's
// other workgroups.  65,536 rays x 64 samples, 64 boxes: 50 us against 27 + 11 + 15 for the three kernels (profiles/r05/r05l).  max_hits <= 8 (7 KiB of LDS); the separate kernels remain for larger lists.
#define SETUP_MAXH 8
struct RaySetupArgs {
    const float* rays; int64_t R; const float* box; int M; int max_hits; const int32_t* box_ids;
    int N, lindisp, hull; const float* t_rand;
    float* hit_t; int32_t* hit_box; int32_t* hit_count; float* z; int32_t* label_sem; int32_t* label_inst;
};
__global__ __launch_bounds__(256) void k_ray_setup(const RaySetupArgs a)
{
    __shared__ float2 s_t[SETUP_MAXH][64];      // (t_in, t_out)
    __shared__ float s_nr[64], s_fr[64];
    __shared__ int s_hb[SETUP_MAXH][64], s_cnt[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int mh = a.max_hits, N = a.N;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < a.R; base += (int64_t)gridDim.x * 64) {
        const int64_t r = base + tid;
        if (tid < 64 && r < a.R) {              // phase 1: the block's first wave (the others are in another block's phase 2)
            const float4 ra = *reinterpret_cast<const float4*>(a.rays + r * 8), rb = *reinterpret_cast<const float4*>(a.rays + r * 8 + 4);
            const float o0 = ra.x, o1 = ra.y, o2 = ra.z, d0 = ra.w, d1 = rb.x, d2 = rb.y;
            float nr = rb.z, fr = rb.w;
            for (int h = 0; h < SETUP_MAXH; ++h) { s_hb[h][tid] = -1; s_t[h][tid] = make_float2(0.0f, 0.0f); }
            int cnt = 0;
            for (int m = 0; m < a.M; ++m) {
                const float* b = a.box + m * 15;
                const float p0 = o0 - b[0], p1 = o1 - b[1], p2 = o2 - b[2];
                float tmin = nr, tmax = fr;
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const float r0 = b[3 + 3 * x], r1 = b[4 + 3 * x], r2 = b[5 + 3 * x];
                    const float ol = (r0 * p0 + r1 * p1) + r2 * p2;
                    const float dl = (r0 * d0 + r1 * d1) + r2 * d2;
                    const float inv = 1.0f / dl;
                    const float e = b[12 + x];
                    const float t1 = (-e - ol) * inv, t2 = (e - ol) * inv;
                    tmin = fmaxf(tmin, fminf(t1, t2));
                    tmax = fminf(tmax, fmaxf(t1, t2));
                }
                if (tmin <= tmax) {             // k_bbox_hits' insertion: the max_hits nearest, ascending (t_in, box index)
                    const int n = cnt < mh ? cnt : mh;
                    int pos = n;
                    while (pos > 0 && s_t[pos - 1][tid].x > tmin) --pos;
                    if (pos < mh) {
                        for (int k = (n < mh ? n : mh - 1); k > pos; --k) {
                            s_t[k][tid] = s_t[k - 1][tid];
                            s_hb[k][tid] = s_hb[k - 1][tid];
                        }
                        s_t[pos][tid] = make_float2(tmin, tmax);
                        s_hb[pos][tid] = m;
                    }
                    ++cnt;
                }
            }
            a.hit_count[r] = cnt;
            const int kept = cnt < mh ? cnt : mh;
            for (int h = 0; h < mh; ++h) {
                *reinterpret_cast<float2*>(a.hit_t + (r * mh + h) * 2) = s_t[h][tid];
                a.hit_box[r * mh + h] = s_hb[h][tid];
            }
            if (a.hull && kept > 0) {           // k_restrict_rays: the hull of the kept intervals replaces [near, far]
                float lo = s_t[0][tid].x, hi = s_t[0][tid].y;
                for (int h = 1; h < kept; ++h) {
                    lo = fminf(lo, s_t[h][tid].x);
                    hi = fmaxf(hi, s_t[h][tid].y);
                }
                nr = lo; fr = hi;
            }
            s_nr[tid] = nr; s_fr[tid] = fr; s_cnt[tid] = kept;
        }
        __syncthreads();
        // Phase 2: each of the four waves takes 16 of the 64 rays, one ray in hand at a time.  Everything a ray needs from LDS is
        // requested for RU rays at once (wave-uniform addresses: broadcast reads), so the chains overlap; the hit loop runs over
        // registers with the ray's own (wave-uniform) count.
        constexpr int RU = 2;
        for (int rr = wv * 16; rr < wv * 16 + 16; rr += RU) {
            if (base + rr >= a.R) break;                            // wave-uniform
            float nr[RU], fr[RU];
            int cnt[RU], hb[RU][SETUP_MAXH];
            float2 iv[RU][SETUP_MAXH];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int t = rr + u;
                // wave-uniform values: scalar registers (as vector registers these 26 values per ray set the occupancy)
                auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
                nr[u] = uni(s_nr[t]); fr[u] = uni(s_fr[t]);
                cnt[u] = __builtin_amdgcn_readfirstlane(s_cnt[t]);
#pragma unroll
                for (int h = 0; h < SETUP_MAXH; ++h) {
                    const float2 v = s_t[h][t];
                    iv[u][h] = make_float2(uni(v.x), uni(v.y));
                    hb[u][h] = __builtin_amdgcn_readfirstlane(s_hb[h][t]);
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t ray = base + rr + u;
                if (ray >= a.R) break;                              // wave-uniform
                for (int i = lane; i < N; i += 64) {
                    const int64_t s = ray * N + i;
                    const float zz = strat_sample(nr[u], fr[u], i, N, a.lindisp, a.t_rand ? a.t_rand + s : nullptr);
                    a.z[s] = zz;
                    if (a.label_sem) {
                        int best = -1, m = -1;
                        float bt = 0.0f;
#pragma unroll
                        for (int h = 0; h < SETUP_MAXH; ++h) {      // label_hit, unrolled over registers
                            if (h >= cnt[u]) break;
                            const float ti = iv[u][h].x, to = iv[u][h].y;
                            if (ti <= zz && zz <= to && (best < 0 || ti < bt)) { best = h; bt = ti; m = hb[u][h]; }
                        }
                        int ls = -1, li = -1;
                        if (best >= 0) {
                            ls = a.box_ids[m * 2];
                            li = a.box_ids[m * 2 + 1];
                        }
                        a.label_sem[s] = ls;
                        a.label_inst[s] = li;
                    }
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_sample_labels(const float* __restrict__ z, int64_t R, int N,
                                                        const float* __restrict__ hit_t,
                                                        const int32_t* __restrict__ hit_box,
                                                        const int32_t* __restrict__ hit_count, int max_hits,
                                                        const int32_t* __restrict__ box_ids,
                                                        int32_t* __restrict__ label_sem,
                                                        int32_t* __restrict__ label_inst)
{
    const int64_t total = R * N;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = s / N;
        const float zz = z[s];
        const int cnt = hit_count[r] < max_hits ? hit_count[r] : max_hits;     // hit_count is the true count (overflow)
        const int best = label_hit(zz, cnt, [&](int h, float& ti, float& to) {
            ti = hit_t[(r * max_hits + h) * 2];
            to = hit_t[(r * max_hits + h) * 2 + 1];
        });
        int ls = -1, li = -1;
        if (best >= 0) {
            const int m = hit_box[r * max_hits + best];
            ls = box_ids[m * 2];
            li = box_ids[m * 2 + 1];
        }
        label_sem[s] = ls;
        label_inst[s] = li;
    }
}

// ------------------------------------------------------------------------------- C-ABI
PNR_EXPORT int pnr_stratified(const float* rays, int64_t n_rays, int n_samples, int lindisp,
                              const float* t_rand, float* z_out, void* stream)
{
    PNR_REQUIRE(n_rays >= 0 && n_samples >= 1, "pnr_stratified: bad size R=%lld N=%d", (long long)n_rays, n_samples);
    if (n_rays == 0) return PNR_OK;     // empty input is a no-op
    PNR_REQUIRE(rays && z_out, "pnr_stratified: null pointer");
    const int64_t total = n_rays * n_samples;
    hipLaunchKernelGGL(k_stratified, dim3(pnr_grid_cap((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rays, n_rays, n_samples, lindisp, t_rand, z_out);
    PNR_CHECK_LAUNCH("pnr_stratified");
    return PNR_OK;
}

PNR_EXPORT int pnr_gen_rays(const float* intr4_host, const float* c2w12_host, int width, int height, float near_, float far_,
                            const int32_t* pix, int64_t n_rays, float* rays, void* stream)
{
    PNR_REQUIRE(intr4_host && c2w12_host, "pnr_gen_rays: null camera");
    PNR_REQUIRE(width >= 1 && height >= 1 && n_rays >= 0, "pnr_gen_rays: bad size");
    if (n_rays == 0) return PNR_OK;             // before the pointer checks: an empty pixel list has a null pointer
    PNR_REQUIRE(pix || n_rays == (int64_t)width * height, "pnr_gen_rays: without pixel indices n_rays must be width*height");
    PNR_REQUIRE(intr4_host[0] != 0.0f && intr4_host[1] != 0.0f, "pnr_gen_rays: zero focal length");
    PNR_REQUIRE(rays && (((uintptr_t)rays) & 15) == 0, "pnr_gen_rays: rays must be a 16-byte aligned device buffer");
    GenRaysArgs a;
    a.fx = intr4_host[0]; a.fy = intr4_host[1]; a.cx = intr4_host[2]; a.cy = intr4_host[3];
    for (int k = 0; k < 12; ++k) a.c2w[k] = c2w12_host[k];
    a.width = width; a.near_ = near_; a.far_ = far_; a.pix = pix; a.R = n_rays; a.rays = rays;
    hipLaunchKernelGGL(k_gen_rays, dim3(pnr_grid_cap((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    PNR_CHECK_LAUNCH("pnr_gen_rays");
    return PNR_OK;
}

PNR_EXPORT int pnr_points(const float* rays, const float* z, int64_t n_rays, int n_samples, float* pts_out,
                          void* stream)
{
    PNR_REQUIRE(n_rays >= 0 && n_samples >= 1, "pnr_points: bad size");
    if (n_rays == 0) return PNR_OK;
    PNR_REQUIRE(rays && z && pts_out, "pnr_points: null pointer");
    const int64_t total = n_rays * n_samples * 3;
    hipLaunchKernelGGL(k_points, dim3(pnr_grid_cap((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rays, z, n_rays, n_samples, pts_out);
    PNR_CHECK_LAUNCH("pnr_points");
    return PNR_OK;
}

PNR_EXPORT int pnr_embed(const float* x, int64_t n, int L, float* out, void* stream)
{
    PNR_REQUIRE(n >= 0 && L >= 0 && L <= 16, "pnr_embed: bad size n=%lld L=%d", (long long)n, L);
    if (n == 0) return PNR_OK;
    PNR_REQUIRE(x && out, "pnr_embed: null pointer");
    const int64_t total = n * (3 + 6 * L);
    hipLaunchKernelGGL(k_embed, dim3(pnr_grid_cap((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, n, L, out);
    PNR_CHECK_LAUNCH("pnr_embed");
    return PNR_OK;
}

static void launch_sample_pdf(const float* z, const float* weights, const float* u, int64_t n_rays, int n_coarse, int n_fine,
                              float* z_samples, int32_t* inds, float* z_fine, const PdfLabelArgs& lab, hipStream_t st)
{
    int P = 1;
    while (P < n_coarse + n_fine) P <<= 1;              // the bitonic path pads the union to a power of two
    const bool general_only = getenv("PNR_SAMPLE_PDF_GENERAL") != nullptr;      // A/B switch (tests, tools): the general body everywhere
    if (!u && !z_samples && !inds && z_fine && n_coarse <= 64 && n_fine <= 192 && P <= 256 && lab.max_hits <= 64 && !general_only)
        hipLaunchKernelGGL(k_sample_pdf_det, dim3(pnr_grid_cap(n_rays, 32)), dim3(64), 0, st, z, weights, n_rays, n_coarse, n_fine, z_fine, lab);
    else if (n_coarse <= 64 && P <= 256)
        hipLaunchKernelGGL(k_sample_pdf, dim3(pnr_grid_cap(n_rays, 32)), dim3(64), 0, st, z, weights, u, n_rays, n_coarse, n_fine,
                           z_samples, inds, z_fine, lab);
    else
        hipLaunchKernelGGL(k_sample_pdf_big, dim3(pnr_grid_cap(n_rays, 32)), dim3(64), 0, st, z, weights, u, n_rays, n_coarse,
                           n_fine, z_samples, inds, z_fine, lab);
}

PNR_EXPORT int pnr_sample_pdf(const float* z, const float* weights, const float* u, int64_t n_rays, int n_coarse,
                              int n_fine, float* z_samples, int32_t* inds, float* z_fine, void* stream)
{
    PNR_REQUIRE(n_rays <= 0 || (z && weights), "pnr_sample_pdf: null pointer");
    PNR_REQUIRE(n_coarse >= 3 && n_coarse <= PDF_MAXC, "pnr_sample_pdf: n_coarse=%d outside [3,%d]", n_coarse, PDF_MAXC);
    PNR_REQUIRE(n_fine >= 1 && n_coarse + n_fine <= PDF_MAXT, "pnr_sample_pdf: n_coarse+n_fine=%d > %d",
                n_coarse + n_fine, PDF_MAXT);
    if (n_rays <= 0) return PNR_OK;
    PdfLabelArgs lab;
    memset(&lab, 0, sizeof(lab));
    launch_sample_pdf(z, weights, u, n_rays, n_coarse, n_fine, z_samples, inds, z_fine, lab, (hipStream_t)stream);
    PNR_CHECK_LAUNCH("pnr_sample_pdf");
    return PNR_OK;
}

// a7 + a8: z_fine as pnr_sample_pdf, and the labels pnr_sample_labels would give for it, in the same launch
PNR_EXPORT int pnr_sample_pdf_labels(const float* z, const float* weights, const float* u, int64_t n_rays, int n_coarse, int n_fine,
                                     float* z_fine, const float* hit_t, const int32_t* hit_box, const int32_t* hit_count,
                                     int max_hits, const int32_t* box_ids, int32_t* label_sem, int32_t* label_inst, void* stream)
{
    PNR_REQUIRE(n_rays <= 0 || (z && weights && z_fine), "pnr_sample_pdf_labels: null pointer");
    PNR_REQUIRE(n_rays <= 0 || (hit_t && hit_box && hit_count && box_ids && label_sem && label_inst), "pnr_sample_pdf_labels: null pointer");
    PNR_REQUIRE(n_coarse >= 3 && n_coarse <= PDF_MAXC, "pnr_sample_pdf_labels: n_coarse=%d outside [3,%d]", n_coarse, PDF_MAXC);
    PNR_REQUIRE(n_fine >= 1 && n_coarse + n_fine <= PDF_MAXT, "pnr_sample_pdf_labels: n_coarse+n_fine=%d > %d", n_coarse + n_fine, PDF_MAXT);
    PNR_REQUIRE(max_hits >= 1, "pnr_sample_pdf_labels: bad max_hits");
    if (n_rays <= 0) return PNR_OK;
    PdfLabelArgs lab;
    lab.hit_t = hit_t; lab.hit_box = hit_box; lab.hit_count = hit_count; lab.max_hits = max_hits; lab.box_ids = box_ids;
    lab.label_sem = label_sem; lab.label_inst = label_inst;
    launch_sample_pdf(z, weights, u, n_rays, n_coarse, n_fine, nullptr, nullptr, z_fine, lab, (hipStream_t)stream);
    PNR_CHECK_LAUNCH("pnr_sample_pdf_labels");
    return PNR_OK;
}

// a8 + a3 (+ a8): hit lists as pnr_bbox_hits, z as pnr_stratified (over the hull of the kept intervals with hull != 0:
// pnr_restrict_rays), labels as pnr_sample_labels (label_sem / label_inst may both be null), in one launch.  max_hits <= 8.
PNR_EXPORT int pnr_ray_setup(const float* rays, int64_t n_rays, const float* box, int n_box, int max_hits, const int32_t* box_ids,
                             int n_samples, int lindisp, const float* t_rand, int hull, float* hit_t, int32_t* hit_box,
                             int32_t* hit_count, float* z_out, int32_t* label_sem, int32_t* label_inst, void* stream)
{
    PNR_REQUIRE(n_rays >= 0 && n_samples >= 1, "pnr_ray_setup: bad size R=%lld N=%d", (long long)n_rays, n_samples);
    PNR_REQUIRE(max_hits >= 1 && max_hits <= SETUP_MAXH, "pnr_ray_setup: max_hits=%d outside [1,%d] (use the separate entry points)",
                max_hits, SETUP_MAXH);
    PNR_REQUIRE(n_box >= 0 && (n_box == 0 || box), "pnr_ray_setup: bad box table");
    if (n_rays == 0) return PNR_OK;
    PNR_REQUIRE(rays && hit_t && hit_box && hit_count && z_out, "pnr_ray_setup: null pointer");
    PNR_REQUIRE((label_sem != nullptr) == (label_inst != nullptr) && (!label_sem || box_ids || n_box == 0),
                "pnr_ray_setup: labels need both outputs and box_ids");
    PNR_REQUIRE((((uintptr_t)rays) & 15) == 0 && (((uintptr_t)hit_t) & 7) == 0, "pnr_ray_setup: rays must be 16-byte aligned");
    RaySetupArgs a;
    a.rays = rays; a.R = n_rays; a.box = box; a.M = n_box; a.max_hits = max_hits; a.box_ids = box_ids;
    a.N = n_samples; a.lindisp = lindisp; a.hull = hull; a.t_rand = t_rand;
    a.hit_t = hit_t; a.hit_box = hit_box; a.hit_count = hit_count; a.z = z_out; a.label_sem = label_sem; a.label_inst = label_inst;
    hipLaunchKernelGGL(k_ray_setup, dim3(pnr_grid_cap((n_rays + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a);
    PNR_CHECK_LAUNCH("pnr_ray_setup");
    return PNR_OK;
}

PNR_EXPORT int pnr_bbox_hits(const float* rays, int64_t n_rays, const float* box, int n_box, int max_hits,
                             float* hit_t, int32_t* hit_box, int32_t* hit_count, void* stream)
{
    PNR_REQUIRE(n_rays <= 0 || (rays && hit_t && hit_box && hit_count), "pnr_bbox_hits: null pointer");
    PNR_REQUIRE(n_box >= 0 && (n_box == 0 || box), "pnr_bbox_hits: bad box table");
    PNR_REQUIRE(max_hits >= 1 && max_hits <= 64, "pnr_bbox_hits: max_hits=%d outside [1,64]", max_hits);
    if (n_rays <= 0) return PNR_OK;
    hipLaunchKernelGGL(k_bbox_hits, dim3(pnr_grid_cap((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       rays, n_rays, box, n_box, max_hits, hit_t, hit_box, hit_count);
    PNR_CHECK_LAUNCH("pnr_bbox_hits");
    return PNR_OK;
}

PNR_EXPORT int pnr_sample_labels(const float* z, int64_t n_rays, int n_samples, const float* hit_t,
                                 const int32_t* hit_box, const int32_t* hit_count, int max_hits,
                                 const int32_t* box_ids, int32_t* label_sem, int32_t* label_inst, void* stream)
{
    PNR_REQUIRE(n_rays <= 0 || (z && hit_t && hit_box && hit_count && box_ids && label_sem && label_inst),
                "pnr_sample_labels: null pointer");
    PNR_REQUIRE(n_samples >= 1 && max_hits >= 1, "pnr_sample_labels: bad size");
    if (n_rays <= 0) return PNR_OK;
    const int64_t total = n_rays * n_samples;
    hipLaunchKernelGGL(k_sample_labels, dim3(pnr_grid_cap((total + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, z, n_rays, n_samples, hit_t, hit_box, hit_count, max_hits, box_ids,
                       label_sem, label_inst);
    PNR_CHECK_LAUNCH("pnr_sample_labels");
    return PNR_OK;
}

// ---- a8 / a3 switch: sampling restricted to the bbox prior ("is z uniform in [near, far] or restricted to
// bbox hit intervals?" -- the reference's answer is not in the mount; this is the restricted form as a CONFIG switch,
// cfg.bbox_sampling = "hull").  A ray that hits boxes is sampled over the hull of its kept intervals, [min t_in, max t_out]
// (the slab test starts from [near, far], so the hull already lies inside it); a ray without a hit keeps [near, far].  Only
// the near / far columns of the ray record change: everything downstream (stratified, sample_pdf, labels) is unchanged.
__global__ __launch_bounds__(256) void k_restrict_rays(const float* __restrict__ rays, int64_t R, const float* __restrict__ hit_t,
                                                        const int32_t* __restrict__ hit_count, int max_hits, float* __restrict__ out)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = *reinterpret_cast<const float4*>(rays + r * 8);
        float4 b = *reinterpret_cast<const float4*>(rays + r * 8 + 4);
        const int cnt = hit_count[r] < max_hits ? hit_count[r] : max_hits;
        if (cnt > 0) {
            float lo = hit_t[(r * max_hits) * 2], hi = hit_t[(r * max_hits) * 2 + 1];
            for (int h = 1; h < cnt; ++h) {
                lo = fminf(lo, hit_t[(r * max_hits + h) * 2]);
                hi = fmaxf(hi, hit_t[(r * max_hits + h) * 2 + 1]);
            }
            b.z = lo; b.w = hi;
        }
        *reinterpret_cast<float4*>(out + r * 8) = a;
        *reinterpret_cast<float4*>(out + r * 8 + 4) = b;
    }
}

PNR_EXPORT int pnr_restrict_rays(const float* rays, int64_t n_rays, const float* hit_t, const int32_t* hit_count, int max_hits,
                                 float* rays_out, void* stream)
{
    PNR_REQUIRE(n_rays >= 0 && max_hits >= 1, "pnr_restrict_rays: bad size");
    if (n_rays == 0) return PNR_OK;
    PNR_REQUIRE(rays && hit_t && hit_count && rays_out, "pnr_restrict_rays: null pointer");
    PNR_REQUIRE(((((uintptr_t)rays) | ((uintptr_t)rays_out)) & 15) == 0, "pnr_restrict_rays: ray records must be 16-byte aligned");
    hipLaunchKernelGGL(k_restrict_rays, dim3(pnr_grid_cap((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rays, n_rays,
                       hit_t, hit_count, max_hits, rays_out);
    PNR_CHECK_LAUNCH("pnr_restrict_rays");
    return PNR_OK;
}
