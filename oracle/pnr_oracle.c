/*
 * pnr_oracle.c -- CPU restatement (plain C, strict fp32 op order) of the PanopticNeRF
 * render_rays hot path.  TEST INFRASTRUCTURE ONLY: nothing in the product path
 * (panopticnerf_amd/) may import, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED.  The mounted reference (/root/reference) holds only README.md
 * (README.md:7 and README.md:13 point at the un-mounted code branches), so no function
 * below can cite a reference file:line.  Each follows SURVEY.md section 8a (rows a3..a8),
 * i.e. the canonical NeRF formulation that the function names in BASELINE.json's
 * north_star (render_rays, sample_pdf, raw2outputs, Embedder) denote.
 *
 * Why C and a fixed op order: BASELINE.json asks for bit-exact sample indices and
 * ray-bbox hits.  searchsorted flips at bin edges when the CDF differs by one ulp, so the
 * order of every fp32 operation has to be pinned.  Since round 3 the order pinned here is
 * TORCH'S OWN, as its CPU kernels (torch 2.10, this container, dispatched at
 * torch.backends.cpu.get_cpu_capability() == "AVX512": ATen vectorises per CPU capability, so
 * "as written" is a statement about THAT build -- tests/test_oracle.py asserts the capability)
 * evaluate the reference's expressions -- restated from measurements against torch, bit for bit
 * (tests/test_oracle.py::test_c_oracle_is_torch_as_written):
 *   torch.linspace(0, 1, N)   step = 1/(N-1);  t_i = step*i for i < N/2, else 1 - step*(N-1-i) with ONE rounding (fma)
 *   torch.sum(x, -1)          8-lane vector partial sums, 4 of them interleaved (pnro_torch_sum below)
 *   torch.cumsum(x, -1)       running sum in DOUBLE, every output rounded to fp32
 * everything else is elementwise: one rounding per operation, no FMA contraction (build with
 * -ffp-contract=off).  The HIP kernels reproduce this file instruction for instruction; the
 * vectorised torch restatement (oracle/torch_oracle.py) IS torch as written and must agree with
 * this file exactly on z, sample indices, z_samples and z_fine.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PNRO_API __attribute__((visibility("default")))

/* Ray generation (the dataset-side producer of batch['rays']).  Pinhole camera:
 * intr = {fx, fy, cx, cy};  c2w = 3x4 row-major camera-to-world [R | t], camera axes x right, y down, z forward;
 * pixel (i = column, j = row):  x = (i - cx)/fx,  y = (j - cy)/fy,  d_k = (R_k0*x + R_k1*y) + R_k2,  o = t.
 * pix: n_rays linear pixel indices j*width + i, or NULL = the whole frame in row-major order.
 * rays: (n_rays, 8) = o(3) d(3) near far.  d is NOT normalised (raw2outputs multiplies by |d|). */
PNRO_API void pnro_gen_rays(const float* intr, const float* c2w, int width, int height, float near_, float far_,
                            const int32_t* pix, int64_t n_rays, float* rays)
{
    (void)height;
    for (int64_t r = 0; r < n_rays; ++r) {
        const int64_t p = pix ? (int64_t)pix[r] : r;
        const int j = (int)(p / width), i = (int)(p - (int64_t)j * width);
        const float x = ((float)i - intr[2]) / intr[0];
        const float y = ((float)j - intr[3]) / intr[1];
        for (int k = 0; k < 3; ++k) {
            const float a = c2w[k * 4 + 0] * x, b = c2w[k * 4 + 1] * y;
            rays[r * 8 + 3 + k] = (a + b) + c2w[k * 4 + 2];
            rays[r * 8 + k] = c2w[k * 4 + 3];
        }
        rays[r * 8 + 6] = near_;
        rays[r * 8 + 7] = far_;
    }
}

/* torch.linspace(0, 1, steps = N)[i], fp32, as ATen's CPU kernel computes it (RangeFactories: two-sided, the upper half from
 * the end point; the product is not rounded separately there).  Measured bit-exact against torch for N = 7 .. 192. */
static inline float pnro_linspace01(int i, int N)
{
    if (N <= 1) return 0.0f;
    const float step = 1.0f / (float)(N - 1);
    if (i < N / 2) return step * (float)i;
    return fmaf(-step, (float)(N - 1 - i), 1.0f);
}

/* torch.sum over a contiguous fp32 row of n >= 8 elements, as ATen's CPU reduction orders it: the row is cut into vectors of
 * 8 lanes; vectors 0..3 (mod 4) accumulate into four partial vectors, leftover vectors into partial 0; partials 1..3 are
 * added to partial 0; the scalar tail is summed sequentially from 0; finally the 8 lanes of partial 0 are added to that in
 * lane order.  Measured bit-exact against torch.sum for n = 30 .. 200 (tests); rows shorter than one vector: sequential. */
static float pnro_torch_sum(const float* x, int n)
{
    enum { V = 8, ILP = 4 };
    const int nv = n / V;
    float part[ILP][V];
    for (int k = 0; k < ILP; ++k) for (int l = 0; l < V; ++l) part[k][l] = 0.0f;
    const int groups = nv / ILP;
    for (int g = 0; g < groups; ++g)
        for (int k = 0; k < ILP; ++k)
            for (int l = 0; l < V; ++l) part[k][l] = part[k][l] + x[(g * ILP + k) * V + l];
    for (int v = groups * ILP; v < nv; ++v)
        for (int l = 0; l < V; ++l) part[0][l] = part[0][l] + x[v * V + l];
    for (int k = 1; k < ILP; ++k)
        for (int l = 0; l < V; ++l) part[0][l] = part[0][l] + part[k][l];
    float acc = 0.0f;
    for (int i = nv * V; i < n; ++i) acc = acc + x[i];
    for (int l = 0; l < V; ++l) acc = acc + part[0][l];
    return acc;
}
PNRO_API float pnro_torch_sum_row(const float* x, int n) { return pnro_torch_sum(x, n); }
PNRO_API float pnro_linspace01_at(int i, int N) { return pnro_linspace01(i, N); }

/* ---------------------------------------------------------------- a3: stratified sampler
 * SURVEY 8a row a3.  t = torch.linspace(0, 1, N) (pnro_linspace01);
 * z = near*(1-t) + far*t, or 1/(1/near*(1-t) + 1/far*t) when lindisp.
 * With t_rand (perturb>0): mids = .5*(z[1:]+z[:-1]); upper = cat(mids, z[-1]);
 * lower = cat(z[0], mids); z = lower + (upper-lower)*t_rand.
 * rays: (R,8) = o(3) d(3) near far.   z_out: (R,N). */
PNRO_API void pnro_stratified(const float* rays, int64_t R, int N, int lindisp,
                              const float* t_rand, float* z_out)
{
    float* zb = (float*)malloc(sizeof(float) * (size_t)N);
    for (int64_t r = 0; r < R; ++r) {
        const float nr = rays[r * 8 + 6], fr = rays[r * 8 + 7];
        for (int i = 0; i < N; ++i) {
            const float t = pnro_linspace01(i, N);
            const float omt = 1.0f - t;
            float z;
            if (!lindisp) {
                const float a = nr * omt, b = fr * t;
                z = a + b;
            } else {
                const float a = (1.0f / nr) * omt, b = (1.0f / fr) * t;
                z = 1.0f / (a + b);
            }
            zb[i] = z;
        }
        if (t_rand) {
            for (int i = 0; i < N; ++i) {
                const float lo = (i == 0) ? zb[0] : 0.5f * (zb[i] + zb[i - 1]);
                const float up = (i == N - 1) ? zb[N - 1] : 0.5f * (zb[i + 1] + zb[i]);
                const float w = up - lo;
                const float m = w * t_rand[r * N + i];
                z_out[r * N + i] = lo + m;
            }
        } else {
            for (int i = 0; i < N; ++i) z_out[r * N + i] = zb[i];
        }
    }
    free(zb);
}

/* pts = o + d*z (mul then add, no FMA).  pts_out: (R,N,3). */
PNRO_API void pnro_points(const float* rays, const float* z, int64_t R, int N, float* pts_out)
{
    for (int64_t r = 0; r < R; ++r)
        for (int i = 0; i < N; ++i)
            for (int a = 0; a < 3; ++a) {
                const float m = rays[r * 8 + 3 + a] * z[r * N + i];
                pts_out[(r * N + i) * 3 + a] = rays[r * 8 + a] + m;
            }
}

/* ---------------------------------------------------------------- a4: Embedder
 * SURVEY 8a row a4.  gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x),
 * cos(2^(L-1) x)], include_input, log-sampled bands 2^k (exact in fp32), blocks of 3.
 * x: (n,3) -> out: (n, 3+6L). */
PNRO_API void pnro_embed(const float* x, int64_t n, int L, float* out)
{
    const int E = 3 + 6 * L;
    for (int64_t s = 0; s < n; ++s) {
        float* o = out + s * E;
        for (int a = 0; a < 3; ++a) o[a] = x[s * 3 + a];
        for (int k = 0; k < L; ++k) {
            const float f = ldexpf(1.0f, k);
            for (int a = 0; a < 3; ++a) {
                const float arg = x[s * 3 + a] * f;
                o[3 + 6 * k + a] = sinf(arg);
                o[3 + 6 * k + 3 + a] = cosf(arg);
            }
        }
    }
}

/* ---------------------------------------------------------------- a6: raw2outputs
 * SURVEY 8a row a6.  raw element (sample s, channel c) lives at raw[s*stride_s + c*stride_c]
 * with channels [r g b sigma | C semantic logits | K instance logits].
 *   dists_i = z_{i+1}-z_i, last = 1e10;  dists *= ||d||
 *   alpha_i = 1 - exp(-relu(sigma_i + noise_i) * dists_i)
 *   T_i = prod_{j<i} (1 - alpha_j + 1e-10)   (sequential product)
 *   w_i = alpha_i * T_i
 *   rgb = sum w_i sigmoid(raw_rgb_i); depth = sum w_i z_i; acc = sum w_i
 *   sem = sum w_i s_i with s_i = logits (sem_mode 0) or softmax(logits) (sem_mode 1); same for inst
 *   fixed fields: fix_sem[c] = sum_i w_i [label_sem_i == c]; fix_inst likewise
 *   white_bkgd: rgb += 1 - acc
 * Outputs may be NULL when not wanted. */
PNRO_API void pnro_composite(const float* raw, int64_t stride_s, int64_t stride_c,
                             const float* z, const float* rays, const float* noise,
                             const int32_t* label_sem, const int32_t* label_inst,
                             int64_t R, int N, int C, int K, int sem_mode, int white_bkgd,
                             float* rgb, float* depth, float* acc, float* weights,
                             float* sem, float* inst, float* fix_sem, float* fix_inst)
{
    float* tmp = (float*)malloc(sizeof(float) * (size_t)((C > K ? C : K) + 1));
    for (int64_t r = 0; r < R; ++r) {
        const float dx = rays[r * 8 + 3], dy = rays[r * 8 + 4], dz = rays[r * 8 + 5];
        const float dn = sqrtf((dx * dx + dy * dy) + dz * dz);
        float T = 1.0f;
        float o_rgb[3] = {0, 0, 0}, o_d = 0, o_a = 0;
        if (sem) for (int c = 0; c < C; ++c) sem[r * C + c] = 0;
        if (inst) for (int c = 0; c < K; ++c) inst[r * K + c] = 0;
        if (fix_sem) for (int c = 0; c < C; ++c) fix_sem[r * C + c] = 0;
        if (fix_inst) for (int c = 0; c < K; ++c) fix_inst[r * K + c] = 0;
        for (int i = 0; i < N; ++i) {
            const int64_t s = r * N + i;
            const float* rw = raw + s * stride_s;
            float dist = (i + 1 < N) ? (z[s + 1] - z[s]) : 1e10f;
            dist = dist * dn;
            float sg = rw[3 * stride_c];
            if (noise) sg = sg + noise[s];
            sg = sg > 0.0f ? sg : 0.0f;
            const float alpha = 1.0f - expf(-(sg * dist));
            const float w = alpha * T;
            T = T * ((1.0f - alpha) + 1e-10f);
            if (weights) weights[s] = w;
            for (int a = 0; a < 3; ++a) {
                const float c = 1.0f / (1.0f + expf(-rw[a * stride_c]));
                o_rgb[a] += w * c;
            }
            o_d += w * z[s];
            o_a += w;
            if (sem && C > 0) {
                if (sem_mode == 0) {
                    for (int c = 0; c < C; ++c) sem[r * C + c] += w * rw[(4 + c) * stride_c];
                } else {
                    float m = -INFINITY, sum = 0;
                    for (int c = 0; c < C; ++c) { const float v = rw[(4 + c) * stride_c]; if (v > m) m = v; }
                    for (int c = 0; c < C; ++c) { tmp[c] = expf(rw[(4 + c) * stride_c] - m); sum += tmp[c]; }
                    for (int c = 0; c < C; ++c) sem[r * C + c] += w * (tmp[c] / sum);
                }
            }
            if (inst && K > 0) {
                if (sem_mode == 0) {
                    for (int c = 0; c < K; ++c) inst[r * K + c] += w * rw[(4 + C + c) * stride_c];
                } else {
                    float m = -INFINITY, sum = 0;
                    for (int c = 0; c < K; ++c) { const float v = rw[(4 + C + c) * stride_c]; if (v > m) m = v; }
                    for (int c = 0; c < K; ++c) { tmp[c] = expf(rw[(4 + C + c) * stride_c] - m); sum += tmp[c]; }
                    for (int c = 0; c < K; ++c) inst[r * K + c] += w * (tmp[c] / sum);
                }
            }
            if (fix_sem && label_sem) { const int l = label_sem[s]; if (l >= 0 && l < C) fix_sem[r * C + l] += w; }
            if (fix_inst && label_inst) { const int l = label_inst[s]; if (l >= 0 && l < K) fix_inst[r * K + l] += w; }
        }
        if (white_bkgd) for (int a = 0; a < 3; ++a) o_rgb[a] = o_rgb[a] + (1.0f - o_a);
        if (rgb) for (int a = 0; a < 3; ++a) rgb[r * 3 + a] = o_rgb[a];
        if (depth) depth[r] = o_d;
        if (acc) acc[r] = o_a;
    }
    free(tmp);
}

/* ---------------------------------------------------------------- a7: sample_pdf
 * SURVEY 8a row a7.  Torch-as-written order:
 *   bins_k = .5*(z_{k+1}+z_k), k=0..Nc-2            (Nc-1 bins)
 *   w_j = weights_{j+1} + 1e-5,  j=0..Nc-3          (Nc-2 weights)
 *   total = torch.sum(w)                            (pnro_torch_sum)
 *   pdf_j = w_j / total
 *   cdf_0 = 0; cdf_{j+1} = fl32(sum_{i<=j} (double) pdf_i)   (torch.cumsum: running sum in double; Nc-1 entries)
 *   u = torch.linspace(0, 1, Nf) when u==NULL (det) else given
 *   inds = #{k : cdf_k <= u}  (searchsorted right=True); below=max(inds-1,0);
 *   above=min(inds, Nc-2); denom = cdf[above]-cdf[below]; denom<1e-5 -> 1
 *   t=(u-cdf[below])/denom;  z_s = bins[below] + t*(bins[above]-bins[below])
 * Outputs: z_samples (R,Nf) (unsorted, in u order), inds_out (R,Nf) int32 (may be NULL). */
PNRO_API void pnro_sample_pdf(const float* z, const float* weights, const float* u,
                              int64_t R, int Nc, int Nf, float* z_samples, int32_t* inds_out)
{
    const int nb = Nc - 1, nw = Nc - 2;
    float* bins = (float*)malloc(sizeof(float) * (size_t)nb);
    float* cdf = (float*)malloc(sizeof(float) * (size_t)nb);
    float* wbuf = (float*)malloc(sizeof(float) * (size_t)(nw > 0 ? nw : 1));
    for (int64_t r = 0; r < R; ++r) {
        const float* zr = z + r * Nc;
        const float* wr = weights + r * Nc;
        for (int k = 0; k < nb; ++k) bins[k] = 0.5f * (zr[k + 1] + zr[k]);
        for (int j = 0; j < nw; ++j) wbuf[j] = wr[j + 1] + 1e-5f;
        const float total = pnro_torch_sum(wbuf, nw);
        cdf[0] = 0.0f;
        double run = 0.0;
        for (int j = 0; j < nw; ++j) {
            const float p = wbuf[j] / total;
            run += (double)p;
            cdf[j + 1] = (float)run;
        }
        for (int i = 0; i < Nf; ++i) {
            const float uu = u ? u[r * Nf + i] : pnro_linspace01(i, Nf);
            int lo = 0, hi = nb; /* upper_bound: first k with cdf[k] > uu */
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= uu) lo = mid + 1; else hi = mid; }
            const int inds = lo;
            const int below = inds - 1 > 0 ? inds - 1 : 0;
            const int above = inds < nb - 1 ? inds : nb - 1;
            float denom = cdf[above] - cdf[below];
            if (denom < 1e-5f) denom = 1.0f;
            const float t = (uu - cdf[below]) / denom;
            const float span = bins[above] - bins[below];
            const float m = t * span;
            z_samples[r * Nf + i] = bins[below] + m;
            if (inds_out) inds_out[r * Nf + i] = inds;
        }
    }
    free(bins); free(cdf); free(wbuf);
}

static int cmp_float(const void* a, const void* b)
{
    const float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}

/* z_fine = sort(cat(z_coarse, z_samples))   (R, Nc+Nf) */
PNRO_API void pnro_merge_sorted(const float* z, const float* zs, int64_t R, int Nc, int Nf, float* out)
{
    const int Nt = Nc + Nf;
    for (int64_t r = 0; r < R; ++r) {
        float* o = out + r * Nt;
        memcpy(o, z + r * Nc, sizeof(float) * (size_t)Nc);
        memcpy(o + Nc, zs + r * Nf, sizeof(float) * (size_t)Nf);
        qsort(o, (size_t)Nt, sizeof(float), cmp_float);
    }
}

/* ---------------------------------------------------------------- a8: ray / 3D bbox prior
 * SURVEY 8a row a8.  Oriented boxes: box (M,15) = centre c(3), rotation rows Rm(9)
 * (row a = box axis a in world frame), half extents e(3);  box_ids (M,2) int32 =
 * (semantic id, instance id).  Slab test in the box frame, fixed op order:
 *   p = o - c;  ol_a = (R_a0 p0 + R_a1 p1) + R_a2 p2;  dl_a likewise with d
 *   inv = 1/dl_a; t1 = (-e_a - ol_a)*inv; t2 = (e_a - ol_a)*inv
 *   tmin = fmax(tmin, fmin(t1,t2)); tmax = fmin(tmax, fmax(t1,t2)); init tmin=near, tmax=far
 *   hit iff tmin <= tmax
 * Per ray the max_hits NEAREST hits (smallest t_in; ties: lower box index) in ascending (t_in, box index)
 * order: hit_t (R,max_hits,2) = (t_in,t_out), hit_box (R,max_hits) int32 (-1 = none), hit_count (R) int32 =
 * the TRUE number of intersected boxes (> max_hits: the farthest were dropped; consumers use
 * min(hit_count, max_hits) entries). */
PNRO_API void pnro_bbox_hits(const float* rays, int64_t R, const float* box, int M, int max_hits,
                             float* hit_t, int32_t* hit_box, int32_t* hit_count)
{
    for (int64_t r = 0; r < R; ++r) {
        const float* ry = rays + r * 8;
        int cnt = 0;
        for (int h = 0; h < max_hits; ++h) {
            hit_box[r * max_hits + h] = -1;
            hit_t[(r * max_hits + h) * 2 + 0] = 0.0f;
            hit_t[(r * max_hits + h) * 2 + 1] = 0.0f;
        }
        for (int m = 0; m < M; ++m) {
            const float* b = box + m * 15;
            const float p0 = ry[0] - b[0], p1 = ry[1] - b[1], p2 = ry[2] - b[2];
            float tmin = ry[6], tmax = ry[7];
            for (int a = 0; a < 3; ++a) {
                const float* Ra = b + 3 + 3 * a;
                const float ol = (Ra[0] * p0 + Ra[1] * p1) + Ra[2] * p2;
                const float dl = (Ra[0] * ry[3] + Ra[1] * ry[4]) + Ra[2] * ry[5];
                const float inv = 1.0f / dl;
                const float e = b[12 + a];
                const float t1 = (-e - ol) * inv, t2 = (e - ol) * inv;
                tmin = fmaxf(tmin, fminf(t1, t2));
                tmax = fminf(tmax, fmaxf(t1, t2));
            }
            if (tmin <= tmax) {
                /* keep the max_hits nearest intervals in ascending (t_in, box index) order */
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
                ++cnt;   /* the TRUE number of intersected boxes; > max_hits means the farthest were dropped */
            }
        }
        hit_count[r] = cnt;
    }
}

/* cfg.bbox_sampling = "hull" (a switch: the reference's choice is not in the mount): near / far of a ray that
 * hits boxes become the hull [min t_in, max t_out] of its kept intervals; rays without a hit are copied. */
PNRO_API void pnro_restrict_rays(const float* rays, int64_t R, const float* hit_t, const int32_t* hit_count, int max_hits, float* out)
{
    for (int64_t r = 0; r < R; ++r) {
        memcpy(out + r * 8, rays + r * 8, 8 * sizeof(float));
        const int cnt = hit_count[r] < max_hits ? hit_count[r] : max_hits;
        if (cnt <= 0) continue;
        float lo = hit_t[(r * max_hits) * 2], hi = hit_t[(r * max_hits) * 2 + 1];
        for (int h = 1; h < cnt; ++h) {
            lo = fminf(lo, hit_t[(r * max_hits + h) * 2]);
            hi = fmaxf(hi, hit_t[(r * max_hits + h) * 2 + 1]);
        }
        out[r * 8 + 6] = lo;
        out[r * 8 + 7] = hi;
    }
}

/* Per-sample fixed labels: among the ray's hits with t_in <= z <= t_out take the smallest
 * t_in (ties: first in list); label = box_ids of that box, else -1.  Outputs (R,N) int32. */
PNRO_API void pnro_sample_labels(const float* z, int64_t R, int N, const float* hit_t,
                                 const int32_t* hit_box, const int32_t* hit_count, int max_hits,
                                 const int32_t* box_ids, int32_t* label_sem, int32_t* label_inst)
{
    for (int64_t r = 0; r < R; ++r)
        for (int i = 0; i < N; ++i) {
            const float zz = z[r * N + i];
            int best = -1; float bt = 0.0f;
            const int cnt = hit_count[r] < max_hits ? hit_count[r] : max_hits;
            for (int h = 0; h < cnt; ++h) {
                const float ti = hit_t[(r * max_hits + h) * 2], to = hit_t[(r * max_hits + h) * 2 + 1];
                if (ti <= zz && zz <= to && (best < 0 || ti < bt)) { best = h; bt = ti; }
            }
            int ls = -1, li = -1;
            if (best >= 0) { const int m = hit_box[r * max_hits + best]; ls = box_ids[m * 2]; li = box_ids[m * 2 + 1]; }
            label_sem[r * N + i] = ls;
            label_inst[r * N + i] = li;
        }
}

/* ---------------------------------------------------------------- a5: dense layer helper
 * y = act(W x + b), W (out,in) row-major, sequential fp32 fmaf-free dot (k ascending).
 * emulate_bf16: round x and W to bf16 (RNE) before the product (products then exact in
 * fp32) -- the arithmetic the MFMA bf16 path performs, up to accumulation order.
 * Used by tests for small single-layer KATs; the full MLP oracle is torch_oracle.py. */
static float bf16_round(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f; /* NaN */
    const uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb; u &= 0xffff0000u;
    float o; memcpy(&o, &u, 4); return o;
}
PNRO_API void pnro_bf16_round(const float* x, int64_t n, float* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = bf16_round(x[i]);
}
PNRO_API void pnro_linear(const float* x, int64_t n, int in, const float* W, const float* b, int out_f,
                          int relu, int emulate_bf16, float* y)
{
    for (int64_t s = 0; s < n; ++s)
        for (int o = 0; o < out_f; ++o) {
            float acc = b ? b[o] : 0.0f;
            for (int k = 0; k < in; ++k) {
                float xv = x[s * in + k], wv = W[(int64_t)o * in + k];
                if (emulate_bf16) { xv = bf16_round(xv); wv = bf16_round(wv); }
                acc = fmaf(xv, wv, acc);
            }
            if (relu && acc < 0.0f) acc = 0.0f;
            y[s * out_f + o] = acc;
        }
}

PNRO_API int pnro_version(void) { return 1; }
