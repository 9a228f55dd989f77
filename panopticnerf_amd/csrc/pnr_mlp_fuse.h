// Fused compositing epilogue of the inference MLP (SURVEY.md 8a rows a5 + a6 in one pass): instead of writing the 4 + C + K raw
// channels of every sample to HBM (324 B per sample at 45 / 32 heads) and reading them back in k_composite, the wave that
// evaluated a 32-sample tile keeps only what the ray needs of it:
//   per tile   Q   = prod_i (1 - alpha_i + 1e-10)                         the tile's transmittance factor
//              S_c = sum_i lw_i logit_ci   for the C + K semantic / instance logits
//   per sample lw_i = alpha_i * prod_{j < i, j in tile} (1 - alpha_j + 1e-10)   (the weight as if the ray started at the tile)
//              and the three raw colour channels                           (16 B)
// = 26 B per sample at 45 / 32 heads.  k_composite_combine (pnr_composite.hip; one wave per ray) finishes a ray from its
// N / 32 records and N per-sample quadruples: T_k = prod_{k' < k} Q_k', w_i = T_k lw_i, acc / depth / rgb = sum_i w_i {1, z_i,
// sigmoid(rgb_i)}, the fixed (bbox-prior) fields = histogram of w over the labels, logits_c = sum_k T_k S_c(k).
// Round 2 also reduced acc / depth / rgb and the label histogram per tile inside this epilogue (five 32-lane butterflies,
// three sigmoids, LDS atomics, 32 B of stores): ~2700 cycles of an L phase whose partner group has only the 24 MFMAs of the
// rgb / sigma chunk to run -- exposed twice per sample group (per-chunk trace, profiles/r03/r03a).  Everything that is O(1) per
// sample went to the combine kernel, where it costs nothing that matters; what stays here is what needs the logits in registers.
// Same formulas as k_composite (pnr_composite.hip; no reference file is mounted): alpha = 1 - exp(-relu(sigma)
// * dist * |d|), dist = z_{i+1} - z_i, 1e10 for the ray's last sample.  The sums are associated differently than in
// k_composite: results agree to fp32 rounding, not bit for bit.
// Requires N % 32 == 0 (a tile never straddles two rays) and the plan order "appearance first" (sigma before the logits).
#pragma once
#include "pnr_lane_ops.h"

#include "pnr_fuse_record.h"

struct FuseState {
    float lwr[16];     // lane (n, hi), register r: the local weight of sample row(r, hi) -- what a TRANSPOSED logit accumulator (lane =
                       // channel, register r = sample row(r, hi)) is multiplied by.  16 VGPRs; round 2 kept the 32 weights wave-uniform in
                       // SGPRs, which lived across five chunks, spilled to VGPR lanes and cost each logit block 64 FMAs + the reloads
    float lw;          // this lane's own sample weight (lane n of either half)
    float zz, zn, dn;  // z of the sample, z of the ray's next sample, |d|
    int samp;          // sample index or -1
    bool last;         // the sample is its ray's last one (interval 1e10), from the input prefetch
    float* rec;        // this tile's record
};

// exclusive prefix product over lanes 0..31 (the hi = 0 half; the other half computes on its own, unused, values), and the
// total.  All DPP: row_shr 1, 2, 4, 8 scan each 16-lane row, row_bcast15 carries row 0's total into row 1 (and row 2's into
// row 3); the __shfl_up form of this scan was a chain of six LDS-crossbar round trips.
__device__ __forceinline__ void tile_scan(float f, int n, float& excl, float& total)
{
    float x = f;
    x *= dpp_or<0x111>(x, 1.0f);                 // row_shr:1
    x *= dpp_or<0x112>(x, 1.0f);                 // row_shr:2
    x *= dpp_or<0x114>(x, 1.0f);                 // row_shr:4
    x *= dpp_or<0x118>(x, 1.0f);                 // row_shr:8   -> inclusive product inside each row of 16
    x *= dpp_or<0x142, 0xA>(x, 1.0f);            // row_bcast15 into rows 1 and 3 -> inclusive product over the 32 lanes
    excl = dpp_or<0x138>(x, 1.0f);               // wave_shr:1 (lane 32 receives lane 31: unused)
    if (n == 0) excl = 1.0f;
    total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31));
}

// the rgb / sigma block (rows 0..2 rgb, row 3 sigma: registers 0..3 of the hi = 0 half, lane = sample): the tile's local
// weights and Q; per sample (lw, r, g, b) to `ps`
__device__ __forceinline__ void fuse_rgbs(const MlpArgs& a, FuseState& st, int hi, int n, const f32x16& acc)
{
    const bool valid = st.samp >= 0;
    const float sig = acc[3];
    float dist = st.last ? 1e10f : (st.zn - st.zz);
    dist *= st.dn;
    const float alpha = valid ? 1.0f - expf(-(fmaxf(sig, 0.0f) * dist)) : 0.0f;
    const float f = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
    float excl, q;
    tile_scan(f, n, excl, q);
    const float lw = alpha * excl;
#if defined(PNR_FUSE_TRANSPOSED) && !PNR_FUSE_TRANSPOSED
    st.lw = __shfl(lw, n, 64);                   // butterfly form of the logit blocks: both halves need the weight of sample n
#else
    st.lw = lw;
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float w0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lw), pnr_row_of(r, 0)));
        const float w1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lw), pnr_row_of(r, 1)));
        st.lwr[r] = hi ? w1 : w0;
    }
    if (hi == 0) {
        if (valid) a.ps[st.samp] = make_float4(lw, acc[0], acc[1], acc[2]);
        if (n == 0) st.rec[0] = q;
    }
}

// a 32-row logit block (rows fb*32 + row(r, hi), valid below n_out) -> record floats at rec_base + row
__device__ __forceinline__ void fuse_logits(FuseState& st, int hi, int n, int fb, int n_out, int rec_base, const f32x16& acc)
{
    const float wn = st.lw;                              // this lane's sample weight
    (void)n;
    float r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = wn * acc[k];
    group_sum_batch<32, 16>(r);
    // every lane of a half-wave now holds the 16 sums of its half: lane n = k keeps sum k, one store instruction writes them
    float v = r[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) v = (n == k) ? r[k] : v;
    const int row = fb * 32 + pnr_row_of(n & 15, hi);
    if (n < 16 && row < n_out) st.rec[rec_base + row] = v;
}

// a 32-channel logit block computed TRANSPOSED (PPChunk::mma<SWAP>: lane = channel fb*32 + (lane & 31), register r = sample
// row(r, hi)): the weighted sum over the tile's samples is 16 FMAs against the lane's weight registers lwr, one exchange between
// the half-waves, and one contiguous 128-byte store
__device__ __forceinline__ void fuse_logits_t(FuseState& st, int hi, int lane, int fb, int n_out, int rec_base, const f32x16& acc)
{
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s = fmaf(st.lwr[r], acc[r], s);
    s += __shfl_xor(s, 32, 64);
    const int ch = fb * 32 + (lane & 31);
    if (hi == 0 && ch < n_out) st.rec[rec_base + ch] = s;
}

// semantic_activation = softmax (`k_composite` mode 1): the field is composited from softmax(logits) over its
// channels instead of the logits.  NB transposed blocks of one head at once (lane = channel b*32 + (lane & 31), register r =
// sample row(r, hi)): per sample register the max and the denominator are 32-lane butterflies inside the half-wave (four DPP
// steps and one swizzle each), then s_c += lw_r / den_r * e_cr.  exp through v_exp_f32 on (x - max) log2(e) -- the rounding of
// max * log2(e) is common to every channel of the sample and cancels in e / den.  The record keeps its layout (sums over the
// tile's samples of lw * value): k_composite_combine is unchanged.
// Channels past n_out arrive as -inf (pp_logits_merged starts their accumulators there; their weights are zero): exp gives 0.
template <int NB>
__device__ __forceinline__ void fuse_softmax_t(FuseState& st, int hi, int lane, int n_out, int rec_base, f32x16* acc)
{
    constexpr float L2E = 1.4426950408889634f;
    // every butterfly step over the 16 sample registers at once (independent DPP operations back to back); the xor-16 step on the
    // VALU as well (v_permlane16_swap): as ds_swizzle it was a serialised LDS-crossbar round trip per register, +9 % launch time
    float m[16], d[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        m[r] = acc[0][r];
#pragma unroll
        for (int b = 1; b < NB; ++b) m[r] = max_raw(m[r], acc[b][r]);
    }
    group_max_batch<32, 16>(m);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float mm = -m[r] * L2E;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            acc[b][r] = __builtin_amdgcn_exp2f(fmaf(acc[b][r], L2E, mm));
            d[r] = b ? d[r] + acc[b][r] : acc[b][r];
        }
    }
    group_sum_batch<16, 16>(d);
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = xor16_add_swap(d[r]);
    float s[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) s[b] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float wr = st.lwr[r] * __builtin_amdgcn_rcpf(d[r]);
#pragma unroll
        for (int b = 0; b < NB; ++b) s[b] = fmaf(wr, acc[b][r], s[b]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        s[b] += __shfl_xor(s[b], 32, 64);
        const int ch = b * 32 + (lane & 31);
        if (hi == 0 && ch < n_out) st.rec[rec_base + ch] = s[b];
    }
}
