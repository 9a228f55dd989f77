// Fused compositing epilogue of the inference MLP (SURVEY.md 8a rows a5 + a6 in one pass): instead of writing the 4 + C + K raw
// channels of every sample to HBM (324 B per sample at 45 / 32 heads) and reading them back in k_composite, the wave that
// evaluated a 32-sample tile reduces it on the spot to ONE partial record per tile -- 6 + 2 (C + K) floats = 20 B per sample:
//   Q   = prod_i (1 - alpha_i + 1e-10)                      the tile's transmittance factor
//   lw_i = alpha_i * prod_{j < i, j in tile} (1 - alpha_j + 1e-10)   the weight of sample i as if the ray started at the tile
//   S_c = sum_i lw_i v_ci   for v = 1 (acc), z (depth), sigmoid(rgb), the semantic / instance logits, and the fixed
//         (bbox-prior) fields [label_i == c]
// and k_composite_combine (pnr_composite.hip) finishes a ray from its N / 32 records: out_c = sum_k T_k S_c(k),
// T_k = prod_{k' < k} Q_k'; weights_i = T_k lw_i when the caller wants them (coarse level -> sample_pdf).
// Same formulas as k_composite (pnr_composite.hip; no reference file is mounted): alpha = 1 - exp(-relu(sigma)
// * dist * |d|), dist = z_{i+1} - z_i, 1e10 for the ray's last sample.  The sums are associated differently (per tile, then
// over tiles) than in k_composite: results agree to fp32 rounding, not bit for bit.
// Requires N % 32 == 0 (a tile never straddles two rays) and the plan order "appearance first" (sigma before the logits).
#pragma once
#include "pnr_lane_ops.h"

#define PNR_FUSE_FIX_SCALE 1073741824.0f      /* 2^30: fixed-point bins of the bbox-prior histograms (a tile's lw sum to <= 1) */

// record layout (floats): [0] Q  [1] acc  [2] depth  [3..5] rgb  [6 .. 6+C) semantic  [6+C .. 6+C+K) instance
//                         [6+C+K .. 6+2C+K) fixed semantic  [.. 6+2C+2K) fixed instance; padded to a multiple of 4
__host__ __device__ static inline int pnr_fuse_record_floats(int C, int K) { return (6 + 2 * (C + K) + 3) & ~3; }

struct FuseState {
    float lws[32];     // the tile's 32 local weights as wave-uniform values (v_readlane of lw): the logit blocks multiply by them
    float lw;          // this lane's sample weight inside its tile (both half-waves hold their sample n = lane & 31)
    float zz, zn, dn;  // z of the sample, z of the ray's next sample, |d|
    int samp;          // sample index or -1
    int ls, li;        // bbox-prior labels of the sample (-1: none), fetched with the sample's inputs
    bool last;         // the sample is its ray's last one (interval 1e10), from the input prefetch
    float* rec;        // this tile's record
};

// exclusive prefix product over the 32 lanes of a half-wave (both halves hold the same data), and the total.  All DPP:
// row_shr 1, 2, 4, 8 scan each 16-lane row, row_bcast15 carries row 0's total into row 1 (and row 2's into row 3); the
// __shfl_up form of this scan was a chain of six LDS-crossbar round trips in an L phase that a short M phase cannot cover.
__device__ __forceinline__ void tile_scan(float f, int n, float& excl, float& total)
{
    float x = f;
    x *= dpp_or<0x111>(x, 1.0f);                 // row_shr:1
    x *= dpp_or<0x112>(x, 1.0f);                 // row_shr:2
    x *= dpp_or<0x114>(x, 1.0f);                 // row_shr:4
    x *= dpp_or<0x118>(x, 1.0f);                 // row_shr:8   -> inclusive product inside each row of 16
    x *= dpp_or<0x142, 0xA>(x, 1.0f);            // row_bcast15 into rows 1 and 3 -> inclusive product over the 32 lanes
    excl = dpp_or<0x138>(x, 1.0f);               // wave_shr:1 (lane 32 receives lane 31: overwritten below)
    if (n == 0) excl = 1.0f;
    total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31));
}

// the rgb / sigma block (rows 0..2 rgb, row 3 sigma: registers 0..3 of the hi = 0 half): weights of the tile, Q, acc, depth, rgb,
// the per-sample local weights (optional) and the fixed fields
__device__ __forceinline__ void fuse_rgbs(const MlpArgs& a, FuseState& st, int hi, int n, const f32x16& acc, uint32_t* hist)
{
    const bool valid = st.samp >= 0;
    const float sig = __shfl(acc[3], n, 64);                   // sigma of sample n lives in lane n (hi = 0)
    float dist = st.last ? 1e10f : (st.zn - st.zz);
    dist *= st.dn;
    const float alpha = valid ? 1.0f - expf(-(fmaxf(sig, 0.0f) * dist)) : 0.0f;
    const float f = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
    float excl, q;
    tile_scan(f, n, excl, q);
    const float lw = alpha * excl;
    st.lw = lw;
#pragma unroll
    for (int k = 0; k < 32; ++k) st.lws[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lw), k));
    const bool lo = hi == 0;
    float r[5];
    r[0] = lo ? lw : 0.0f;
    r[1] = lo ? lw * st.zz : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) r[2 + c] = lo ? lw * (1.0f / (1.0f + expf(-acc[c]))) : 0.0f;
    group_sum_batch<32, 5>(r);
    if (lo && n == 0) {
        st.rec[0] = q; st.rec[1] = r[0]; st.rec[2] = r[1]; st.rec[3] = r[2]; st.rec[4] = r[3]; st.rec[5] = r[4];
    }
    if (a.lw && lo && valid) a.lw[st.samp] = lw;
    // fixed (bbox-prior) fields: fixed-point histogram of lw over the labels (integer adds: order-independent, deterministic);
    // LDS operations of one wave execute in order and only this wave touches its histogram: no barrier
    const int C = a.n_sem, K = a.n_inst, CK = C + K;
    const bool want_s = a.lab_s && C, want_i = a.lab_i && K;
    if (want_s || want_i) {
        const int lane = hi * 32 + n;
        for (int c = lane; c < CK; c += 64) hist[c] = 0;
        if (lo && valid) {
            const uint32_t fx = (uint32_t)(lw * PNR_FUSE_FIX_SCALE + 0.5f);
            if (want_s) { const int l = st.ls; if (l >= 0 && l < C) atomicAdd(&hist[l], fx); }
            if (want_i) { const int l = st.li; if (l >= 0 && l < K) atomicAdd(&hist[C + l], fx); }
        }
        float* fixrec = st.rec + 6 + CK;
        for (int c = lane; c < CK; c += 64)
            if ((c < C) ? want_s : want_i) fixrec[c] = (float)hist[c] * (1.0f / PNR_FUSE_FIX_SCALE);
            else fixrec[c] = 0.0f;
    }
}

// a 32-row logit block (rows fb*32 + row(r, hi), valid below n_out) -> record floats at rec_base + row
__device__ __forceinline__ void fuse_logits(FuseState& st, int hi, int n, int fb, int n_out, int rec_base, const f32x16& acc)
{
    float r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = st.lw * acc[k];
    group_sum_batch<32, 16>(r);
    // every lane of a half-wave now holds the 16 sums of its half: lane n = k keeps sum k, one store instruction writes them
    float v = r[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) v = (n == k) ? r[k] : v;
    const int row = fb * 32 + pnr_row_of(n & 15, hi);
    if (n < 16 && row < n_out) st.rec[rec_base + row] = v;
}

// a 32-channel logit block computed TRANSPOSED (PPChunk::mma<SWAP>: lane = channel fb*32 + (lane & 31), register r = sample
// row(r, hi)): the weighted sum over the tile's samples is 16 FMAs against the wave-uniform weights (both halves' row sets are
// evaluated, the lane keeps its own), one exchange between the half-waves, and one contiguous 128-byte store
__device__ __forceinline__ void fuse_logits_t(FuseState& st, int hi, int lane, int fb, int n_out, int rec_base, const f32x16& acc)
{
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        s0 = fmaf(st.lws[pnr_row_of(r, 0)], acc[r], s0);
        s1 = fmaf(st.lws[pnr_row_of(r, 1)], acc[r], s1);
    }
    float s = hi ? s1 : s0;
    s += __shfl_xor(s, 32, 64);
    const int ch = fb * 32 + (lane & 31);
    if (hi == 0 && ch < n_out) st.rec[rec_base + ch] = s;
}
