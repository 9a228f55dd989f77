// K3-wgrad: weight gradients of the fused NeRF MLP, bf16 MFMA, fp32 accumulate.
//   dW_l = dY_l^T X_l   (M = layer outputs, N = layer inputs, reduction over the S samples),   db_l = sum_s dY_l
// for every Linear of the network, straight from the slot-ordered bf16 buffers the training forward (acts: X) and
// the data-gradient pass (dys: dY) leave in HBM (pnr_train_layout).  No reference file exists in the mount
// (SURVEY.md 0); the arithmetic is what autograd does for nn.Linear.
//
// Both MFMA operands are needed k-major per lane (8 consecutive SAMPLES of one feature), so both go through LDS and
// come back through gfx950's transpose read:
//   * a workgroup owns one job (a (dY region, X region) pair) and one slab of samples; it copies KT = 64-sample
//     tiles of both regions into LDS by LDS-DMA, verbatim: in the saved-tensor layout (pnr_mlp_layout.h) a tile is one
//     contiguous block (1 KiB = 8 lines per wave instruction, double buffered, one barrier per tile);
//   * ds_read_b64_tr_b16 hands a lane 4 consecutive samples of its feature; a 16-lane group addresses a
//     [4 samples][2 chunks of 8 features] block = two half-lines, two reads make one MFMA operand; the layout's XOR of the
//     sample position with bit 1 of the chunk index puts the four groups of a half-wave on all 64 banks once;
//   * one straight-line instance of the loop per job shape (wg_body<MB, NB>), fragment reads as inline asm with
//     hand-counted waits and an explicit software pipeline -- see the note at the reads;
//   * db comes from one more MFMA per row block against an all-ones operand (the kernel is HBM-bound: ~13.2 KB per
//     sample over all jobs, ~1.3 MFLOP);
//   * every (job, slab) writes its partial sums; k_wgrad_reduce adds the slabs in a fixed order (deterministic)
//     and un-permutes slots into the nn.Linear layout, so no atomics and no host-side index maps are needed.
// Contract: rows S..S_pad of every dys region hold zeros and of every acts region finite values (the kernels that write
// them do so); no row masks here.
#include <hip/hip_runtime.h>
#include <string.h>
#include <type_traits>

#include "pnr_common.h"
#include "pnr_mlp_layout.h"
#include "pnr_mlp_plan.h"

int pnr_mlp_validate(const pnr_mlp_desc* d);

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(8))) short i16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;

// LDS: two buffers of WG_KT-sample tile pairs (A = dY tile | B = X tile), the next pair requested while the current one
// is consumed.  Measured at 786 K samples (tools/train_profile.py, same box) with the round-1 loop: 64 x 2 buffers 2.89 ms,
// 48 x 3: 3.19, 32 x 4: 3.61 -- deeper prefetch of smaller tiles does not pay: the loop was not waiting for HBM latency but
// for itself (every fragment read and MFMA sat behind a run-time shape test and its own s_waitcnt).  Hence the shape
// specialisation and the explicit software pipeline below.
#ifndef WG_KT
#define WG_KT 64                    /* samples per LDS tile (a multiple of the 16-sample k-step) */
#endif
#ifndef WG_DMA_AUX
#define WG_DMA_AUX 2                /* cache policy of the tile loads: 0 default, 2 nt (every byte is read once per job): -3 % */
#endif
#ifndef WG_SLAB
#define WG_SLAB 8192
#endif
#ifndef WG_SLAB_MAJOR
#define WG_SLAB_MAJOR 0           /* 1: the jobs of one slab side by side (re-reads of shared regions from cache): measured +12 % time */
#endif
#ifndef WG_ISSUE_SPLIT
#define WG_ISSUE_SPLIT 0            /* 1: the next tile's LDS-DMA pieces are issued between the k-steps instead of en bloc */
#endif
#ifndef WG_NBUF
#define WG_NBUF 2                   /* LDS ring: tile pairs resident; WG_NBUF - 1 requested ahead.  After the rewrite, same box:
                                       64 x 2: 1.98 ms, 32 x 4: 2.00-2.11, 32 x 5: 2.04-2.11, 16 x 8: 2.75-2.80 */
#endif
#define WG_TILE_BYTES (WG_KT * 512) /* one operand tile at the widest region (256 slots) */
static_assert(WG_KT % 16 == 0 && WG_NBUF >= 2 && WG_NBUF * (2 * WG_TILE_BYTES + WG_KT * 64) <= 163840, "wgrad tile ring does not fit the 160 KiB LDS");
#ifndef WG_STACK_HEADS
#define WG_STACK_HEADS 1            /* sem0 + inst0 as one stacked-dY job (A/B knob) */
#endif
#define WG_MAX_JOBS 24
#define WG_BIAS_COLS 32             /* partial block: [ma][nb + 32], column nb = row sum (bias gradient) */

struct WgJob {
    int64_t a_off, b_off;           // element offsets of the dY region (in dys) and the X region (in acts)
    int64_t a2_off;                 // -1, or a SECOND dY region of the same width stacked under the first (rows ma/2 .. ma-1):
                                    // two layers that read the same X (the semantic and instance heads' first Linear) in one pass
    int64_t p_off;                  // float offset of this job's partials: [n_slabs][ma][nb + WG_BIAS_COLS]
    int64_t a3_off, p3_off;         // -1, or an EXTRA 32-slot dY region multiplied by the same X in the same pass (round 6: the [rgb, sigma]
                                    // gradient block beside dY_feature -- alpha_linear reads h like feature_linear does, and its own
                                    // job read all of h again for ONE useful row): its partials [n_slabs][32][nb + WG_BIAS_COLS] at p3_off
    int ma, nb;                     // widths in slots: 32, 64, 128 or 256 (ma: both stacked regions together)
};
#ifndef WG_MERGE_ALPHA
#define WG_MERGE_ALPHA 1            /* the alpha row rides in the feature job (A/B knob) */
#endif
#define WG_EXTRA_TILE_BYTES (WG_KT * 64)    /* one 32-slot bf16 tile */
#define WG_LDS_BYTES (WG_NBUF * (2 * WG_TILE_BYTES + WG_EXTRA_TILE_BYTES))
struct WgArgs {
    const uint16_t* acts; const uint16_t* dys;
    float* partial;
    int S, slab, n_slabs, n_jobs;
    WgJob job[WG_MAX_JOBS];
};

// How the 8 waves tile an (MB x NB)-block gradient: a WM x WN grid of waves, TM x TN blocks of 32 x 32 per wave.  As many
// waves as the shape allows, then the fewest fragment reads per k-step (TM + TN).
template <int N, class F, int I = 0>
__device__ __forceinline__ void pp_static_for_wg(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); pp_static_for_wg<N, F, I + 1>(static_cast<F&&>(f)); }
}
// N fragments: ds_read_b64_tr_b16 at addr[i] + OFF (samples 0..3 of this lane's k-half) and addr2[i] + OFF (samples 4..7)
template <int OFF, int N>
__device__ __forceinline__ void wg_load_frags(const int (&addr)[N], const int (&addr2)[N], bf16x8 (&f)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        i16x4 lo, hi4;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr[i]), "n"(OFF));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi4) : "v"(addr2[i]), "n"(OFF));
        f[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
    }
}
template <int PENDING, int TM, int TN>
__device__ __forceinline__ void wg_landed(bf16x8 (&fa)[TM], bf16x8 (&fb)[TN])
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fa[0]) : "n"(PENDING));
#pragma unroll
    for (int i = 1; i < TM; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[j]));
}
__device__ __forceinline__ void wg_touch(bf16x8& f) { asm volatile("" : "+v"(f)); }     // (covered by the same counted wait)
struct WgGridT { int wm, wn; };
constexpr WgGridT wg_grid(int MB, int NB)
{
    WgGridT best{1, 1};
    int score = -1;
    for (int wm = 1; wm <= 8 && wm <= MB; wm *= 2) {
        int wn = 8 / wm < NB ? 8 / wm : NB;
        const int sc = wm * wn * 100 - (MB / wm + NB / wn);
        if (sc > score) { score = sc; best = WgGridT{wm, wn}; }
    }
    return best;
}

// One (job, slab) of shape MB x NB blocks: partial[ma][nb + 32] = sum over the slab's samples of dY^T [X | 1].
// STACK: the dY tile is two verbatim sub-tiles of MB / 2 row blocks each (WgJob::a2_off), one behind the other in the LDS.
// EXTRA: one more 32-slot dY region (WgJob::a3_off) against the same X tile: its row block x the NB column blocks is spread over the
// waves -- wave (wm, wn) takes column block wn * TN + wm, whose X fragment it reads anyway (needs WM == TN) -- and accumulated in
// the same k order as a job of its own would (bit-identical partials).
template <int MB, int NB, bool STACK = false, bool EXTRA = false>
__device__ __forceinline__ void wg_body(const WgArgs& a, char* const smem, const int jb, const int slab, const int lane, const int wave)
{
    constexpr WgGridT G = wg_grid(MB, NB);
    constexpr int WM = G.wm, WN = G.wn, TM = MB / WM, TN = NB / WN;
    constexpr int MBS = STACK ? MB / 2 : MB;                        // row blocks per dY region
    constexpr int cprA = MBS * 4, cprB = NB * 4;                    // 16 B chunks per tile row (of one region)
    constexpr int subA = WG_KT * cprA * 16;                         // bytes of one dY (sub-)tile
    static_assert(!STACK || (MB % 2 == 0 && TM <= MBS && MBS % TM == 0), "a wave's row blocks lie in one of the stacked regions");
    constexpr int piecesS = WG_KT * cprA / 64;                      // 1 KiB pieces of one dY (sub-)tile
    constexpr int piecesE = EXTRA ? WG_KT * 4 / 64 : 0;             // the extra region's tile: 4 chunks per row
    static_assert(!EXTRA || (WM == TN && WM * WN == 8 && !STACK), "extra row block: one column block per wave");
    constexpr int piecesA = (STACK ? 2 : 1) * piecesS, piecesB = WG_KT * cprB / 64, pieces = piecesA + piecesB + piecesE;
    constexpr int NQ = (pieces + 7) / 8;                            // LDS-DMA pieces per wave and tile (at most)
    constexpr int NKS = WG_KT / 16;

    const uint16_t* const Ag = a.dys + a.job[jb].a_off;
    const uint16_t* const Bg = a.acts + a.job[jb].b_off;
    const int s_begin = slab * a.slab;
    const int s_end = a.S < s_begin + a.slab ? a.S : s_begin + a.slab;
    const int ntiles = (s_end - s_begin + WG_KT - 1) / WG_KT;

    // ---- LDS-DMA: a tile (WG_KT samples of a region) is contiguous in the saved-tensor layout (pnr_mlp_layout.h); it is
    // copied verbatim, 1 KiB (8 lines) per wave instruction.  Rows past S exist (S_pad) and hold zeros in dys.
    const char* const srcA = reinterpret_cast<const char*>(Ag);         // wave-uniform bases: pnr_dma_piece adds 16 * lane
    const char* const srcA2 = STACK ? reinterpret_cast<const char*>(a.dys + a.job[jb].a2_off) : srcA;
    const char* const srcB = reinterpret_cast<const char*>(Bg);
    const char* const srcE = EXTRA ? reinterpret_cast<const char*>(a.dys + a.job[jb].a3_off) : srcA;
    char* const smemE = smem + WG_NBUF * 2 * WG_TILE_BYTES;         // the extra tiles live behind the ring
    auto issue = [&](int t, int buf, int q0, int q1) {
        const int64_t g0 = (s_begin + t * WG_KT) >> 3;              // first 8-sample group of the tile
        char* const dst = smem + buf * 2 * WG_TILE_BYTES;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q < q0 || q >= q1) continue;
            const int p = wave + 8 * q;
            if (p >= pieces) continue;
            if (EXTRA && p >= piecesA + piecesB) {
                const int pe = p - piecesA - piecesB;
                pnr_dma_piece<WG_DMA_AUX>(srcE + g0 * (4 * 128) + pe * 1024, smemE + buf * WG_EXTRA_TILE_BYTES + pe * 1024, lane * 16);
                continue;
            }
            const bool isA = p < piecesA;
            const int pp = isA ? p : p - piecesA;                   // LDS side: the sub-tiles of a stacked dY lie back to back
            const bool second = STACK && isA && p >= piecesS;
            const char* src = (isA ? (second ? srcA2 : srcA) + g0 * (cprA * 128) : srcB + g0 * (cprB * 128)) + (second ? pp - piecesS : pp) * 1024;
            pnr_dma_piece<WG_DMA_AUX>(src, dst + (isA ? 0 : WG_TILE_BYTES) + pp * 1024, lane * 16);
        }
    };

    // ---- this wave's blocks: rows wm*TM + i, columns wn*TN + j; its bias block: i = wn (< TM)
    // Fragment addresses in a tile [8 sample groups][cpr lines][128 B]: a 16-lane group of ds_read_b64_tr_b16 covers
    // [4 samples][16 features = 2 chunks]; lane a: sample a >> 2 of its k-half (hi), chunk blk*4 + 2*(gq&1) + ((a&3) >> 1),
    // 8 B half a & 1.  The second read of a fragment takes samples +4: position ^ 4, i.e. address ^ 64.
    const bool active = wave < WM * WN;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int al = lane & 15, gq = lane >> 4, hi = gq >> 1;
    const int cin = 2 * (gq & 1) + ((al & 3) >> 1);                 // chunk within the 32-feature block; (chunk >> 1) & 1 = gq & 1
    const int pos = ((al >> 2) ^ ((gq & 1) << 2)) * 16 + (al & 1) * 8;
    int offA[TM], offB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int blk = wm * TM + i, sub = STACK ? blk / MBS : 0;
        offA[i] = sub * subA + (hi * cprA + (blk - sub * MBS) * 4 + cin) * 128 + pos;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) offB[j] = (hi * cprB + (wn * TN + j) * 4 + cin) * 128 + pos;

    const int offE = WG_NBUF * 2 * WG_TILE_BYTES + (hi * 4 + cin) * 128 + pos;      // the extra block: chunks 0..3 of a 4-chunk row
    static_assert(TM <= WN || (TM == 1 && WN == 1), "one bias block per wave at most");
    const int isel = wn;                                            // the row block whose bias (row sum) this wave accumulates
    const bool has_bias = wn < TM;
    f32x16 acc[TM][TN], bacc, eacc;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { bacc[r] = 0.0f; eacc[r] = 0.0f; }
    bf16x8 ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    // tile t has landed once only this wave's pieces of the younger tiles already requested are outstanding (vector-memory
    // operations of a wave complete in order); s_waitcnt takes an immediate: a ladder over the possible counts
    int n_mine = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) n_mine += (wave + 8 * q < pieces) ? 1 : 0;
    auto landed_tile = [&](int t) {
        const int ahead = ntiles - 1 - t < WG_NBUF - 2 ? ntiles - 1 - t : WG_NBUF - 2;
#define PNR_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        switch (n_mine * ahead) {
            PNR_VM(1) PNR_VM(2) PNR_VM(3) PNR_VM(4) PNR_VM(5) PNR_VM(6) PNR_VM(7) PNR_VM(8) PNR_VM(9) PNR_VM(10) PNR_VM(11) PNR_VM(12)
            PNR_VM(13) PNR_VM(14) PNR_VM(15) PNR_VM(16) PNR_VM(17) PNR_VM(18) PNR_VM(19) PNR_VM(20) PNR_VM(21) PNR_VM(22) PNR_VM(23) PNR_VM(24)
            PNR_VM(25) PNR_VM(26) PNR_VM(27) PNR_VM(28) PNR_VM(29) PNR_VM(30) PNR_VM(31) PNR_VM(32)
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
#undef PNR_VM
    };
    if (!active) {      // shapes with fewer than 8 wave tiles: the spare waves only help moving the tiles
#pragma unroll
        for (int d = 0; d < WG_NBUF - 1; ++d)
            if (d < ntiles) issue(d, d, 0, NQ);
        for (int t = 0; t < ntiles; ++t) {
            landed_tile(t);
            __syncthreads();
            if (t + WG_NBUF - 1 < ntiles) issue(t + WG_NBUF - 1, (t + WG_NBUF - 1) % WG_NBUF, 0, NQ);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // Fragment reads are inline asm with hand-counted waits: behind the builtin, hipcc puts s_waitcnt vmcnt(0) in front of
    // every LDS read that follows an LDS-DMA request (it cannot tell the buffers apart) and lgkmcnt(0) in front of every
    // MFMA, which serialises request -> land -> read -> multiply per k-step.  ds_read_b64_tr_b16 hands a lane 4 consecutive
    // samples of its feature; the reads at rows ks*16 and ks*16 + 4 make one k = 16 MFMA operand.
    bf16x8 fa[2][TM], fb[2][TN], fe[2][1];
    int adA[TM], adB[TN], adA2[TM], adB2[TN], adE[1], adE2[1];
    auto load = [&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        wg_load_frags<ks * 2 * cprA * 128, TM>(adA, adA2, fa[ks & 1]);              // k-step ks = sample groups 2ks, 2ks + 1
        wg_load_frags<WG_TILE_BYTES + ks * 2 * cprB * 128, TN>(adB, adB2, fb[ks & 1]);
        if constexpr (EXTRA) wg_load_frags<ks * 2 * 4 * 128, 1>(adE, adE2, fe[ks & 1]);
    };
    // the fragments of k-step ks have landed once at most PENDING younger reads are outstanding (LDS reads return in order)
    auto landed = [&](auto ks_c, auto pending_c) {
        constexpr int ks = decltype(ks_c)::value;
        wg_landed<decltype(pending_c)::value, TM, TN>(fa[ks & 1], fb[ks & 1]);
        if constexpr (EXTRA) wg_touch(fe[ks & 1][0]);
    };
    auto mma = [&](auto ks_c) {
        constexpr int ks = decltype(ks_c)::value;
        bf16x8 fsel = fa[ks & 1][0];
#pragma unroll
        for (int i = 1; i < TM; ++i) fsel = isel == i ? fa[ks & 1][i] : fsel;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
        if (has_bias) bacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fsel, ones, bacc, 0, 0, 0);
        if constexpr (EXTRA) {
            bf16x8 bsel = fb[ks & 1][0];
#pragma unroll
            for (int j = 1; j < TN; ++j) bsel = wm == j ? fb[ks & 1][j] : bsel;
            eacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fe[ks & 1][0], bsel, eacc, 0, 0, 0);
        }
    };
    constexpr int RD = 2 * (TM + TN + (EXTRA ? 1 : 0));             // ds_reads per k-step
    static_assert(RD <= 15, "lgkmcnt is a 4-bit counter");

#pragma unroll
    for (int d = 0; d < WG_NBUF - 1; ++d)
        if (d < ntiles) issue(d, d, 0, NQ);
    int buf = 0;
    for (int t = 0; t < ntiles; ++t) {
        landed_tile(t);                                             // this wave's pieces of tile t
        __syncthreads();                                            // tile t landed for everybody; everybody is done with tile t-1
        const bool more = t + WG_NBUF - 1 < ntiles;
        const int nbuf = buf == 0 ? WG_NBUF - 1 : buf - 1;          // tile t-1's buffer takes tile t + WG_NBUF - 1
#pragma unroll
        for (int i = 0; i < TM; ++i) { adA[i] = offA[i] + buf * 2 * WG_TILE_BYTES; adA2[i] = adA[i] ^ 64; }
#pragma unroll
        for (int j = 0; j < TN; ++j) { adB[j] = offB[j] + buf * 2 * WG_TILE_BYTES; adB2[j] = adB[j] ^ 64; }
        if constexpr (EXTRA) { adE[0] = offE + buf * WG_EXTRA_TILE_BYTES; adE2[0] = adE[0] ^ 64; }
        // software pipeline: the fragments of k-step ks+1 are requested before the MFMAs of k-step ks are issued; the next
        // tile's LDS-DMA goes out behind the first fragment requests, so only those are exposed per tile
        load(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        if (more && !WG_ISSUE_SPLIT) issue(t + WG_NBUF - 1, nbuf, 0, NQ);
        __builtin_amdgcn_sched_barrier(0);
        pp_static_for_wg<NKS>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            if constexpr (ks + 1 < NKS) {
                load(std::integral_constant<int, ks + 1>{});
                landed(ks_c, std::integral_constant<int, RD>{});
            } else landed(ks_c, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (WG_ISSUE_SPLIT && ks < NKS - 1)
                if (more) issue(t + WG_NBUF - 1, nbuf, ks * NQ / (NKS - 1), (ks + 1) * NQ / (NKS - 1));
            mma(ks_c);
            __builtin_amdgcn_sched_barrier(0);
        });
        buf = buf + 1 == WG_NBUF ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- partial sums of this (job, slab): [ma][nb + 32] fp32, slot order on both axes
    constexpr int ldp = NB * 32 + WG_BIAS_COLS;
    float* const P = a.partial + a.job[jb].p_off + (int64_t)slab * (MB * 32) * ldp;
    const int n = lane & 31, hl = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * TM + i) * 32 + pnr_row_of(r, hl);
#pragma unroll
            for (int j = 0; j < TN; ++j) P[(int64_t)row * ldp + (wn * TN + j) * 32 + n] = acc[i][j][r];
            if (has_bias && i == isel && n == 0) P[(int64_t)row * ldp + NB * 32] = bacc[r];
        }
    }
    if constexpr (EXTRA) {      // the extra region's partials: [32][nb + 32] of this slab, column block wn * TN + wm
        float* const PE = a.partial + a.job[jb].p3_off + (int64_t)slab * 32 * ldp;
#pragma unroll
        for (int r = 0; r < 16; ++r) PE[(int64_t)pnr_row_of(r, hl) * ldp + (wn * TN + wm) * 32 + n] = eacc[r];
    }
}

__global__ __launch_bounds__(512, 1) void k_wgrad(const WgArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][A tile | B tile]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#if WG_SLAB_MAJOR
    // slab-major: the jobs of one slab run side by side, so the regions several jobs read (h = X_D: feature, alpha, sem0,
    // inst0; gamma(x): layers 0 and skip+1; dY_views, d[rgb, sigma]: twice each) are re-read while still cached
    const int slab = blockIdx.x / a.n_jobs, jb = blockIdx.x - slab * a.n_jobs;
#else
    const int jb = blockIdx.x / a.n_slabs, slab = blockIdx.x - jb * a.n_slabs;
#endif
    // region widths are 32, 64, 128 or 256 slots: one straight-line instance of the loop per shape
    const int shape = (31 - __builtin_clz(a.job[jb].ma >> 5)) * 4 + (31 - __builtin_clz(a.job[jb].nb >> 5));
    if (a.job[jb].a2_off >= 0) {       // two stacked 128-slot dY regions against one 256-slot X (wg_plan makes no other stack)
        wg_body<8, 8, true>(a, smem, jb, slab, lane, wave);
        return;
    }
    if (a.job[jb].a3_off >= 0) {       // dY_feature (256 slots) + the [rgb, sigma] gradient block against h (256 slots)
        wg_body<8, 8, false, true>(a, smem, jb, slab, lane, wave);
        return;
    }
    switch (shape) {
#define WG_CASE(mb, nb) case (mb) * 4 + (nb): wg_body<1 << (mb), 1 << (nb)>(a, smem, jb, slab, lane, wave); break;
        WG_CASE(0, 0) WG_CASE(0, 1) WG_CASE(0, 2) WG_CASE(0, 3)
        WG_CASE(1, 0) WG_CASE(1, 1) WG_CASE(1, 2) WG_CASE(1, 3)
        WG_CASE(2, 0) WG_CASE(2, 1) WG_CASE(2, 2) WG_CASE(2, 3)
        WG_CASE(3, 0) WG_CASE(3, 1) WG_CASE(3, 2) WG_CASE(3, 3)
#undef WG_CASE
    }
}

// ---- reduction over slabs + un-permutation into the nn.Linear layout
struct WgRed {
    int64_t p_off;                  // partial block of the job
    int ma, nb;
    int row0, n_rows;               // canonical dY features [row0, row0 + n_rows) -> rows 0.. of `out`
    int col_kind, col_L, n_cols;    // canonical X columns: PNR_SEG_FEAT (n_cols features) / PNR_SEG_GX / PNR_SEG_GD (encoding of L bands)
    float* out; int ld, col_off;    // out[row * ld + col_off + col]
    float* out_b;                   // bias gradient (n_rows) or null
};
struct WgRedArgs { const float* partial; int n_slabs, n_items; WgRed item[WG_MAX_JOBS]; };

__device__ __forceinline__ int wg_feat_slot(int f)
{
    const int w = f & 31;
    return (f & ~31) + ((w >> 2) & 1) * 16 + (w & 3) + 4 * (w >> 3);
}
__device__ __forceinline__ int wg_embed_slot(int kind, int L, int col)
{
    const int nv = kind == PNR_SEG_GX ? 32 : 16;
    for (int h = 0; h < 2; ++h)
        for (int v = 0; v < nv; ++v)
            if (pnr_seg_col(kind, L, h, v) == col) return h * nv + v;
    return 0;
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const WgRedArgs a)
{
    const WgRed& it = a.item[blockIdx.y];
    const int ncol = it.n_cols + (it.out_b ? 1 : 0);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= it.n_rows * ncol) return;
    const int o = idx / ncol, i = idx - o * ncol;
    const int srow = wg_feat_slot(it.row0 + o);
    const bool bias = i == it.n_cols;
    const int scol = bias ? it.nb : it.col_kind == PNR_SEG_FEAT ? wg_feat_slot(i) : wg_embed_slot(it.col_kind, it.col_L, i);
    const int ldp = it.nb + WG_BIAS_COLS;
    const float* p = a.partial + it.p_off + (int64_t)srow * ldp + scol;
    const int64_t stride = (int64_t)it.ma * ldp;
    // the slabs' partial sums in slab order; eight loads requested before the first is added (same order, same bits: with one load
    // in flight per thread the kernel waited out ~100 dependent round trips)
    float s = 0.0f;
    int k = 0;
    for (; k + 8 <= a.n_slabs; k += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[(k + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    for (; k < a.n_slabs; ++k) s += p[k * stride];
    if (bias) it.out_b[o] = s;
    else it.out[(int64_t)o * it.ld + it.col_off + i] = s;
}

// ------------------------------------------------------------------------------- host side
static int wg_slab(int64_t S)
{
    // samples per slab: enough slabs to fill the GPU a few times over, few enough to keep the partials small
    // (measured at 786 K samples: 2048 / 4096 / 8192 / 16384 / 32768 samples per slab -> 3.28 / 2.97 / 2.79 / 2.96 / 3.18 ms)
    // (round 4, slab sizes chosen for whole rounds of the 256 CUs: 7680 / 8192 / 8768 / 9216 / 10240 -> 1.90 / 1.84 / 1.81 / 1.81 / 1.82 ms at
    //  786 K samples and 0.69 / 0.63 / 0.64 / 0.68 / 0.70 ms at 262 K: a wash over the two levels, 8192 stays)
    int slab = WG_SLAB;
    while (slab > 1024 && S / slab < 24) slab >>= 1;
    return slab;
}

struct WgPlan { int n, n_red; WgJob job[WG_MAX_JOBS]; WgRed red[WG_MAX_JOBS]; int64_t partial_floats; int slab, n_slabs; };

// grads: pnr_mlp_params_host whose pointers are DEVICE pointers to the gradients (null for a plan used for sizes only)
static void wg_plan(const pnr_mlp_desc& d, int64_t S, const pnr_mlp_params_host* g, WgPlan& pl)
{
    int64_t ao[24], dof[24];
    pnr_train_layout(d, S, ao, dof);
    const int D = d.D, W = d.W, H = d.W / 2;
    const int EXn = 3 + 6 * d.xyz_L, EDn = 3 + 6 * d.dir_L;
    pl.slab = wg_slab(S);
    pl.n_slabs = (int)((S + pl.slab - 1) / pl.slab);
    pl.n = 0;
    pl.n_red = 0;
    int64_t po = 0;
    auto F = [](const float* p) { return const_cast<float*>(p); };
    // a reduction item: rows [row0, row0 + n_rows) of the partial block at p_off -> one weight (and bias) gradient
    auto add_red = [&](int64_t p_off, int ma, int nb, int row0, int n_rows, int ck, int cL, int n_cols,
                       float* out, int ld, int col_off, float* out_b) {
        WgRed& r = pl.red[pl.n_red++];
        r.p_off = p_off; r.ma = ma; r.nb = nb; r.row0 = row0; r.n_rows = n_rows; r.col_kind = ck; r.col_L = cL; r.n_cols = n_cols;
        r.out = out; r.ld = ld; r.col_off = col_off; r.out_b = out_b;
    };
    auto add_job = [&](int64_t a_off, int64_t a2_off, int ma, int64_t b_off, int nb) {
        WgJob& j = pl.job[pl.n++];
        j.a_off = a_off; j.a2_off = a2_off; j.b_off = b_off; j.ma = ma; j.nb = nb; j.p_off = po;
        j.a3_off = -1; j.p3_off = 0;
        po += (int64_t)pl.n_slabs * ma * (nb + WG_BIAS_COLS);
        return j.p_off;
    };
    auto add = [&](int64_t a_off, int ma, int64_t b_off, int nb, int row0, int n_rows, int ck, int cL, int n_cols,
                   float* out, int ld, int col_off, float* out_b) {
        add_red(add_job(a_off, -1, ma, b_off, nb), ma, nb, row0, n_rows, ck, cL, n_cols, out, ld, col_off, out_b);
    };
    const bool have = g != nullptr;
    // trunk
    for (int l = 0; l < D; ++l) {
        const int64_t dy = dof[4 + l];
        float* w = have ? F(g->pts_w[l]) : nullptr;
        float* b = have ? F(g->pts_b[l]) : nullptr;
        if (l == 0) add(dy, W, ao[0], 64, 0, W, PNR_SEG_GX, d.xyz_L, EXn, w, EXn, 0, b);
        else if (l - 1 == d.skip) {
            add(dy, W, ao[1 + l], W, 0, W, PNR_SEG_FEAT, 0, W, w, EXn + W, EXn, b);
            add(dy, W, ao[0], 64, 0, W, PNR_SEG_GX, d.xyz_L, EXn, w, EXn + W, 0, nullptr);
        } else add(dy, W, ao[1 + l], W, 0, W, PNR_SEG_FEAT, 0, W, w, W, 0, b);
    }
    const int64_t Xh = ao[1 + D];
    add(dof[1], W, Xh, W, 0, W, PNR_SEG_FEAT, 0, W, have ? F(g->feature_w) : nullptr, W, 0, have ? F(g->feature_b) : nullptr);
    const bool merge_alpha = WG_MERGE_ALPHA && W == 256;        // (the 8 x 8-block shape: WM == TN == 2)
    const int feat_job = pl.n - 1;
    add(dof[0], H, ao[2 + D], W, 0, H, PNR_SEG_FEAT, 0, W, have ? F(g->views_w) : nullptr, W + EDn, 0, have ? F(g->views_b) : nullptr);
    add(dof[0], H, ao[1], 32, 0, H, PNR_SEG_GD, d.dir_L, EDn, have ? F(g->views_w) : nullptr, W + EDn, W, nullptr);
    add(dof[4 + D], 32, ao[3 + D], H, 0, 3, PNR_SEG_FEAT, 0, H, have ? F(g->rgb_w) : nullptr, H, 0, have ? F(g->rgb_b) : nullptr);
    if (merge_alpha) {
        // alpha_linear reads h as feature_linear does: its dY row (sigma = row 3 of the [rgb, sigma] block) rides in the feature job as
        // an extra row block -- h is read once less (512 + 64 of the 13.2 KB a sample cost this kernel); its bias gradient is the
        // row sum the rgb job (previous add) already forms for the same block
        WgJob& fj = pl.job[feat_job];
        fj.a3_off = dof[4 + D];
        fj.p3_off = po;
        po += (int64_t)pl.n_slabs * 32 * (W + WG_BIAS_COLS);
        add_red(fj.p3_off, 32, W, 3, 1, PNR_SEG_FEAT, 0, W, have ? F(g->alpha_w) : nullptr, W, 0, nullptr);
        add_red(pl.job[pl.n - 1].p_off, 32, H, 3, 1, PNR_SEG_FEAT, 0, 0, nullptr, 1, 0, have ? F(g->alpha_b) : nullptr);
    } else
    add(dof[4 + D], 32, Xh, W, 3, 1, PNR_SEG_FEAT, 0, W, have ? F(g->alpha_w) : nullptr, W, 0, have ? F(g->alpha_b) : nullptr);
    const int64_t Xtap = d.head_tap ? ao[2 + D] : Xh;       // what the heads read: the feature (head_tap 1) or h
    // the first Linear of both heads reads the same X: one pass over it with the two dY regions stacked (X read once instead
    // of twice: 512 of the ~13.8 KB a sample costs this kernel).  The stacked shape exists for 128 + 128 rows x 256 only.
    if (d.head_depth == 1) {
        // one Linear W -> n per head: dW = (the logit gradients, 64 slots)^T x (the tap), one 64 x W job per head
        if (d.n_sem) add(dof[5 + D], 64, Xtap, W, 0, d.n_sem, PNR_SEG_FEAT, 0, W, have ? F(g->sem1_w) : nullptr, W, 0, have ? F(g->sem1_b) : nullptr);
        if (d.n_inst) add(dof[6 + D], 64, Xtap, W, 0, d.n_inst, PNR_SEG_FEAT, 0, W, have ? F(g->inst1_w) : nullptr, W, 0, have ? F(g->inst1_b) : nullptr);
        pl.partial_floats = po;
        return;
    }
    const bool stack = d.n_sem && d.n_inst && H == 128 && W == 256 && WG_STACK_HEADS;
    if (stack) {
        const int64_t p = add_job(dof[2], dof[3], 2 * H, Xtap, W);
        add_red(p, 2 * H, W, 0, H, PNR_SEG_FEAT, 0, W, have ? F(g->sem0_w) : nullptr, W, 0, have ? F(g->sem0_b) : nullptr);
        add_red(p, 2 * H, W, H, H, PNR_SEG_FEAT, 0, W, have ? F(g->inst0_w) : nullptr, W, 0, have ? F(g->inst0_b) : nullptr);
    }
    if (d.n_sem) {
        if (!stack) add(dof[2], H, Xtap, W, 0, H, PNR_SEG_FEAT, 0, W, have ? F(g->sem0_w) : nullptr, W, 0, have ? F(g->sem0_b) : nullptr);
        add(dof[5 + D], 64, ao[4 + D], H, 0, d.n_sem, PNR_SEG_FEAT, 0, H, have ? F(g->sem1_w) : nullptr, H, 0, have ? F(g->sem1_b) : nullptr);
    }
    if (d.n_inst) {
        if (!stack) add(dof[3], H, Xtap, W, 0, H, PNR_SEG_FEAT, 0, W, have ? F(g->inst0_w) : nullptr, W, 0, have ? F(g->inst0_b) : nullptr);
        add(dof[6 + D], 64, ao[5 + D], H, 0, d.n_inst, PNR_SEG_FEAT, 0, H, have ? F(g->inst1_w) : nullptr, H, 0, have ? F(g->inst1_b) : nullptr);
    }
    pl.partial_floats = po;
}

#define WG_ZERO_BYTES 1024          /* head of the workspace, unused since the saved-tensor layout (kept: the partials stay 1 KiB in) */

PNR_EXPORT int64_t pnr_mlp_wgrad_workspace_bytes(const pnr_mlp_desc* desc, int64_t n_samples)
{
    if (pnr_mlp_validate(desc) != PNR_OK || n_samples < 0 || desc->precision != PNR_PREC_BF16) return -1;
    WgPlan pl;
    wg_plan(*desc, n_samples, nullptr, pl);
    return WG_ZERO_BYTES + pl.partial_floats * (int64_t)sizeof(float);
}

PNR_EXPORT int pnr_mlp_wgrad(const pnr_mlp_desc* desc, const void* acts, const void* dys, int64_t n_samples,
                             const pnr_mlp_params_host* grads_dev, void* workspace, void* stream)
{
    int rc = pnr_mlp_validate(desc);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(desc->precision == PNR_PREC_BF16, "pnr_mlp_wgrad: the training path is bf16 only");
    PNR_REQUIRE(desc->n_sem <= PNR_BWD_OUT_SLOTS && desc->n_inst <= PNR_BWD_OUT_SLOTS, "pnr_mlp_wgrad: n_sem, n_inst <= %d", PNR_BWD_OUT_SLOTS);
    PNR_REQUIRE(n_samples >= 1 && n_samples < ((int64_t)1 << 31) - 65536, "pnr_mlp_wgrad: bad sample count");
    PNR_REQUIRE(acts && dys && grads_dev && workspace, "pnr_mlp_wgrad: null pointer");
    PNR_REQUIRE((((uintptr_t)acts | (uintptr_t)dys | (uintptr_t)workspace) & 15) == 0, "pnr_mlp_wgrad: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    WgPlan pl;
    wg_plan(*desc, n_samples, grads_dev, pl);
    PNR_REQUIRE(pl.n <= WG_MAX_JOBS && pl.n_red <= WG_MAX_JOBS, "pnr_mlp_wgrad: too many jobs");
    for (int i = 0; i < pl.n_red; ++i)  // out_b is null by design for the second job of a concatenated weight
        PNR_REQUIRE(pl.red[i].out || pl.red[i].n_cols == 0, "pnr_mlp_wgrad: a weight-gradient pointer of grads_dev is null");
    PNR_REQUIRE(grads_dev->alpha_b && grads_dev->rgb_b && grads_dev->feature_b && grads_dev->views_b,
                "pnr_mlp_wgrad: a bias-gradient pointer of grads_dev is null");
    WgArgs a;
    memset(&a, 0, sizeof(a));
    a.acts = (const uint16_t*)acts; a.dys = (const uint16_t*)dys;
    a.partial = (float*)((char*)workspace + WG_ZERO_BYTES);
    a.S = (int)n_samples; a.slab = pl.slab; a.n_slabs = pl.n_slabs; a.n_jobs = pl.n;
    for (int i = 0; i < pl.n; ++i) a.job[i] = pl.job[i];
    static thread_local bool attr_set = false;
    if (!attr_set) {
        PNR_HIP(hipFuncSetAttribute((const void*)k_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_wgrad, dim3(pl.n * pl.n_slabs), dim3(512), WG_LDS_BYTES, st, a);
    PNR_CHECK_LAUNCH("pnr_mlp_wgrad");
    WgRedArgs r;
    memset(&r, 0, sizeof(r));
    r.partial = a.partial; r.n_slabs = pl.n_slabs; r.n_items = pl.n_red;
    int maxel = 0;
    for (int i = 0; i < pl.n_red; ++i) {
        r.item[i] = pl.red[i];
        const int el = pl.red[i].n_rows * (pl.red[i].n_cols + 1);
        if (el > maxel) maxel = el;
    }
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((maxel + 255) / 256, pl.n_red), dim3(256), 0, st, r);
    PNR_CHECK_LAUNCH("pnr_mlp_wgrad (reduce)");
    return PNR_OK;
}
