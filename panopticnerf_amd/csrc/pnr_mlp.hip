// K3 (+K2 fused): the NeRF MLP with semantic / instance heads as ONE fused gfx950 kernel.
// Reference: Network / NeRF forward + Embedder (branch not in the
// mount, see include/pnr.h).  Layout of the packed weights: pnr_mlp_layout.h.
//
//  * Every layer is evaluated transposed (H^T = W * H_in^T) so the 32x32 MFMA accumulator of
//    one layer is, after bias/ReLU/convert, directly the B operand of the next: activations
//    stay in VGPRs from gamma(x) to the raw outputs, never touching LDS or HBM.
//  * Weights are the A operand.  They are pre-permuted on the host into 1 KiB fragments in
//    consumption order, cut into chunks of 2 (hidden layers) or 4 (layer 0) 32-row output blocks,
//    and streamed L2 -> LDS with global_load_lds (16 B/lane, lane-linear => conflict-free
//    ds_read_b128) one chunk ahead of the MFMAs into a double buffer; the 8 waves of a workgroup
//    share every fragment; one s_barrier per chunk.
//  * The blocks of a chunk are independent accumulator chains issued round-robin, so no two
//    consecutive MFMAs share an accumulator and the ds_reads slotted between them cost ~6 cycles
//    instead of ~43 (MI355X_MICROARCH.md cycle table).
//  * gamma(x), gamma(d) are computed in registers by the lanes that need them (the two
//    half-waves split the frequency bands), so the 63/27-wide encodings never exist in memory.
//  * bf16 path: v_mfma_f32_32x32x16_bf16, fp32 accumulate, RNE conversion of activations.
//    fp32 path (parity mode): v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain.
//
// HBM traffic per sample: 4 B of z (+32 B/ray) in, 4*(4+C+K) B of raw out; the kernel is
// MFMA-bound (1.19 MFLOP/sample trunk + heads).
//
// Measured and rejected on MI355X (round 1, same box A/B; kept out of the code, numbers in profiles/README.md):
// LDS-counter flow control instead of the barrier (-3 %), DMA two chunks ahead with counted vmcnt (-9 %), staggering
// or spreading the DMA pieces / only half of the waves refilling (-0.5..-13 %), ping-pong unrolling of the layer loop
// (-3 %), one wave per SIMD with 1 or 2 tiles (-30 %), two 4-wave workgroups per CU (-10 %), 4 blocks per hidden
// chunk (-2 %), a phase-split weight stream (half the waves join each chunk's barrier mid-loop, 3 slots: -7..-9 %).
// Debug: -DPNR_TRACE=1 builds stamp s_memtime per chunk phase into LDS (tools/mlp_trace.py).
#include <stdlib.h>
#include <string.h>

#include "pnr_mlp_plan.h"

int pnr_mlp_validate(const pnr_mlp_desc* d);

#include "pnr_mlp_core.h"
#include "pnr_mlp_pp.h"
#ifndef PNR_FUSE_TRANSPOSED
#define PNR_FUSE_TRANSPOSED 1      /* fused epilogue: logit blocks computed transposed (operands swapped), see PPChunk::mma<SWAP> */
#endif
#include "pnr_mlp_fuse.h"
#include "pnr_mlp_tt.h"
#ifndef PNR_OPT_EAGER_EPI
#define PNR_OPT_EAGER_EPI 1
#endif
#ifndef PNR_PP_STORES_LAST
#define PNR_PP_STORES_LAST 0      /* pieces first + counted vmcnt (1): no gain measured; the plain order stays */
#endif
#ifndef PNR_RAW_STORE
#define PNR_RAW_STORE 0           /* cache policy of the raw-output stores: 0 plain, 1 nt, 2 sc1 */
#endif
#ifndef PNR_PP_EPI_IN_M
#define PNR_PP_EPI_IN_M 0
#endif
#ifndef PNR_TRAIN_TILES_EXPERIMENT
#define PNR_TRAIN_TILES_EXPERIMENT 0
#endif
#ifndef PNR_PP_UNROLL2
#define PNR_PP_UNROLL2 0     /* two trunk layers per loop trip (no hand-over copies): -2 % measured (code size) */
#endif
// Hidden layer: inputs = up to two register segments, output -> registers (next B operand).
// save != nullptr (training, bf16): the output block is also stored slot-ordered for the backward.
template <int PREC, int TILES, class CTX, int KIND, int NA, int NB, int NFB_OUT, int MODE, int NOUT>
__device__ __forceinline__ void layer_regs(CTX& c, const uint32_t (&inA)[TILES][NA],
                                           const uint32_t (&inB)[TILES][NB > 0 ? NB : 1],
                                           uint32_t (&out)[TILES][NOUT], uint16_t* save = nullptr,
                                           const int* srow = nullptr)
{
    constexpr int RPB = PrecT<PREC>::RPB;
    constexpr int KS = NA / 4 + NB / 4;
    constexpr int FBC0 = pnr_layer_fbc(KIND, PREC);
    constexpr int FBC = (NFB_OUT % FBC0 == 0) ? FBC0 : 1;      // mirrors pnr_build_plan
    constexpr int G = (CTX::GDB / FBC) < 1 ? 1 : (CTX::GDB / FBC);   // the read window costs 2*G*FBC*4 registers
    static_assert(NOUT >= NFB_OUT * RPB, "output register array too small");
#pragma unroll
    for (int cb = 0; cb < NFB_OUT / FBC; ++cb) {
        c.begin();
        const char* base = c.base();
        f32x16 acc[FBC][TILES];
#pragma unroll
        for (int b = 0; b < FBC; ++b) {
            load_bias(base + FBC * KS * PNR_FRAG_BYTES + b * 128, c.hi, acc[b][0]);
#pragma unroll
            for (int t = 1; t < TILES; ++t) acc[b][t] = acc[b][0];
        }
        c.stamp(2);
        mma_chunk<PREC, TILES, FBC, G, NA, NB>(base + c.lane * 16, inA, inB, acc);
        c.stamp(3);
#pragma unroll
        for (int b = 0; b < FBC; ++b) {
            const int fb = cb * FBC + b;
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                if constexpr (PREC == PNR_PREC_BF16) {
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        uint32_t v = pack_bf16(acc[b][t][2 * p], acc[b][t][2 * p + 1]);
                        // ReLU after rounding (they commute, bit for bit): a negative bf16 is a negative int16,
                        // so one v_pk_max_i16 against 0 gates both halves -- 8 ops per block instead of 16 v_max_f32.
                        if (MODE == MODE_RELU) v = relu_bf16x2(v);
#if PNR_OPT_EAGER_EPI
                        asm volatile("" : "+v"(v));      // materialise here: hipcc otherwise parks all of a layer's pack/ReLU at its end
#endif
                        out[t][fb * RPB + p] = v;
                    }
                    if (save) store_slots(save, NFB_OUT * 32, srow[t], fb, c.hi, &out[t][fb * RPB]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[b][t][r];
                        if (MODE == MODE_RELU) v = fmaxf(v, 0.0f);
                        out[t][fb * RPB + r] = __float_as_uint(v);
                    }
                }
            }
        }
        c.finish(save && PREC == PNR_PREC_BF16 ? 2 * FBC * TILES : 0);
    }
}

// One 32-row output block of one sample tile -> raw channels ch_base + row.  The lane-dependent part of the address
// (the sample, and the half-wave's 4-row offset) is folded into ONE 64-bit base per call; every row then adds a
// wave-uniform offset, so a store costs a compare, an exec mask and one 64-bit add instead of two 64-bit multiplies
// per lane (the stores' address arithmetic was ~1000 cycles per output block).
__device__ __forceinline__ void store_raw_block(const MlpArgs& a, int samp, int hi, int fb, int n_out, int ch_base, const f32x16& acc)
{
    if (samp < 0) return;
    float* const dst0 = a.raw + ((int64_t)samp * a.ss + (int64_t)(4 * hi) * a.sc);
    const int lim = n_out - 4 * hi;                 // row u (of the hi = 0 half) is stored iff u < lim
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = fb * 32 + (r & 3) + 8 * (r >> 2);
        if (u < lim) {
            float* const p = dst0 + (int64_t)(ch_base + u) * a.sc;
#if PNR_RAW_STORE == 1
            __builtin_nontemporal_store(acc[r], p);                                   // nt
#elif PNR_RAW_STORE == 2
            __hip_atomic_store(p, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sc1: write-through, line dropped from L2
#else
            *p = acc[r];
#endif
        }
    }
}

// Output layer: rows [0, n_out) are stored to raw channels ch_base + row (one 32-row block per chunk).
template <int PREC, int TILES, class CTX, int NA, int NB>
__device__ __forceinline__ void layer_out(CTX& c, const uint32_t (&inA)[TILES][NA],
                                          const uint32_t (&inB)[TILES][NB > 0 ? NB : 1], int n_out, int ch_base,
                                          const int (&samp)[TILES])
{
    constexpr int KS = NA / 4 + NB / 4;
    const int nfb = (n_out + 31) >> 5;
#pragma unroll 1
    for (int fb = 0; fb < nfb; ++fb) {
        c.begin();
        const char* base = c.base();
        f32x16 acc[1][TILES];
        load_bias(base + KS * PNR_FRAG_BYTES, c.hi, acc[0][0]);
#pragma unroll
        for (int t = 1; t < TILES; ++t) acc[0][t] = acc[0][0];
        mma_chunk<PREC, TILES, 1, CTX::GDB, NA, NB>(base + c.lane * 16, inA, inB, acc);
        // Chunk hand-over first, stores second: finish()'s vmcnt(0) must cover only the LDS-DMA issued a
        // chunk ago, not the raw stores below (an HBM write round trip per output block otherwise).
        c.finish();
#pragma unroll
        for (int t = 0; t < TILES; ++t) store_raw_block(c.a, samp[t], c.hi, fb, n_out, ch_base, acc[0][t]);
    }
}

// sin and cos of x for the bf16 encoding: Cody-Waite reduction by pi/2 in two exact-product FMAs (k = rint(x * 2/pi);
// the FMA does not round k * C1, so the difference is rounded once: |error| < 1e-7 for |x| up to ~1e6) and the two degree-7 / 8
// minimax polynomials on [-pi/4, pi/4].  ~24 VALU instructions, no branch.  libm's sincosf carries a Payne-Hanek path
// (21 v_mad_u64_u32 per call site) that a wave executes as soon as ONE lane's argument is large -- at the half-wave's
// base band 2^5 every scene coordinate beyond ~3 m is -- and the six calls per sample group sat, fully exposed, in L
// phases whose partner group has nothing to multiply (group start, first views chunk: ~4 % of the fine-level launch).
// Absolute error <= 2e-7, doubled per octave by the recurrences that follow: far below the bf16 rounding (2^-9).
// The fp32 parity mode keeps libm's sincosf at every band.
#ifndef PNR_EMBED_FAST_SINCOS
#define PNR_EMBED_FAST_SINCOS 1
#endif
__device__ __forceinline__ void sincos_cw(float x, float& s, float& c)
{
#if PNR_EMBED_FAST_SINCOS
    const float kf = __builtin_rintf(x * 0.636619772367581343f);
    float r = fmaf(-kf, 1.57079637050628662109375f, x);            // fl(pi/2)
    r = fmaf(-kf, -4.371139000186241e-08f, r);                      // pi/2 - fl(pi/2)
    const int k = (int)kf;
    const float r2 = r * r;
    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(r2, sp, -1.6666654611e-1f);
    const float sn = fmaf(r * r2, sp, r);
    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(r2, cp, 4.166664568298827e-2f);
    cp = fmaf(r2, cp, -0.5f);
    const float cs = fmaf(r2, cp, 1.0f);
    const float so = (k & 1) ? cs : sn, co = (k & 1) ? sn : cs;
    s = __uint_as_float(__float_as_uint(so) ^ ((uint32_t)(k & 2) << 30));
    c = __uint_as_float(__float_as_uint(co) ^ ((uint32_t)((k + 1) & 2) << 30));
#else
    sincosf(x, &s, &c);
#endif
}

// The bf16 encoding in stages (k_mlp_pp computes the NEXT sample group's gamma(x) and this group's gamma(d) piece by piece in
// the memory phases of the last two trunk layers, where the partner wave group's MFMAs leave the VALU idle): sin / cos of the three
// coordinates at the half-wave's base band, then per band one packed register triple (s_x s_y | s_z c_x | c_y c_z) and the
// double-angle step to the next band.  embed_lane<bf16> below is these same calls in a row: bit-identical by construction.
struct EmbedSC { float s[3], c[3]; };
template <int NF>
__device__ __forceinline__ void embed_sincos(float p, int a, int hi, EmbedSC& e)
{
    const float base = hi ? (float)(1 << NF) : 1.0f;
    sincos_cw(p * base, e.s[a], e.c[a]);
}
__device__ __forceinline__ void embed_pack_band(const EmbedSC& e, uint32_t* out3)
{
    out3[0] = pack_bf16(e.s[0], e.s[1]); out3[1] = pack_bf16(e.s[2], e.c[0]); out3[2] = pack_bf16(e.c[1], e.c[2]);
}
// sin 2t = 2 s c, cos 2t = 1 - 2 s^2: the error doubles per octave (<= 2^(NF-1) * 1e-7 ~ 2e-6), far below the bf16 rounding (4e-3)
__device__ __forceinline__ void embed_next_band(EmbedSC& e)
{
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float s2 = 2.0f * e.s[a] * e.c[a], c2 = fmaf(-2.0f * e.s[a], e.s[a], 1.0f);
        e.s[a] = s2; e.c[a] = c2;
    }
}
__device__ __forceinline__ uint32_t embed_pack_xyz(float p0, float p1, float p2, int hi)
{
    return pack_bf16(hi ? p2 : p0, hi ? 0.0f : p1);
}

// gamma() of one 3-vector into this lane's share of the lane vector (pnr_mlp_layout.h).
// NF = frequency bands per half-wave (5 for xyz, 2 for view directions); NV = values per lane.
template <int PREC, int NF, int NV, int NREG>
__device__ __forceinline__ void embed_lane(float p0, float p1, float p2, int hi, uint32_t (&out)[NREG])
{
    if constexpr (PREC == PNR_PREC_BF16) {
        static_assert(NREG == NV / 2 && 1 + 3 * NF <= NREG, "packed lane vector: xyz register + three per band");
        EmbedSC e;
        embed_sincos<NF>(p0, 0, hi, e);
        embed_sincos<NF>(p1, 1, hi, e);
        embed_sincos<NF>(p2, 2, hi, e);
        out[0] = embed_pack_xyz(p0, p1, p2, hi);
#pragma unroll
        for (int fp = 0; fp < NF; ++fp) {
            embed_pack_band(e, &out[1 + 3 * fp]);
            if (fp + 1 < NF) embed_next_band(e);
        }
#pragma unroll
        for (int p = 1 + 3 * NF; p < NREG; ++p) out[p] = 0u;
    } else {
        static_assert(NREG == NV, "fp32 lane vector: one register per value");
        float v[NV];
        v[0] = hi ? p2 : p0;
        v[1] = hi ? 0.0f : p1;
#pragma unroll
        for (int fp = 0; fp < NF; ++fp) {
            const float sc = hi ? (float)(1 << (NF + fp)) : (float)(1 << fp);
            float s, co;
            sincosf(p0 * sc, &s, &co); v[2 + 6 * fp + 0] = s; v[2 + 6 * fp + 3] = co;
            sincosf(p1 * sc, &s, &co); v[2 + 6 * fp + 1] = s; v[2 + 6 * fp + 4] = co;
            sincosf(p2 * sc, &s, &co); v[2 + 6 * fp + 2] = s; v[2 + 6 * fp + 5] = co;
        }
#pragma unroll
        for (int i = 2 + 6 * NF; i < NV; ++i) v[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < NV; ++i) out[i] = __float_as_uint(v[i]);
    }
}

// TRAIN: additionally saves gamma(x), gamma(d) and every layer's (bf16, post-activation) output,
// slot-ordered, for the backward kernel and the weight-gradient GEMMs (MlpArgs::acts).
template <int PREC, int W, int TILES, int WAVES, int MINW, bool TRAIN>
__global__ __launch_bounds__(64 * WAVES, MINW) void k_mlp_fused(const MlpArgs a)
{
    using CTX = Ctx<WAVES, 4, 2, (TRAIN && WAVES == 8) ? PNR_TRAIN_FWD_ISSUERS : 0>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RPB = PrecT<PREC>::RPB;
    constexpr int NFB = W / 32, HFB = W / 64;
    constexpr int HR = NFB * RPB, GR = HFB * RPB;
    constexpr int GXR = PREC == PNR_PREC_BF16 ? 16 : 32;
    constexpr int GDR = PREC == PNR_PREC_BF16 ? 8 : 16;

    CTX c{a, smem, (int)(threadIdx.x & 63), __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)),
          (int)((threadIdx.x & 63) >> 5), 0, 0, {0, 0}, {0, 0}};
#if PNR_TRACE
    c.tr = reinterpret_cast<unsigned long long*>(smem + 2 * a.slot_bytes) + c.wave * PNR_TRACE_CHUNKS * PNR_TRACE_STAMPS;
    c.titer = 0;
#endif
    const int n = c.lane & 31;
    c.start();
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (a.clk) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }

    uint32_t dummy[TILES][1];
#pragma unroll
    for (int t = 0; t < TILES; ++t) dummy[t][0] = 0;

    // per-sample inputs of one tile: o(3) dx | dy dz (near far unused) | z
    struct SampleIn { float4 o4, d4; float zz; };
    auto fetch = [&](int grp, int t) {
        const int s = ((grp * WAVES + c.wave) * TILES + t) * 32 + n;
        const int sl = s < a.S ? s : a.S - 1;
        const int ray = pnr_div_magic(sl, a.n_magic, a.n_shift);
        SampleIn in;
        in.o4 = *reinterpret_cast<const float4*>(a.rays + (int64_t)ray * 8);
        in.d4 = *reinterpret_cast<const float4*>(a.rays + (int64_t)ray * 8 + 4);
        in.zz = a.z[sl];
        return in;
    };
    SampleIn nextin[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) nextin[t] = fetch(blockIdx.x < a.n_groups ? blockIdx.x : 0, t);

    // persistent loop: one group = WAVES * TILES tiles of 32 samples
    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        int samp[TILES], srow[TILES];      // sample of this lane (-1 past the end) / its row in the saved tensors (padded)
        float vd[TILES][3];
        uint32_t ex[TILES][GXR];
        // the last tile of the group is the first to run out of samples
        c.st_full = TRAIN;                  // saves are unmasked: the store count per chunk is exact
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int s = ((grp * WAVES + c.wave) * TILES + t) * 32 + n;
            samp[t] = s < a.S ? s : -1;
            srow[t] = s;
            const float4 o4 = nextin[t].o4, d4 = nextin[t].d4;
            const float zz = nextin[t].zz;
            const float dx = o4.w, dy = d4.x, dz = d4.y;
            // pts = o + d*z: separate multiply and add, as the sampler's pnr_points does
            const float px = __fadd_rn(o4.x, __fmul_rn(dx, zz));
            const float py = __fadd_rn(o4.y, __fmul_rn(dy, zz));
            const float pz = __fadd_rn(o4.z, __fmul_rn(dz, zz));
            const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
            vd[t][0] = dx / nrm; vd[t][1] = dy / nrm; vd[t][2] = dz / nrm;
            embed_lane<PREC, 5, 32, GXR>(px, py, pz, c.hi, ex[t]);
            if constexpr (TRAIN) store_ex(a.acts + a.acts_off[0], srow[t], c.hi, ex[t]);
        }
        auto sv = [&](int idx) -> uint16_t* { return TRAIN ? a.acts + a.acts_off[idx] : nullptr; };
        // ReLU gate bits of a layer output for the data-gradient pass (pnr_train_layout: gate_off)
        auto gv = [&](int idx, auto& regs) {
            if constexpr (TRAIN && PREC == PNR_PREC_BF16) {
#pragma unroll
                for (int t = 0; t < TILES; ++t) save_gates(a.acts + a.gate_off[idx], srow[t], c.hi, regs[t]);
            }
        };

        // trunk
        uint32_t cur[TILES][HR], nxt[TILES][HR];
        layer_regs<PREC, TILES, CTX, PNR_L_TRUNK0, GXR, 0, NFB, MODE_RELU, HR>(c, ex, dummy, cur, sv(2), srow);
        gv(2, cur);
#pragma unroll 1
        for (int l = 1; l < a.D; ++l) {
            if (l - 1 == a.skip)
                layer_regs<PREC, TILES, CTX, PNR_L_TRUNK, GXR, HR, NFB, MODE_RELU, HR>(c, ex, cur, nxt, sv(2 + l), srow);
            else
                layer_regs<PREC, TILES, CTX, PNR_L_TRUNK, HR, 0, NFB, MODE_RELU, HR>(c, cur, dummy, nxt, sv(2 + l), srow);
            gv(2 + l, nxt);
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int i = 0; i < HR; ++i) cur[t][i] = nxt[t][i];
        }
        // next sample group's inputs: issued here so their HBM latency hides under the feature/views layers
        {
            const int g2 = grp + (int)gridDim.x < a.n_groups ? grp + (int)gridDim.x : grp;
#pragma unroll
            for (int t = 0; t < TILES; ++t) nextin[t] = fetch(g2, t);
        }
        layer_regs<PREC, TILES, CTX, PNR_L_FEATURE, HR, 0, NFB, MODE_LINEAR, HR>(c, cur, dummy, nxt, sv(2 + a.D), srow);
        uint32_t ed[TILES][GDR];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            embed_lane<PREC, 2, 16, GDR>(vd[t][0], vd[t][1], vd[t][2], c.hi, ed[t]);
            if constexpr (TRAIN) store_slots(a.acts + a.acts_off[1], 32, srow[t], 0, c.hi, ed[t]);   // ED: 32 slots
        }
        uint32_t g[TILES][GR];
        layer_regs<PREC, TILES, CTX, PNR_L_VIEWS, HR, GDR, HFB, MODE_RELU, GR>(c, nxt, ed, g, sv(3 + a.D), srow);
        gv(3 + a.D, g);
        layer_out<PREC, TILES, CTX, GR, HR>(c, g, cur, 4, 0, samp);
        // panoptic heads, after the appearance branch (plan order).  They read the trunk output h = cur, or -- head_tap 1 -- the
        // feature_linear output: h is dead once sigma is out, so the feature registers simply take its place.
        if (a.head_tap && (a.n_sem || a.n_inst)) {
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int i = 0; i < HR; ++i) cur[t][i] = nxt[t][i];
        }
        if (a.head_depth == 1) {          // one Linear per head, straight from the tap (nothing more to save: the tap is saved)
            if (a.n_sem) layer_out<PREC, TILES, CTX, HR, 0>(c, cur, dummy, a.n_sem, 4, samp);
            if (a.n_inst) layer_out<PREC, TILES, CTX, HR, 0>(c, cur, dummy, a.n_inst, 4 + a.n_sem, samp);
        } else {
        if (a.n_sem) {
            uint32_t sh[TILES][GR];
            layer_regs<PREC, TILES, CTX, PNR_L_SEM0, HR, 0, HFB, MODE_RELU, GR>(c, cur, dummy, sh, sv(4 + a.D), srow);
            gv(4 + a.D, sh);
            layer_out<PREC, TILES, CTX, GR, 0>(c, sh, dummy, a.n_sem, 4, samp);
        }
        if (a.n_inst) {
            uint32_t sh[TILES][GR];
            layer_regs<PREC, TILES, CTX, PNR_L_INST0, HR, 0, HFB, MODE_RELU, GR>(c, cur, dummy, sh, sv(5 + a.D), srow);
            gv(5 + a.D, sh);
            layer_out<PREC, TILES, CTX, GR, 0>(c, sh, dummy, a.n_inst, 4 + a.n_sem, samp);
        }
        }
#if PNR_TRACE
        ++c.titer;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && blockIdx.x == 0 && threadIdx.x == 0) {
        a.clk[0] = __builtin_amdgcn_s_memtime() - clk_c0;
        a.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }   // the refill issued past the last chunk
#if PNR_TRACE
    __syncthreads();
    if (blockIdx.x == PNR_TRACE_WG && a.trace) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(smem + 2 * a.slot_bytes);
        for (int i = threadIdx.x; i < WAVES * PNR_TRACE_CHUNKS * PNR_TRACE_STAMPS; i += blockDim.x) a.trace[i] = src[i];
    }
#endif
}

// ------------------------------------------------------------------------------- ping-pong form (pnr_mlp_pp.h)
// Same arithmetic, same packed image, same register-resident activations as k_mlp_fused<bf16, W, 1 tile, 8 waves>;
// only the time structure of the weight stream differs (two wave groups in phase opposition, 3 LDS slots).
struct NoSide { __device__ __forceinline__ void operator()(int) const {} };
// side(slot): work that is not this layer's, placed between the refill pieces of the L phases (slot = cb * FBC + b, one per
// output block; -1 - cb: behind the last piece of chunk cb's L phase): a wave blocks ~100-200 cycles per LDS-DMA piece while
// the CU's queue is full, and beside a hidden layer's M phase the SIMD's VALU is ~80 % idle -- VALU work parked here costs
// next to nothing.
template <class CTX, int KIND, int NA, int NB, int NFB_OUT, int MODE, int NOUT, int FBC_PLAN = 0, class SIDE = NoSide>
__device__ __forceinline__ void pp_layer_regs(CTX& c, u32x4 (&A)[CTX::P], const uint32_t (&inA)[NA],
                                              const uint32_t (&inB)[NB > 0 ? NB : 1], uint32_t (&out)[NOUT],
                                              uint16_t* save, int srow, SIDE&& side = NoSide{})
{
    constexpr int FBC0 = pnr_layer_fbc(KIND, PNR_PREC_BF16);
    constexpr int FBC = FBC_PLAN > 0 ? FBC_PLAN : (NFB_OUT % FBC0 == 0) ? FBC0 : 1;      // mirrors pnr_build_plan
    using CH = PPChunk<FBC, NA, NB>;
    static_assert(NOUT >= NFB_OUT * 8, "output register array too small");
    f32x4 q[FBC][4];                                            // bias quads in flight across the chunk boundary
#pragma unroll
    for (int cb = 0; cb < NFB_OUT / FBC; ++cb) {
        // L of chunk cb, last part.  For the layer's 2nd, 3rd, ... chunk the first fragments and the bias were already
        // requested (below) in the shadow of the previous chunk's refill / epilogue; only the drain is left.
        if (cb == 0 || !PNR_PP_EARLY) CH::first_frags(c.frag_addr(), A);
        if (cb == 0 || !PNR_PP_EARLY_BIAS) CH::bias_issue(c.bias_addr(), q);
        f32x16 acc[FBC];
        CH::bias_finish(q, acc);                                // waits for every LDS read of the phase
        CH::mma(c.frag_addr(), A, inA, inB, acc, [&](auto... k) { if constexpr (sizeof...(k) == 0) { c.stamp(6); c.barrier(); c.stamp(2); } else c.stamp(k...); });      // L -> M (barrier inside, see mma)
        auto epilogue = [&](int b) {
            const int fb = cb * FBC + b;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                uint32_t v = pack_bf16(acc[b][2 * p], acc[b][2 * p + 1]);
                if (MODE == MODE_RELU) v = relu_bf16x2(v);
                asm volatile("" : "+v"(v));
                out[fb * 8 + p] = v;
            }
        };
        // The M wave reaches the barrier ~200 cycles before its partner finishes L: the pack / ReLU of the chunk's first
        // PNR_PP_EPI_IN_M blocks (complete one MFMA before the chunk's last) is done here, in that slack, instead of in L.
#pragma unroll
        for (int b = 0; b < (PNR_PP_EPI_IN_M < FBC ? PNR_PP_EPI_IN_M : FBC); ++b) epilogue(b);
        c.m_done();                                             // M -> L of the next chunk; own refill pieces landed
        const bool nxt_same = cb + 1 < NFB_OUT / FBC;           // the next chunk has this chunk's shape
        if (PNR_PP_EARLY && nxt_same) CH::first_frags(c.next_frag_addr(), A);
        c.refill_begin();
#pragma unroll
        for (int b = 0; b < FBC; ++b) {
            c.refill_one();                                     // one LDS-DMA piece, then a block's pack / ReLU in its shadow
            if (b >= PNR_PP_EPI_IN_M) epilogue(b);
            if (save) store_slots(save, NFB_OUT * 32, srow, cb * FBC + b, c.hi, &out[(cb * FBC + b) * 8]);
            side(cb * FBC + b);                                 // a constant once the loops are unrolled
        }
        if (PNR_PP_EARLY_BIAS && nxt_same) CH::bias_issue(c.next_bias_addr(), q);   // accumulators free: next chunk's bias, asynchronous
        c.refill_rest();
        side(-1 - cb);                                          // end of chunk cb's refill: behind every piece of this L phase
        c.advance();
    }
}

// FUSE: the block is not stored; it is reduced into the tile's compositing record (pnr_mlp_fuse.h), all in the L phase
template <bool TRAIN, bool FUSE, class CTX, int NA, int NB>
__device__ __forceinline__ void pp_layer_out(CTX& c, u32x4 (&A)[CTX::P], const uint32_t (&inA)[NA],
                                             const uint32_t (&inB)[NB > 0 ? NB : 1], int n_out, int ch_base, int samp,
                                             FuseState* st = nullptr)
{
    using CH = PPChunk<1, NA, NB>;
    const int nfb = (n_out + 31) >> 5;
#pragma unroll 1
    for (int fb = 0; fb < nfb; ++fb) {
        f32x16 acc[1];
        const bool logits_t = FUSE && PNR_FUSE_TRANSPOSED && ch_base != 0;      // wave-uniform
        if (logits_t) {
            CH::prologue_swapped(c.frag_addr(), c.bias_addr() - c.hi * 16, c.lane, A, acc);
            CH::template mma<true>(c.frag_addr(), A, inA, inB, acc, [&](auto... k) { if constexpr (sizeof...(k) == 0) { c.stamp(6); c.barrier(); c.stamp(2); } else c.stamp(k...); });
        } else {
            CH::prologue(c.frag_addr(), c.bias_addr(), A, acc);
            CH::mma(c.frag_addr(), A, inA, inB, acc, [&](auto... k) { if constexpr (sizeof...(k) == 0) { c.stamp(6); c.barrier(); c.stamp(2); } else c.stamp(k...); });
        }
        c.m_done();          // its vmcnt(0) precedes the stores below: it never waits for an HBM write issued in this phase
        c.refill_begin();
        c.refill_one();
#if PNR_PP_STORES_LAST
        static_assert(!FUSE, "PNR_PP_STORES_LAST: two-kernel path only (the fused kernel has no raw stores)");
        // Every piece first, then the stores, and the NUMBER of store instructions this wave issues is handed to the next
        // m_done(): VMEM operations of a wave complete in order, so `s_waitcnt vmcnt(#stores)` there covers the pieces
        // without waiting for the stores' HBM write acknowledgements (which take longer than two phases).
        c.refill_rest();
        if (!(PNR_PP_ABL & 4)) {
            store_raw_block(c.a, samp, c.hi, fb, n_out, ch_base, acc[0]);
            int cnt = 0;
            if (!TRAIN && __builtin_amdgcn_ballot_w64(samp >= 0) != 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) cnt += (fb * 32 + (r & 3) + 8 * (r >> 2) < n_out) ? 1 : 0;   // row of the hi = 0 half
            }
            c.pending_stores = cnt;
        }
#else
        if constexpr (FUSE) {
            if (ch_base == 0) fuse_rgbs(c.a, *st, c.hi, c.lane & 31, acc[0]);
            else if (PNR_FUSE_TRANSPOSED) fuse_logits_t(*st, c.hi, c.lane, fb, n_out, PNR_FUSE_REC_LOGITS + (ch_base - 4), acc[0]);
            else fuse_logits(*st, c.hi, c.lane & 31, fb, n_out, PNR_FUSE_REC_LOGITS + (ch_base - 4), acc[0]);
        } else if (!(PNR_PP_ABL & 4)) store_raw_block(c.a, samp, c.hi, fb, n_out, ch_base, acc[0]);
        c.refill_rest();
#endif
        c.advance();
    }
}

// Plan 1 (fused inference): the two logit layers as ONE chunk of NBS + NBI transposed 32-channel blocks (8 k-steps each:
// head_W = 128), block b < NBS from the semantic head's hidden activations, the others from the instance head's.  One
// M phase of 8 (NBS + NBI) MFMAs on NBS + NBI interleaved accumulator chains instead of NBS + NBI single-chain chunks of 8
// whose L phases (refill of a full-size chunk each, prologue, epilogue) nothing covered: 4400 + 2300 cycles per sample group
// in the per-chunk trace (profiles/r03/r03a) for 1536 cycles of MFMA.
template <int FB, int KS>
struct PPLogitsGeom {       // fragments of the chunk in consumption order i = ks * FB + b (block-major in the image)
    static constexpr int NF = FB * KS, BIAS_OFF = FB * KS * PNR_FRAG_BYTES;
    static constexpr int frag_off(int i) { return ((i % FB) * KS + (i / FB)) * PNR_FRAG_BYTES; }
};
template <int NBS, int NBI, bool SOFTMAX, class CTX>
__device__ __forceinline__ void pp_logits_merged(CTX& c, u32x4 (&A)[CTX::P], const uint32_t (&shs)[32], const uint32_t (&shi)[32],
                                                 FuseState& st)
{
    constexpr int P = CTX::P, FB = NBS + NBI, KS = 8;
    using GEO = PPLogitsGeom<FB, KS>;
    constexpr int NF = GEO::NF, BIAS_OFF = GEO::BIAS_OFF;
    const uint32_t fa = c.frag_addr();
    // L, last part: first fragments, bias of channel fb*32 + (lane & 31) into every register of block fb's accumulator
    pp_static_for<(P - 1 < NF ? P - 1 : NF)>([&](auto I) {
        constexpr int i = I;
        pp_lds_read<GEO::frag_off(i)>(A[i % P], fa);
    });
    f32x16 acc[FB];
    {
        float bj[FB];
        const uint32_t ba = c.bias_addr() - c.hi * 16 + (uint32_t)(c.lane & 31) * 4u;
        pp_static_for<FB>([&](auto B) {
            constexpr int b = B;
            pp_lds_read_b32<BIAS_OFF + b * 128>(bj[b], ba);
        });
#pragma unroll
        for (int b = 0; b < FB; ++b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bj[b]));
        if constexpr (SOFTMAX) {            // channels past the head's last: -inf (zero weights keep them there), exp() makes them 0
#pragma unroll
            for (int b = 0; b < FB; ++b) {
                const int ch = (b < NBS ? b : b - NBS) * 32 + (c.lane & 31);
                if (ch >= (b < NBS ? c.a.n_sem : c.a.n_inst)) bj[b] = -INFINITY;
            }
        }
#pragma unroll
        for (int b = 0; b < FB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = bj[b];
        __builtin_amdgcn_sched_barrier(0);
    }
    c.stamp(6);
    c.barrier();                                                    // L -> M
    c.stamp(2);
#if PNR_PP_PRIO
    __builtin_amdgcn_s_setprio(PNR_PP_PRIO);
#endif
    pp_static_for<NF>([&](auto I) {
        constexpr int i = I;
        constexpr int ks = i / FB, b = i % FB;
        constexpr int younger = (NF - 1 - i) < (P - 2) ? (NF - 1 - i) : (P - 2);
        pp_wait<younger>(A[i % P]);
        const uint32_t* act = b < NBS ? &shs[4 * ks] : &shi[4 * ks];
        u32x4 av;
        av[0] = act[0]; av[1] = act[1]; av[2] = act[2]; av[3] = act[3];
        // operands swapped (PPChunk::mma<SWAP>): lane = output channel, register r = sample row(r, hi)
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, A[i % P]), acc[b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (i + P - 1 < NF) {
            pp_lds_read<GEO::frag_off(i + P - 1)>(A[(i + P - 1) % P], fa);
            __builtin_amdgcn_sched_barrier(0);
        }
    });
#if PNR_PP_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    c.m_done();
    c.refill_begin();
    if constexpr (SOFTMAX) {            // a head's blocks together: the softmax runs over all of its channels
        c.refill_one();
        fuse_softmax_t<NBS>(st, c.hi, c.lane, c.a.n_sem, PNR_FUSE_REC_LOGITS, &acc[0]);
        if constexpr (NBI > 0) {
            c.refill_one();
            fuse_softmax_t<NBI>(st, c.hi, c.lane, c.a.n_inst, PNR_FUSE_REC_LOGITS + c.a.n_sem, &acc[NBS]);
        }
    } else
    pp_static_for<FB>([&](auto B) {
        constexpr int b = B;
        c.refill_one();
        if constexpr (b < NBS) fuse_logits_t(st, c.hi, c.lane, b, c.a.n_sem, PNR_FUSE_REC_LOGITS, acc[b]);
        else fuse_logits_t(st, c.hi, c.lane, b - NBS, c.a.n_inst, PNR_FUSE_REC_LOGITS + c.a.n_sem, acc[b]);
    });
    c.refill_rest();
    c.advance();
}

// TAIL (FUSE only): 0 = classic plan (runtime loops over the logit blocks); 4 NBS + NBI (+ 16: softmax compositing of the two
// fields, pnr_mlp_fuse.h fuse_softmax_t) = plan 1 with NBS semantic and NBI
// instance logit blocks merged into one chunk (pnr_mlp_plan.h)
template <int W, bool TRAIN, bool FUSE = false, int TAIL = 0>
__global__ __launch_bounds__(512, 2) void k_mlp_pp(const MlpArgs a)
{
    static_assert(TAIL == 0 || (FUSE && !TRAIN && W == 256), "plan 1 is the fused inference kernel's");
    constexpr int WAVES = 8;
    using CTX = CtxPP<WAVES>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NFB = W / 32, HFB = W / 64;
    constexpr int HR = NFB * 8, GR = HFB * 8, GXR = 16, GDR = 8;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    CTX c{a, smem, (int)(threadIdx.x & 63), wave, (int)((threadIdx.x & 63) >> 5), wave >= WAVES / 2 ? 1 : 0, 0, 0, 0u, 0u, {0, 0}};
#if PNR_PP_ABL & 16
    c.abl_sink = u32x4{0, 0, 0, 0};
#endif
#if PNR_TRACE
    c.tr = reinterpret_cast<unsigned long long*>(smem + 3 * a.slot_bytes) + c.wave * PNR_TRACE_CHUNKS * PNR_TRACE_STAMPS;
    c.titer = 0;
#endif
    const int n = c.lane & 31;
    c.start();
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (a.clk) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }

    uint32_t dummy[1] = {0};
    u32x4 A[CTX::P];
    struct SampleIn { float4 o4, d4; float zz, zn; bool last; };
    auto fetch = [&](int grp) {
        const int s = (grp * WAVES + c.wave) * 32 + n;
        const int sl = s < a.S ? s : a.S - 1;
        const int ray = pnr_div_magic(sl, a.n_magic, a.n_shift);     // sl / N without the ~40-instruction integer division
        SampleIn in;
        in.o4 = *reinterpret_cast<const float4*>(a.rays + (int64_t)ray * 8);
        in.d4 = *reinterpret_cast<const float4*>(a.rays + (int64_t)ray * 8 + 4);
        in.zz = a.z[sl];
        in.zn = FUSE ? a.z[sl + 1 < a.S ? sl + 1 : sl] : 0.0f;     // z of the next sample (used inside a ray only)
        in.last = FUSE ? (sl - ray * a.N + 1 == a.N) : false;      // the ray's last sample: its interval is 1e10
        return in;
    };
    SampleIn nextin = fetch(blockIdx.x < a.n_groups ? blockIdx.x : 0);
    FuseState fst;
    // Staged encodings (see embed_sincos): gamma(d) of THIS sample group and gamma(x) of the NEXT one are computed piece by
    // piece in the L phases of the feature layer -- the last full-size layer of a group, whose M phases leave the VALU ~80 %
    // idle -- and the next group's inputs are requested behind the refill pieces of its first chunk.  At a group's top and in
    // front of the views layer, where the partner wave group has little or nothing to multiply and round 2 computed them in
    // one lump (per-chunk trace: ~4500 + ~1400 cycles per sample group), a few moves are left.  ex is free for the next
    // group's values from the end of the trunk on (layer 0 and the skip layer are its only readers).
    uint32_t ex[GXR], ed[GDR];
    float dn_next;                         // |d| of the next group's sample
    auto points = [&](const SampleIn& in, float& px, float& py, float& pz, float& nrm) {
        const float dx = in.o4.w, dy = in.d4.x, dz = in.d4.y;
        // pts = o + d*z: separate multiply and add, as the sampler's pnr_points does
        px = __fadd_rn(in.o4.x, __fmul_rn(dx, in.zz));
        py = __fadd_rn(in.o4.y, __fmul_rn(dy, in.zz));
        pz = __fadd_rn(in.o4.z, __fmul_rn(dz, in.zz));
        // k_composite's |d|: sqrtf((dx*dx + dy*dy) + dz*dz), contraction off
        nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    };
    {
        float px, py, pz;
        points(nextin, px, py, pz, dn_next);
        embed_lane<PNR_PREC_BF16, 5, 32, GXR>(px, py, pz, c.hi, ex);
    }

    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        const int s0 = (grp * WAVES + c.wave) * 32 + n;
        const int samp = s0 < a.S ? s0 : -1, srow = s0;
        const float dx = nextin.o4.w, dy = nextin.d4.x, dz = nextin.d4.y, dn = dn_next;
        if constexpr (TRAIN) store_ex(a.acts + a.acts_off[0], srow, c.hi, ex);
        if constexpr (FUSE) {
            fst.zz = nextin.zz; fst.zn = nextin.zn; fst.dn = dn; fst.samp = samp; fst.last = nextin.last;
            fst.rec = a.rec + (int64_t)(grp * WAVES + c.wave) * a.rec_floats;
        }
        auto sv = [&](int idx) -> uint16_t* { return TRAIN ? a.acts + a.acts_off[idx] : nullptr; };

        auto gv = [&](int idx, auto& regs) {
            if constexpr (TRAIN) save_gates(a.acts + a.gate_off[idx], srow, c.hi, regs);
        };
        uint32_t cur[HR], nxt[HR];
        pp_layer_regs<CTX, PNR_L_TRUNK0, GXR, 0, NFB, MODE_RELU, HR, (TAIL > 0 && PNR_PLAN1_TRUNK0_MERGE ? NFB : 0)>(c, A, ex, dummy, cur, sv(2), srow);
        gv(2, cur);
        auto trunk = [&](int l, const uint32_t (&in)[HR], uint32_t (&out)[HR]) {
            if (l - 1 == a.skip)
                pp_layer_regs<CTX, PNR_L_TRUNK, GXR, HR, NFB, MODE_RELU, HR>(c, A, ex, in, out, sv(2 + l), srow);
            else
                pp_layer_regs<CTX, PNR_L_TRUNK, HR, 0, NFB, MODE_RELU, HR>(c, A, in, dummy, out, sv(2 + l), srow);
            gv(2 + l, out);
        };
#if PNR_PP_UNROLL2
        // two layers per trip, cur -> nxt -> cur: no 64-register hand-over copy per layer (it sat in the L phase of every
        // layer's first chunk); one copy per sample group remains when D-1 is odd
#pragma unroll 1
        for (int l = 1; l < a.D; l += 2) {
            trunk(l, cur, nxt);
            if (l + 1 < a.D) trunk(l + 1, nxt, cur);
            else {
#pragma unroll
                for (int i = 0; i < HR; ++i) cur[i] = nxt[i];
            }
        }
#else
#pragma unroll 1
        for (int l = 1; l < a.D; ++l) {
            trunk(l, cur, nxt);
#pragma unroll
            for (int i = 0; i < HR; ++i) cur[i] = nxt[i];       // (32 v_pk_mov_b32 instead of these 64 v_mov_b32: +-0 measured)
        }
#endif
        // side work of the feature layer: eight stages over its NFB slots -- gamma(d) of this group (D0..D3), then gamma(x) of
        // the next one (X0..X3), whose inputs are requested behind chunk 0's refill pieces (two chunks ahead of X0 at W = 256)
        const int g2 = grp + (int)gridDim.x < a.n_groups ? grp + (int)gridDim.x : grp;
        EmbedSC esc;
        float q0, q1, q2;                  // the 3-vector being encoded: d / |d|, then the next group's point
        auto stage = [&](int k) {
            switch (k) {
            case 0: q0 = dx / dn; q1 = dy / dn; q2 = dz / dn; break;
            case 1: embed_sincos<2>(q0, 0, c.hi, esc); embed_sincos<2>(q1, 1, c.hi, esc); break;
            case 2: embed_sincos<2>(q2, 2, c.hi, esc); ed[0] = embed_pack_xyz(q0, q1, q2, c.hi); embed_pack_band(esc, &ed[1]); break;
            case 3: embed_next_band(esc); embed_pack_band(esc, &ed[4]); ed[7] = 0u; break;
            case 4: points(nextin, q0, q1, q2, dn_next); ex[0] = embed_pack_xyz(q0, q1, q2, c.hi); embed_sincos<5>(q0, 0, c.hi, esc); break;
            case 5: embed_sincos<5>(q1, 1, c.hi, esc); embed_sincos<5>(q2, 2, c.hi, esc); embed_pack_band(esc, &ex[1]); break;
            case 6: embed_next_band(esc); embed_pack_band(esc, &ex[4]); embed_next_band(esc); embed_pack_band(esc, &ex[7]); break;
            default: embed_next_band(esc); embed_pack_band(esc, &ex[10]); embed_next_band(esc); embed_pack_band(esc, &ex[13]); break;
            }
        };
        auto side = [&](int slot) {
            if (slot == -1) {
                // the four loads of fetch(): vector-memory operations of a wave complete in order, so the next m_done() may
                // wait for "all but the youngest four" -- the refill pieces -- instead of an HBM round trip
                // (the count is the number of vector-memory instructions hipcc emits for fetch(); the markers let
                // tests/test_asm_lint.py verify it, and that no LDS-DMA piece sits between them, on the compiled assembly)
                __builtin_amdgcn_sched_barrier(0);     // the loads must stay the YOUNGEST operations: nothing may move across
                asm volatile("; PNR_FETCH_BEGIN" ::: "memory");
                nextin = fetch(g2);
                if constexpr (FUSE) asm volatile("; PNR_FETCH_END 4" ::: "memory");
                else asm volatile("; PNR_FETCH_END 3" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                c.pending_stores = FUSE ? 4 : 3;
            }
            if (slot < 0) return;
            constexpr int PER = 8 / NFB;   // stages per slot: 1 (W = 256), 2 (W = 128)
#pragma unroll
            for (int j = 0; j < PER; ++j) stage(slot * PER + j);
        };
        pp_layer_regs<CTX, PNR_L_FEATURE, HR, 0, NFB, MODE_LINEAR, HR>(c, A, cur, dummy, nxt, sv(2 + a.D), srow, side);
        if constexpr (TRAIN) store_slots(a.acts + a.acts_off[1], 32, srow, 0, c.hi, ed);   // ED: 32 slots
        uint32_t g[GR];
        pp_layer_regs<CTX, PNR_L_VIEWS, HR, GDR, HFB, MODE_RELU, GR>(c, A, nxt, ed, g, sv(3 + a.D), srow);
        gv(3 + a.D, g);
        pp_layer_out<TRAIN, FUSE, CTX, GR, HR>(c, A, g, cur, 4, 0, samp, &fst);
        // the heads read the trunk output h = cur, or -- head_tap 1 -- the feature_linear output: h is dead once sigma is out, so
        // the feature registers take its place (64 moves per sample group, only in that mode)
        if (a.head_tap && (a.n_sem || a.n_inst)) {
#pragma unroll
            for (int i = 0; i < HR; ++i) cur[i] = nxt[i];
        }
        if constexpr (TAIL > 0) {
            // plan 1: both head hidden layers, then every logit block in one chunk
            constexpr int NBS = (TAIL >> 2) & 3, NBI = TAIL & 3;
            uint32_t shs[GR], shi[GR];
            pp_layer_regs<CTX, PNR_L_SEM0, HR, 0, HFB, MODE_RELU, GR>(c, A, cur, dummy, shs, nullptr, srow);
            if constexpr (NBI > 0) pp_layer_regs<CTX, PNR_L_INST0, HR, 0, HFB, MODE_RELU, GR>(c, A, cur, dummy, shi, nullptr, srow);
            else {
#pragma unroll
                for (int i = 0; i < GR; ++i) shi[i] = 0;
            }
            pp_logits_merged<NBS, NBI, (TAIL >> 4) != 0>(c, A, shs, shi, fst);
        } else if (a.head_depth == 1) {     // one Linear per head, straight from the tap
            if (a.n_sem) pp_layer_out<TRAIN, FUSE, CTX, HR, 0>(c, A, cur, dummy, a.n_sem, 4, samp, &fst);
            if (a.n_inst) pp_layer_out<TRAIN, FUSE, CTX, HR, 0>(c, A, cur, dummy, a.n_inst, 4 + a.n_sem, samp, &fst);
        } else {
        // panoptic heads, after the appearance branch (plan order)
        if (a.n_sem) {
            uint32_t sh[GR];
            pp_layer_regs<CTX, PNR_L_SEM0, HR, 0, HFB, MODE_RELU, GR>(c, A, cur, dummy, sh, sv(4 + a.D), srow);
            gv(4 + a.D, sh);
            pp_layer_out<TRAIN, FUSE, CTX, GR, 0>(c, A, sh, dummy, a.n_sem, 4, samp, &fst);
        }
        if (a.n_inst) {
            uint32_t sh[GR];
            pp_layer_regs<CTX, PNR_L_INST0, HR, 0, HFB, MODE_RELU, GR>(c, A, cur, dummy, sh, sv(5 + a.D), srow);
            gv(5 + a.D, sh);
            pp_layer_out<TRAIN, FUSE, CTX, GR, 0>(c, A, sh, dummy, a.n_inst, 4 + a.n_sem, samp, &fst);
        }
        }
#if PNR_TRACE
        ++c.titer;
#endif
    }
    c.end();
    if (a.clk && blockIdx.x == 0 && threadIdx.x == 0) {
        a.clk[0] = __builtin_amdgcn_s_memtime() - clk_c0;
        a.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#if PNR_TRACE
    __syncthreads();
    if (blockIdx.x == PNR_TRACE_WG && a.trace) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(smem + 3 * a.slot_bytes);
        for (int i = threadIdx.x; i < WAVES * PNR_TRACE_CHUNKS * PNR_TRACE_STAMPS; i += blockDim.x) a.trace[i] = src[i];
    }
#endif
}

template <int W, bool TRAIN, bool FUSE = false, int TAIL = 0>
static int launch_mlp_pp(const MlpArgs& a0, hipStream_t stream)
{
    MlpArgs a = a0;
    const int lds_bytes = 3 * a.slot_bytes + (PNR_TRACE ? 8 * PNR_TRACE_CHUNKS * PNR_TRACE_STAMPS * 8 : 0);
    PNR_REQUIRE(lds_bytes <= 163840, "pnr_mlp_forward: three weight slots of %d bytes exceed the 160 KiB LDS", a.slot_bytes);
    PNR_REQUIRE(a.n_chunks >= 4, "pnr_mlp_forward: network too small for the weight stream");
    a.n_groups = (a.S + 255) / 256;
    auto kern = k_mlp_pp<W, TRAIN, FUSE, TAIL>;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        PNR_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        attr_set = true;
    }
    const int cap = pnr_cu_count();           // one 8-wave workgroup per CU (2 x 256 registers per SIMD)
    const int grid = a.n_groups < cap ? a.n_groups : cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds_bytes, stream, a);
    PNR_CHECK_LAUNCH("pnr_mlp_forward");
    return PNR_OK;
}

// ------------------------------------------------------------------------------- launcher
template <int PREC, int W, int TILES, int WAVES, int MINW, bool TRAIN = false>
static int launch_mlp(const MlpArgs& a0, hipStream_t stream)
{
    MlpArgs a = a0;
    const int lds_bytes = 2 * a.slot_bytes + (PNR_TRACE ? WAVES * PNR_TRACE_CHUNKS * PNR_TRACE_STAMPS * 8 : 0);
    PNR_REQUIRE(lds_bytes <= 163840, "pnr_mlp_forward: weight double buffer of %d bytes exceeds the 160 KiB LDS", lds_bytes);
    PNR_REQUIRE(a.n_chunks >= 3, "pnr_mlp_forward: network too small for the weight stream");
    const int per_group = 32 * TILES * WAVES;
    a.n_groups = (a.S + per_group - 1) / per_group;
    auto kern = k_mlp_fused<PREC, W, TILES, WAVES, MINW, TRAIN>;
    static thread_local int wg_per_cu = 0;
    if (wg_per_cu == 0) {
        PNR_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        int nb = 0;
        PNR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 64 * WAVES, lds_bytes));
        wg_per_cu = nb < 1 ? 1 : (nb > 4 ? 4 : nb);
    }
    // persistent grid: every resident workgroup slot of the device's CUs, grid-stride over sample groups
    const int cap = pnr_cu_count() * wg_per_cu;
    const int grid = a.n_groups < cap ? a.n_groups : cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds_bytes, stream, a);
    PNR_CHECK_LAUNCH("pnr_mlp_forward");
    return PNR_OK;
}

// pnr_mlp_desc::schedule (tests and A/B tools): 0 = ping-pong for inference / lock-step for the training forward, 1 = lock-step
// everywhere, 2 = ping-pong everywhere.  A descriptor field: the library keeps no mutable process-global (include/pnr.h).

// diagnostics: pnr_mlp_desc.clk_probe = where the kernels of THIS launch leave {shader cycles, 100 MHz ticks} of workgroup 0's first
// wave (their ratio = the mean shader clock during the launch); 0 = off.  A descriptor field: libpnr_bench.so sets it on its own copy
// of the descriptor around the launches it times -- the library keeps no mutable state.
static inline unsigned long long* clk_probe_of(const pnr_mlp_desc* d)
{
    return (unsigned long long*)(uintptr_t)(((uint64_t)(uint32_t)d->clk_probe[1] << 32) | (uint64_t)(uint32_t)d->clk_probe[0]);
}

static int mlp_forward_impl(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z,
                            int64_t n_rays, int n_samples, float* raw, int64_t raw_stride_s, int64_t raw_stride_c,
                            void* acts, void* stream)
{
    int rc = pnr_mlp_validate(desc);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(n_rays >= 0 && n_samples >= 1, "pnr_mlp_forward: bad size");
    if (n_rays == 0) return PNR_OK;
    PNR_REQUIRE(packed && rays && z && raw, "pnr_mlp_forward: null pointer");
    PNR_REQUIRE(n_rays * (int64_t)n_samples < ((int64_t)1 << 31) - 4096, "pnr_mlp_forward: R*N=%lld exceeds 2^31",
                (long long)(n_rays * n_samples));
    PNR_REQUIRE((((uintptr_t)rays) & 15) == 0 && (((uintptr_t)packed) & 15) == 0 && (((uintptr_t)acts) & 15) == 0,
                "pnr_mlp_forward: rays / packed / acts must be 16-byte aligned");
    PNR_REQUIRE(!acts || desc->precision == PNR_PREC_BF16, "pnr_mlp_forward_train: the training path is bf16 only");
    PNR_REQUIRE(desc->plan == 0, "pnr_mlp_forward: plan=%d images are for pnr_mlp_forward_composite only", desc->plan);
    PnrPlan plan;
    pnr_build_plan(*desc, plan);
    MlpArgs a;
    memset(&a, 0, sizeof(a));
    a.data = (const uint8_t*)packed + plan.data_off;
    a.table = (const pnr_chunk_entry*)((const uint8_t*)packed + plan.table_off);
    a.n_chunks = (int)plan.chunks.size();
    a.slot_bytes = plan.max_chunk_frags * PNR_FRAG_BYTES;
    a.rays = rays; a.z = z; a.S = (int)(n_rays * n_samples); a.N = n_samples; a.n_groups = 0;
    pnr_set_div_magic(n_samples, a.n_magic, a.n_shift);
    a.raw = raw; a.ss = raw_stride_s; a.sc = raw_stride_c;
    a.D = desc->D; a.skip = desc->skip; a.n_sem = desc->n_sem; a.n_inst = desc->n_inst;
    a.head_tap = desc->head_tap; a.head_depth = desc->head_depth == 1 ? 1 : 2;
    a.acts = (uint16_t*)acts;
#if PNR_TRACE
    if (const char* e = getenv("PNR_TRACE_PTR")) a.trace = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
    a.clk = clk_probe_of(desc);
    if (acts) pnr_train_layout(*desc, a.S, a.acts_off, a.dys_off, a.gate_off);
    hipStream_t st = (hipStream_t)stream;
    // bf16: 8 waves x 1 tile, registers capped at 256 (2 waves per SIMD, one workgroup per CU);
    // fp32 parity mode: 4 waves x 1 tile, one wave per SIMD (its activations need ~300 registers)
    if (desc->precision == PNR_PREC_BF16) {
        // ping-pong form (pnr_mlp_pp.h): variant 1 = inference launches, 2 = the training forward as well
        if (desc->schedule != 1 && !acts)
            return desc->W == 256 ? launch_mlp_pp<256, false>(a, st) : launch_mlp_pp<128, false>(a, st);
        if (desc->schedule == 2 && acts)
            return desc->W == 256 ? launch_mlp_pp<256, true>(a, st) : launch_mlp_pp<128, true>(a, st);
#if PNR_TRAIN_TILES_EXPERIMENT
        // round-6 experiment (tools/build_ab.sh tt:"-DPNR_TRAIN_TILES_EXPERIMENT=1"; profiles/r06b): the training forward with ONE wave per
        // SIMD and 2 / 3 sample tiles per wave.  3: a weight pass feeds 384 samples instead of 256 (timing only: S % 384 == 0)
        if (acts && desc->schedule == 3 && desc->W == 256) return launch_mlp<PNR_PREC_BF16, 256, 2, 4, 1, true>(a, st);
        if (acts && desc->schedule == 4 && desc->W == 256) {
            PNR_REQUIRE(a.S % 384 == 0, "schedule 4 (experiment): S must be a multiple of 384");
            return launch_mlp<PNR_PREC_BF16, 256, 3, 4, 1, true>(a, st);
        }
#endif
        if (acts)
            return desc->W == 256 ? launch_mlp<PNR_PREC_BF16, 256, 1, 8, 2, true>(a, st)
                                  : launch_mlp<PNR_PREC_BF16, 128, 1, 8, 2, true>(a, st);
        return desc->W == 256 ? launch_mlp<PNR_PREC_BF16, 256, 1, 8, 2>(a, st) : launch_mlp<PNR_PREC_BF16, 128, 1, 8, 2>(a, st);
    }
    return desc->W == 256 ? launch_mlp<PNR_PREC_FP32, 256, 1, 4, 1>(a, st) : launch_mlp<PNR_PREC_FP32, 128, 1, 4, 1>(a, st);
}

PNR_EXPORT int pnr_mlp_forward(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z,
                               int64_t n_rays, int n_samples, float* raw, int64_t raw_stride_s,
                               int64_t raw_stride_c, void* stream)
{
    return mlp_forward_impl(desc, packed, rays, z, n_rays, n_samples, raw, raw_stride_s, raw_stride_c, nullptr, stream);
}

#define PNR_RAY_AUX_BYTES 128
// A/B builds only (tools/build_tt_variant.sh save PNR_TT_SAVE=1, run with PNR_TT_SAVE_PROTO=1): the training-forward prototype of the
// two-tile kernel stores every packed activation block to a scratch region behind the per-ray table (profiles/r06/r06p)
#define PNR_TT_SAVE_REGION (352 * 1024)
static bool pnr_tt_save_proto() { static const bool on = getenv("PNR_TT_SAVE_PROTO") != nullptr; return on; }
int pnr_composite_combine_launch(const float* rec, int rec_floats, const float4* ps, const float* z, const int32_t* label_sem,
                                 const int32_t* label_inst, int64_t R, int N, int C, int K, int white_bkgd, float* rgb, float* depth,
                                 float* acc, float* weights, float* sem, float* inst, float* fix_sem, float* fix_inst, hipStream_t st);

// workspace of pnr_mlp_forward_composite: one record per 32-sample tile (padded to whole 256-sample groups) and one
// (lw, r, g, b) quadruple per sample (want_weights is accepted for ABI stability and no longer changes the size)
PNR_EXPORT int64_t pnr_mlp_forward_composite_workspace_bytes(const pnr_mlp_desc* desc, int64_t n_rays, int n_samples, int want_weights)
{
    (void)want_weights;
    if (pnr_mlp_validate(desc) != PNR_OK || n_rays < 0 || n_samples < 32 || (n_samples & 31)) return -1;
    const int64_t S = n_rays * n_samples, tiles = (S + 255) / 256 * 8;
    return tiles * pnr_fuse_record_floats(desc->n_sem, desc->n_inst) * 4 + S * 16 + 256 + PNR_RAY_AUX_BYTES + n_rays * PNR_RAY_AUX_BYTES +
           (pnr_tt_save_proto() ? tiles / 8 * 4 * PNR_TT_SAVE_REGION + 4096 : 0);
}

// What the two-tile kernel needs of a RAY rather than of a sample: |d| and gamma(d / |d|), in the registers embed_lane hands a lane
// of either half-wave.  k_mlp_tt used to compute them per sample (188 of a tile's ~530 encoder instructions; N samples of a ray
// share them); this kernel computes them once per ray with the SAME calls (bit-identical by construction), 128 bytes per ray:
// [half-wave]{8 packed registers, |d|, 7 unused}.  ~1 us per 100k rays in front of a launch of milliseconds.
__global__ void __launch_bounds__(256) k_ray_aux(const float* __restrict__ rays, int64_t R, uint32_t* __restrict__ aux)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * R) return;
    const int64_t ray = i >> 1;
    const int hi = (int)(i & 1);
    const float dx = rays[ray * 8 + 3], dy = rays[ray * 8 + 4], dz = rays[ray * 8 + 5];
    // k_composite's |d|: sqrtf((dx*dx + dy*dy) + dz*dz), contraction off
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
    uint32_t ed[8];
    embed_lane<PNR_PREC_BF16, 2, 16, 8>(dx / nrm, dy / nrm, dz / nrm, hi, ed);
    uint4* o = reinterpret_cast<uint4*>(aux + i * 16);
    o[0] = make_uint4(ed[0], ed[1], ed[2], ed[3]);
    o[1] = make_uint4(ed[4], ed[5], ed[6], ed[7]);
    o[2] = make_uint4(__float_as_uint(nrm), 0u, 0u, 0u);
}

// the fused MLP launch alone (records + optional local weights into `workspace`); `a` is returned for the combine step
static int fused_mlp_launch(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z, int64_t n_rays,
                            int n_samples, void* workspace, void* stream, MlpArgs& a)
{
    int rc = pnr_mlp_validate(desc);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(desc->precision == PNR_PREC_BF16, "pnr_mlp_forward_composite: bf16 only");
    PNR_REQUIRE(n_rays >= 1 && n_samples >= 32 && n_samples <= 256 && (n_samples & 31) == 0,
                "pnr_mlp_forward_composite: n_samples=%d must be a multiple of 32 in [32,256]", n_samples);
    PNR_REQUIRE(desc->n_sem + desc->n_inst <= 128, "pnr_mlp_forward_composite: n_sem + n_inst <= 128");
    PNR_REQUIRE(packed && rays && z && workspace, "pnr_mlp_forward_composite: null pointer");
    PNR_REQUIRE(n_rays * (int64_t)n_samples < ((int64_t)1 << 31) - 4096, "pnr_mlp_forward_composite: R*N exceeds 2^31");
    PNR_REQUIRE((((uintptr_t)rays | (uintptr_t)packed | (uintptr_t)workspace) & 15) == 0,
                "pnr_mlp_forward_composite: rays / packed / workspace must be 16-byte aligned");
    PnrPlan plan;
    pnr_build_plan(*desc, plan);
    memset(&a, 0, sizeof(a));
    a.data = (const uint8_t*)packed + plan.data_off;
    a.table = (const pnr_chunk_entry*)((const uint8_t*)packed + plan.table_off);
    a.n_chunks = (int)plan.chunks.size();
    a.slot_bytes = plan.max_chunk_frags * PNR_FRAG_BYTES;
    a.rays = rays; a.z = z; a.S = (int)(n_rays * n_samples); a.N = n_samples;
    pnr_set_div_magic(n_samples, a.n_magic, a.n_shift);
    a.D = desc->D; a.skip = desc->skip; a.n_sem = desc->n_sem; a.n_inst = desc->n_inst;
    a.head_tap = desc->head_tap; a.head_depth = desc->head_depth == 1 ? 1 : 2;
    a.rec_floats = pnr_fuse_record_floats(desc->n_sem, desc->n_inst);
    a.rec = (float*)workspace;
    const int64_t tiles = ((int64_t)a.S + 255) / 256 * 8;
    a.ps = (float4*)(a.rec + tiles * a.rec_floats);           // rec_floats % 4 == 0: 16-byte aligned
    a.clk = clk_probe_of(desc);
#if PNR_TRACE
    if (const char* e = getenv("PNR_TRACE_PTR")) a.trace = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
    hipStream_t st = (hipStream_t)stream;
    const bool softmax = (desc->flags & PNR_MLP_SOFTMAX) && desc->n_sem + desc->n_inst > 0;
    if (desc->plan == 2) {      // the two-tile assembly kernel (csrc/asm/gen_mlp_tt.py): same records, bit for bit
        PnrTTArgs t;
        memset(&t, 0, sizeof(t));
        t.image = a.data; t.rays = rays; t.z = z; t.S = a.S; t.N = a.N; t.n_magic = a.n_magic; t.n_shift = a.n_shift;
        t.rec = a.rec; t.rec_floats = a.rec_floats; t.ps = a.ps; t.n_sem = a.n_sem; t.n_inst = a.n_inst; t.clk = a.clk;
        // the per-ray table behind the quadruples (and their 256 bytes of slack), on a 128-byte line of its own per ray
        PNR_REQUIRE(n_rays < ((int64_t)1 << 24), "pnr_mlp_forward_composite: the two-tile kernel addresses 2^24 rays per launch");
        uint32_t* aux = (uint32_t*)(((uintptr_t)((uint8_t*)a.ps + (int64_t)a.S * 16 + 256) + PNR_RAY_AUX_BYTES - 1)
                                    & ~(uintptr_t)(PNR_RAY_AUX_BYTES - 1));
        k_ray_aux<<<dim3((unsigned)((2 * n_rays + 255) / 256)), dim3(256), 0, st>>>(rays, n_rays, aux);
        t.aux = aux;
        t.n_wg = (desc->flags >> 16) & 0x1FF;
        if (pnr_tt_save_proto()) t.save = (void*)(((uintptr_t)((uint8_t*)aux + n_rays * PNR_RAY_AUX_BYTES) + 4095) & ~(uintptr_t)4095);
        const bool trace = (desc->flags & 0xFF00) == PNR_MLP_TRACE;     // + (a << 4), a in 1..7: the timing-only ablation a
        return pnr_mlp_tt_launch(t, (desc->n_sem + 31) / 32, (desc->n_inst + 31) / 32, a.head_depth, a.head_tap, softmax, st, trace,
                                 trace ? ((desc->flags >> 4) & 7) : 0);
    }
    if (desc->plan == 1) {
        const int nbs = (desc->n_sem + 31) / 32, nbi = (desc->n_inst + 31) / 32;
        switch (4 * nbs + nbi + (softmax ? 16 : 0)) {
        case 4: return launch_mlp_pp<256, false, true, 4>(a, st);
        case 5: return launch_mlp_pp<256, false, true, 5>(a, st);
        case 8: return launch_mlp_pp<256, false, true, 8>(a, st);
        case 9: return launch_mlp_pp<256, false, true, 9>(a, st);
        case 20: return launch_mlp_pp<256, false, true, 20>(a, st);
        case 21: return launch_mlp_pp<256, false, true, 21>(a, st);
        case 24: return launch_mlp_pp<256, false, true, 24>(a, st);
        case 25: return launch_mlp_pp<256, false, true, 25>(a, st);
        default: PNR_REQUIRE(false, "pnr_mlp_forward_composite: no plan-1 kernel for %d + %d logit blocks", nbs, nbi);
        }
    }
    PNR_REQUIRE(!softmax, "pnr_mlp_forward_composite: softmax compositing needs the plan-1 image (pnr_mlp_fused_plan >= 1, desc.plan = 1)");
    return desc->W == 256 ? launch_mlp_pp<256, false, true>(a, st) : launch_mlp_pp<128, false, true>(a, st);
}

// First half of pnr_mlp_forward_composite: the fused MLP launch alone -- per-tile records and per-sample quadruples into `workspace`.
PNR_EXPORT int pnr_mlp_forward_tiles(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z, int64_t n_rays,
                                     int n_samples, void* workspace, void* stream)
{
    PNR_REQUIRE(n_rays >= 0, "pnr_mlp_forward_tiles: bad size");
    if (n_rays == 0) return PNR_OK;
    MlpArgs a;
    return fused_mlp_launch(desc, packed, rays, z, n_rays, n_samples, workspace, stream, a);
}

// Second half: every ray's maps from the workspace pnr_mlp_forward_tiles filled (same desc, n_rays, n_samples).
PNR_EXPORT int pnr_composite_combine(const pnr_mlp_desc* desc, const void* workspace, const float* z, int64_t n_rays, int n_samples,
                                     const int32_t* label_sem, const int32_t* label_inst, int white_bkgd, float* rgb, float* depth,
                                     float* acc, float* weights, float* sem, float* inst, float* fix_sem, float* fix_inst, void* stream)
{
    int rc = pnr_mlp_validate(desc);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(n_rays >= 0 && n_samples >= 32 && n_samples <= 256 && (n_samples & 31) == 0, "pnr_composite_combine: bad size");
    if (n_rays == 0) return PNR_OK;
    PNR_REQUIRE(workspace && z, "pnr_composite_combine: null pointer");
    PNR_REQUIRE((!fix_sem || label_sem) && (!fix_inst || label_inst), "pnr_composite_combine: fix_* outputs need their labels");
    const int rf = pnr_fuse_record_floats(desc->n_sem, desc->n_inst);
    const int64_t tiles = (n_rays * n_samples + 255) / 256 * 8;
    const float* rec = (const float*)workspace;
    return pnr_composite_combine_launch(rec, rf, (const float4*)(rec + tiles * rf), z, fix_sem ? label_sem : nullptr,
                                        fix_inst ? label_inst : nullptr, n_rays, n_samples, desc->n_sem, desc->n_inst, white_bkgd,
                                        rgb, depth, acc, weights, sem, inst, fix_sem, fix_inst, (hipStream_t)stream);
}

// a5 + a6 fused (inference, bf16, logits compositing, N % 32 == 0): the maps of every ray without the raw image round trip.
// label_sem / label_inst (R*N int32, -1 = none) and their fix_* outputs are optional; any output may be null.
PNR_EXPORT int pnr_mlp_forward_composite(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z,
                                         int64_t n_rays, int n_samples, const int32_t* label_sem, const int32_t* label_inst,
                                         int white_bkgd, float* rgb, float* depth, float* acc, float* weights, float* sem,
                                         float* inst, float* fix_sem, float* fix_inst, void* workspace, void* stream)
{
    PNR_REQUIRE(n_rays >= 0, "pnr_mlp_forward_composite: bad size");
    if (n_rays == 0) return PNR_OK;
    PNR_REQUIRE((!fix_sem || label_sem) && (!fix_inst || label_inst), "pnr_mlp_forward_composite: fix_* outputs need their labels");
    MlpArgs a;
    int rc = fused_mlp_launch(desc, packed, rays, z, n_rays, n_samples, workspace, stream, a);
    if (rc != PNR_OK) return rc;
    return pnr_composite_combine_launch(a.rec, a.rec_floats, a.ps, z, fix_sem ? label_sem : nullptr, fix_inst ? label_inst : nullptr,
                                        n_rays, n_samples, desc->n_sem, desc->n_inst, white_bkgd, rgb, depth, acc, weights, sem, inst,
                                        fix_sem, fix_inst, (hipStream_t)stream);
}

PNR_EXPORT int pnr_mlp_forward_train(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z,
                                     int64_t n_rays, int n_samples, float* raw, int64_t raw_stride_s,
                                     int64_t raw_stride_c, void* acts, void* stream)
{
    PNR_REQUIRE(acts || n_rays == 0, "pnr_mlp_forward_train: acts is null");
    return mlp_forward_impl(desc, packed, rays, z, n_rays, n_samples, raw, raw_stride_s, raw_stride_c, acts, stream);
}

PNR_EXPORT int pnr_mlp_train_layout(const pnr_mlp_desc* desc, int64_t n_samples, int64_t* acts_off_host,
                                    int64_t* dys_off_host)
{
    int rc = pnr_mlp_validate(desc);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(acts_off_host && dys_off_host && n_samples >= 0, "pnr_mlp_train_layout: bad arguments");
    int64_t a[24], d[24];
    pnr_train_layout(*desc, n_samples, a, d);
    memcpy(acts_off_host, a, sizeof(int64_t) * (size_t)(desc->D + 7));
    memcpy(dys_off_host, d, sizeof(int64_t) * (size_t)(desc->D + 8));
    return PNR_OK;
}
