// K3-bwd: data-gradient pass of the fused NeRF MLP, bf16 MFMA.
// Mirrors the forward kernel (pnr_mlp.hip): every backward layer is  dX^T = W^T * dY^T  with the
// transposed weights streamed through LDS as the MFMA A operand and the gradients register-resident
// as the B operand, so gradients flow from d_raw back to layer 1 without leaving the register file.
//   dY_l = dX_{l+1} (.) [X_{l+1} > 0]   (ReLU mask from the GATE BITS the forward saved: one bit per ReLU output)
// Every dY_l is also stored slot-ordered (MlpArgs::dys) for the weight-gradient GEMMs
// dW_l = dY_l^T X_l, which are plain (S-reduction) GEMMs done outside this kernel.
// Execution order / k-segments: pnr_mlp_plan.h (pnr_build_bwd_plan).  No gradient is needed for
// gamma(x), gamma(d) (the inputs are not learnable), so layer 0 and the view-direction columns have
// no data-gradient step.
//
// 8 waves x 1 tile per workgroup (two waves per SIMD), like the forward kernel.
#include <string.h>

#include "pnr_mlp_plan.h"
#include "pnr_mlp_core.h"

int pnr_mlp_validate(const pnr_mlp_desc* d);

// 8 waves (two per SIMD) measured 7 % faster than 4 (one per SIMD).  The concatenated d h input (136 B registers) leaves no
// room for the layer's 64 output registers at the 256-register cap (round 2: ~150 dwords per lane spilled to scratch); the
// d h layer therefore only STORES its output (it is stored for the weight gradients anyway) and the wave reads it back once
// the inputs are dead (layer_bwd<KEEP = false> + load_layer).
#ifndef PNR_BWD_SLOTS
#define PNR_BWD_SLOTS 2             /* LDS weight slots: the stream runs PNR_BWD_SLOTS - 1 chunks ahead (3: measured +-0) */
#endif
#ifndef PNR_BWD_WAVES
#define PNR_BWD_WAVES 8
#endif
#ifndef PNR_ABL_NOGATE
#define PNR_ABL_NOGATE 0
#endif
#ifndef PNR_BWD_ISSUERS
#define PNR_BWD_ISSUERS 0       /* waves that copy the weight pieces (0 = all 8): 4 measured SLOWER here, see Ctx::issue */
#endif

// One backward layer.  in: NA B registers (k-segments concatenated).  out[t][OFF + fb*8 + p].
// gate  : the ReLU gate BITS of the layer's forward output ([S][NFB_OUT] dwords, pnr_train_layout; nullptr: linear):
//         NFB_OUT/2 dwords per lane, loaded once per layer, instead of the 16 x NFB_OUT bytes of bf16 activations
// store : slot-ordered [S][NFB_OUT*32] destination of the gated gradient
// KEEP  : false = the gated gradient goes to `store` only (the caller reloads it with load_layer once its inputs are dead):
//         the d h layer's 136 input registers + 64 output registers + accumulators + fragment window do not fit 256, and
//         hipcc spilled ~150 dwords per lane to scratch (FETCH 1.6 GB per launch against 0.63 GB algorithmic, round 2)
template <int TILES, class CTX, int NA, int NFB_OUT, int NOUT, int OFF, bool GATED = true, int FBC = PNR_BWD_FBC, bool KEEP = true>
__device__ __forceinline__ void layer_bwd(CTX& c, const uint32_t (&in)[TILES][NA], uint32_t (&out)[TILES][NOUT],
                                          const uint16_t* gate, uint16_t* store, const int (&samp)[TILES], const int (&srow)[TILES])
{
    constexpr int G = 4 / FBC;               // A-fragment read window: 2 * G * FBC * 4 registers
    static_assert(NFB_OUT % FBC == 0 && NOUT >= OFF + NFB_OUT * 8, "bad backward layer geometry");
    uint32_t dummy[TILES][1];
    uint32_t gw[TILES][NFB_OUT / 2];
    if constexpr (GATED) {
#if PNR_ABL_NOGATE              /* ablation (results invalid): no gate-word loads -- what the per-layer global loads cost through the in-order vmcnt */
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int i = 0; i < NFB_OUT / 2; ++i) gw[t][i] = 0xffffffffu;
#else
#pragma unroll
        for (int t = 0; t < TILES; ++t) load_gates<NFB_OUT / 2>(gate, NFB_OUT * 32, samp[t], c.hi, gw[t]);
#endif
    }
#pragma unroll
    for (int cb = 0; cb < NFB_OUT / FBC; ++cb) {
        c.begin();
        const char* base = c.base();
        f32x16 acc[FBC][TILES];
#pragma unroll
        for (int b = 0; b < FBC; ++b)
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.0f;
        c.stamp(2);
        mma_chunk<PNR_PREC_BF16, TILES, FBC, G, NA, 0>(base + c.lane * 16, in, dummy, acc);
        c.stamp(3);
#pragma unroll
        for (int b = 0; b < FBC; ++b) {
            const int fb = cb * FBC + b;
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                uint32_t blk[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const uint32_t v = pack_bf16(acc[b][t][2 * p], acc[b][t][2 * p + 1]);
                    blk[p] = GATED ? gate_apply(v, gw[t][fb / 2], fb, p) : v;
                    if constexpr (KEEP) out[t][OFF + fb * 8 + p] = blk[p];
                }
                // (stores before the hand-over: moving them behind finish() -- so that its vmcnt(0) would not wait for this
                // chunk's write acknowledgements -- measured +17 % time: the gradient registers stay live across the barrier)
                store_slots(store, NFB_OUT * 32, srow[t], fb, c.hi, blk);
            }
        }
        c.finish(2 * FBC * TILES);
    }
}

// A layer output written with KEEP = false, back into registers (the wave reads what it stored itself: L2 / L1 hits)
template <int NFB, int NOUT>
__device__ __forceinline__ void load_layer(const uint16_t* store, int srow, int hi, uint32_t (&out)[NOUT])
{
    static_assert(NOUT >= NFB * 8, "register array too small");
#pragma unroll
    for (int fb = 0; fb < NFB; ++fb) {
        const u32x4* p = reinterpret_cast<const u32x4*>(store + pnr_saved_chunk(NFB * 4, srow, fb * 4 + hi * 2));
        const u32x4 a = p[0], b = p[8];
        out[fb * 8 + 0] = a[0]; out[fb * 8 + 1] = a[1]; out[fb * 8 + 2] = a[2]; out[fb * 8 + 3] = a[3];
        out[fb * 8 + 4] = b[0]; out[fb * 8 + 5] = b[1]; out[fb * 8 + 6] = b[2]; out[fb * 8 + 7] = b[3];
    }
}

// NB blocks (32 channels each) of the upstream gradient d_raw as B registers in FEAT slot order.
template <int NB>
__device__ __forceinline__ void load_draw(const MlpArgs& a, int s, int hi, int ch_base, int n_out, uint32_t (&out)[NB * 8])
{
    const float* const base = a.d_raw + ((int64_t)(ch_base + 4 * hi) * a.S + (s >= 0 ? s : 0));
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            // row = u + 4 hi: the lane-dependent part of the address (the sample, the half-wave's 4-row offset) is ONE base per
            // call; every row then adds the wave-uniform u * S.  (With the whole index per lane hipcc hoisted ~130 64-bit
            // row products out of the sample-group loop and spilled them: that was the kernel's scratch traffic.)
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * p + e;
                const int u = b * 32 + (r & 3) + 8 * (r >> 2);
                v[e] = (s >= 0 && u + 4 * hi < n_out) ? base[(int64_t)u * a.S] : 0.0f;
            }
            out[b * 8 + p] = pack_bf16(v[0], v[1]);
        }
}

template <int W, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_mlp_bwd(const MlpArgs a)
{
    using CTX = Ctx<WAVES, 4, PNR_BWD_SLOTS, PNR_BWD_ISSUERS>;
    constexpr int TILES = 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NFB = W / 32, HFB = W / 64;
    constexpr int HR = NFB * 8, GR = HFB * 8;
    constexpr int OBR = PNR_BWD_OUT_SLOTS / 32 * 8;       // B registers of a head's output gradient
    constexpr int CATR = HR + 8 + GR + GR;                // [dY_feature | d rgb,sigma | dY_sem0 | dY_inst0]

    CTX c{a, smem, (int)(threadIdx.x & 63), __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)),
          (int)((threadIdx.x & 63) >> 5), 0, 0, {0, 0}, {0, 0}};
#if PNR_TRACE
    c.tr = reinterpret_cast<unsigned long long*>(smem + PNR_BWD_SLOTS * a.slot_bytes) + c.wave * PNR_TRACE_CHUNKS * PNR_TRACE_STAMPS;
    c.titer = 0;
#endif
    const int n = c.lane & 31;
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (a.clk) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    c.start();
    const int D = a.D;
    uint16_t* const acts = a.acts;
    uint16_t* const dys = a.dys;

    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        int samp[TILES], srow[TILES];       // sample (-1 past the end) / row in the saved tensors
        {
            const int s = ((grp * WAVES + c.wave) * TILES) * 32 + n;
            samp[0] = s < a.S ? s : -1;
            srow[0] = s;
            c.st_full = true;               // stores are unmasked (rows S..S_pad receive zeros): exact store count per chunk
        }
        uint32_t cat[TILES][CATR];
#pragma unroll
        for (int i = 0; i < CATR; ++i) cat[0][i] = 0;
        uint32_t drs[TILES][8];
        load_draw<1>(a, samp[0], c.hi, 0, 4, drs[0]);
        store_slots(dys + a.dys_off[4 + D], 32, srow[0], 0, c.hi, drs[0]);        // dY of [rgb, sigma] for the wgrad
#pragma unroll
        for (int i = 0; i < 8; ++i) cat[0][HR + i] = drs[0][i];

        // d g = W_rgb^T d rgb ; gate by g ; -> dY_views
        uint32_t dyv[TILES][GR];
        layer_bwd<TILES, CTX, 8, HFB, GR, 0>(c, drs, dyv, acts + a.gate_off[3 + D], dys + a.dys_off[0], samp, srow);
        uint32_t dy[TILES][HR], dn[TILES][HR];
        if (a.head_tap) {
            // the heads read the FEATURE: their gradients join the views gradient in d F (plan order DG | DSHS | DSHI | DF | DH)
            constexpr int C3 = 3 * GR;              // [dY_views | dY_sem0 | dY_inst0]
            uint32_t c3[TILES][C3];
#pragma unroll
            for (int i = 0; i < GR; ++i) { c3[0][i] = dyv[0][i]; c3[0][GR + i] = 0; c3[0][2 * GR + i] = 0; }
            if (a.n_sem) {
                uint32_t ds[TILES][OBR];
                load_draw<PNR_BWD_OUT_SLOTS / 32>(a, samp[0], c.hi, 4, a.n_sem, ds[0]);
#pragma unroll
                for (int b = 0; b < PNR_BWD_OUT_SLOTS / 32; ++b) store_slots(dys + a.dys_off[5 + D], PNR_BWD_OUT_SLOTS, srow[0], b, c.hi, &ds[0][b * 8]);
                if (a.head_depth == 1) {        // one Linear per head: the logit gradients ARE the segment (zero-extended to H slots)
#pragma unroll
                    for (int i = 0; i < OBR; ++i) c3[0][GR + i] = ds[0][i];
                } else
                layer_bwd<TILES, CTX, OBR, HFB, C3, GR>(c, ds, c3, acts + a.gate_off[4 + D], dys + a.dys_off[2], samp, srow);
            }
            if (a.n_inst) {
                uint32_t di[TILES][OBR];
                load_draw<PNR_BWD_OUT_SLOTS / 32>(a, samp[0], c.hi, 4 + a.n_sem, a.n_inst, di[0]);
#pragma unroll
                for (int b = 0; b < PNR_BWD_OUT_SLOTS / 32; ++b) store_slots(dys + a.dys_off[6 + D], PNR_BWD_OUT_SLOTS, srow[0], b, c.hi, &di[0][b * 8]);
                if (a.head_depth == 1) {
#pragma unroll
                    for (int i = 0; i < OBR; ++i) c3[0][2 * GR + i] = di[0][i];
                } else
                layer_bwd<TILES, CTX, OBR, HFB, C3, 2 * GR>(c, di, c3, acts + a.gate_off[5 + D], dys + a.dys_off[3], samp, srow);
            }
            // d F = W_views[:, :W]^T dY_views + W_sem0^T dY_sem0 + W_inst0^T dY_inst0   (feature_linear has no activation)
            constexpr int C2 = HR + 8;              // [dY_feature | d rgb, sigma]
            uint32_t c2[TILES][C2];
            layer_bwd<TILES, CTX, C3, NFB, C2, 0, false>(c, c3, c2, nullptr, dys + a.dys_off[1], samp, srow);
#pragma unroll
            for (int i = 0; i < 8; ++i) c2[0][HR + i] = drs[0][i];
            // d h = W_feature^T d F + alpha^T d sigma ; gate by h = X_D
            layer_bwd<TILES, CTX, C2, NFB, HR, 0, true, PNR_BWD_FBC_DH>(c, c2, dy, acts + a.gate_off[1 + D], dys + a.dys_off[3 + D], samp, srow);
        } else {
        // d f = W_views[:, :W]^T dY_views  (feature_linear has no activation) -> dY_feature
        layer_bwd<TILES, CTX, GR, NFB, CATR, 0, false>(c, dyv, cat, nullptr, dys + a.dys_off[1], samp, srow);
        if (a.n_sem) {
            uint32_t ds[TILES][OBR];
            load_draw<PNR_BWD_OUT_SLOTS / 32>(a, samp[0], c.hi, 4, a.n_sem, ds[0]);
#pragma unroll
            for (int b = 0; b < PNR_BWD_OUT_SLOTS / 32; ++b) store_slots(dys + a.dys_off[5 + D], PNR_BWD_OUT_SLOTS, srow[0], b, c.hi, &ds[0][b * 8]);
            if (a.head_depth == 1) {            // one Linear per head: the logit gradients ARE the segment (zero-extended to H slots)
#pragma unroll
                for (int i = 0; i < OBR; ++i) cat[0][HR + 8 + i] = ds[0][i];
            } else
            layer_bwd<TILES, CTX, OBR, HFB, CATR, HR + 8>(c, ds, cat, acts + a.gate_off[4 + D], dys + a.dys_off[2], samp, srow);
        }
        if (a.n_inst) {
            uint32_t di[TILES][OBR];
            load_draw<PNR_BWD_OUT_SLOTS / 32>(a, samp[0], c.hi, 4 + a.n_sem, a.n_inst, di[0]);
#pragma unroll
            for (int b = 0; b < PNR_BWD_OUT_SLOTS / 32; ++b) store_slots(dys + a.dys_off[6 + D], PNR_BWD_OUT_SLOTS, srow[0], b, c.hi, &di[0][b * 8]);
            if (a.head_depth == 1) {
#pragma unroll
                for (int i = 0; i < OBR; ++i) cat[0][HR + 8 + GR + i] = di[0][i];
            } else
            layer_bwd<TILES, CTX, OBR, HFB, CATR, HR + 8 + GR>(c, di, cat, acts + a.gate_off[5 + D], dys + a.dys_off[3], samp, srow);
        }
        // d h = W_feature^T dY_feature + alpha^T d sigma + W_sem0^T dY_sem0 + W_inst0^T dY_inst0 ; gate by h = X_D
        layer_bwd<TILES, CTX, CATR, NFB, HR, 0, true, PNR_BWD_FBC_DH, false>(c, cat, dy, acts + a.gate_off[1 + D], dys + a.dys_off[3 + D], samp, srow);
        load_layer<NFB>(dys + a.dys_off[3 + D], srow[0], c.hi, dy[0]);       // cat is dead: d h comes back from where it was stored
        }
        // trunk: d X_l = W_l[:, h columns]^T dY_l ; gate by X_l ; -> dY_{l-1}
#pragma unroll 1
        for (int l = D - 1; l >= 1; --l) {
            layer_bwd<TILES, CTX, HR, NFB, HR, 0>(c, dy, dn, acts + a.gate_off[1 + l], dys + a.dys_off[3 + l], samp, srow);
#pragma unroll
            for (int i = 0; i < HR; ++i) dy[0][i] = dn[0][i];
        }
#if PNR_TRACE
        ++c.titer;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (a.clk && blockIdx.x == 0 && threadIdx.x == 0) {     // mean shader clock of this launch: cycles / 100 MHz ticks
        a.clk[0] = __builtin_amdgcn_s_memtime() - clk_c0;
        a.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#if PNR_TRACE
    __syncthreads();
    if (blockIdx.x == PNR_TRACE_WG && a.trace) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(smem + PNR_BWD_SLOTS * a.slot_bytes);
        for (int i = threadIdx.x; i < WAVES * PNR_TRACE_CHUNKS * PNR_TRACE_STAMPS; i += blockDim.x) a.trace[i] = src[i];
    }
#endif
}

template <int W, int WAVES>
static int launch_bwd(const MlpArgs& a0, hipStream_t stream)
{
    MlpArgs a = a0;
    const int lds_bytes = PNR_BWD_SLOTS * a.slot_bytes + (PNR_TRACE ? WAVES * PNR_TRACE_CHUNKS * PNR_TRACE_STAMPS * 8 : 0);
    PNR_REQUIRE(lds_bytes <= 163840, "pnr_mlp_backward: weight slots of %d bytes exceed the 160 KiB LDS", lds_bytes);
    const int per_group = 32 * WAVES;
    a.n_groups = (a.S + per_group - 1) / per_group;
    auto kern = k_mlp_bwd<W, WAVES>;
    static thread_local bool configured = false;
    if (!configured) {
        PNR_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        configured = true;
    }
    const int ncu = pnr_cu_count();               // persistent: one workgroup per CU
    const int grid = a.n_groups < ncu ? a.n_groups : ncu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds_bytes, stream, a);
    PNR_CHECK_LAUNCH("pnr_mlp_backward");
    return PNR_OK;
}

PNR_EXPORT int pnr_mlp_backward(const pnr_mlp_desc* desc, const void* packed_bwd, const float* d_raw, const void* acts,
                                void* dys, int64_t n_rays, int n_samples, void* stream)
{
    int rc = pnr_mlp_validate(desc);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(desc->precision == PNR_PREC_BF16, "pnr_mlp_backward: bf16 only");
    PNR_REQUIRE(desc->n_sem <= PNR_BWD_OUT_SLOTS && desc->n_inst <= PNR_BWD_OUT_SLOTS,
                "pnr_mlp_backward: n_sem / n_inst must be <= %d", PNR_BWD_OUT_SLOTS);
    PNR_REQUIRE(n_rays >= 0 && n_samples >= 1, "pnr_mlp_backward: bad size");
    if (n_rays == 0) return PNR_OK;
    PNR_REQUIRE(packed_bwd && d_raw && acts && dys, "pnr_mlp_backward: null pointer");
    PNR_REQUIRE(n_rays * (int64_t)n_samples < ((int64_t)1 << 31) - 4096, "pnr_mlp_backward: R*N exceeds 2^31");
    PNR_REQUIRE((((uintptr_t)packed_bwd | (uintptr_t)acts | (uintptr_t)dys) & 15) == 0,
                "pnr_mlp_backward: buffers must be 16-byte aligned");
    PnrBPlan plan;
    pnr_build_bwd_plan(*desc, plan);
    MlpArgs a;
    memset(&a, 0, sizeof(a));
    a.data = (const uint8_t*)packed_bwd + plan.data_off;
    a.table = (const pnr_chunk_entry*)((const uint8_t*)packed_bwd + plan.table_off);
    a.n_chunks = (int)plan.chunks.size();
    a.slot_bytes = plan.max_chunk_frags * PNR_FRAG_BYTES;
    a.S = (int)(n_rays * n_samples); a.N = n_samples;
    a.D = desc->D; a.skip = desc->skip; a.n_sem = desc->n_sem; a.n_inst = desc->n_inst;
    a.head_tap = desc->head_tap; a.head_depth = desc->head_depth == 1 ? 1 : 2;
    a.acts = (uint16_t*)acts; a.d_raw = d_raw; a.dys = (uint16_t*)dys;
    pnr_train_layout(*desc, a.S, a.acts_off, a.dys_off, a.gate_off);
#if PNR_TRACE
    if (const char* e = getenv("PNR_TRACE_PTR")) a.trace = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
    hipStream_t st = (hipStream_t)stream;
    return desc->W == 256 ? launch_bwd<256, PNR_BWD_WAVES>(a, st) : launch_bwd<128, PNR_BWD_WAVES>(a, st);
}
