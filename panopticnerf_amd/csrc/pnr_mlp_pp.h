// Ping-pong form of the fused MLP's weight stream (bf16, one 32-sample tile per wave, 8 waves):
// the two waves of every SIMD work in PHASE OPPOSITION, separated by s_barrier.
//
// Why.  In the lock-step kernel (pnr_mlp_core.h) all 8 waves run the same chunk phase at the same
// time: refill issue, bias + first fragment reads, the MFMAs, the epilogue, the barrier.  The matrix
// pipe is ~90 % busy inside the MFMA window but idle for the ~1100 cycles of every ~3400-cycle chunk
// that are not MFMAs (profiles/README.md, round 1).  Here waves 0-3 (group P, one per SIMD) run the
// MFMAs of chunk c while waves 4-7 (group Q, the other wave of each SIMD) run everything else of
// their chunk c -- epilogue of c-1, weight refill (LDS-DMA), bias and first-fragment reads -- and
// after the barrier the roles swap ("matrix beside memory", MI355X_MICROARCH.md, two waves per SIMD).
//
//   phase     2c-1        2c          2c+1        2c+2
//   P:        L(c)        M(c)        L(c+1)      M(c+1)         L = E(c-1) + refill + bias/first reads of c
//   Q:        M(c-1)      L(c)        M(c)        L(c+1)         M = the chunk's MFMAs
//
// LDS: THREE weight slots (chunk c lives from phase 2c-1, when P reads its bias, to 2c+1, when Q
// finishes its MFMAs).  In L(k), P refills its share of chunk k+1 and Q its share of chunk k+2 -- the
// slot of chunk k-2 / k-1, whose last reader passed the previous barrier -- and each wave waits for
// its own pieces (vmcnt(0)) at the END of its next M, almost two phases later and BEFORE that phase's
// barrier: no phase ever waits for a copy issued inside it, and every piece is covered by its issuer's
// wait plus a barrier before any wave reads it.  Both groups execute the same code; Q is one barrier behind.
// Order inside L(k+1), which starts when M(k) ends: refill pieces INTERLEAVED with the epilogue of
// chunk k (a wave blocks 100-200 cycles per 1 KiB LDS-DMA piece while the CU's queue is full; the
// pack/ReLU VALU work runs in those gaps), then bias + first fragments of chunk k+1, then the barrier.
//
// A wave alone on the matrix pipe must not stall on LDS latency (the lock-step kernel's reads were
// 1-2 MFMAs ahead and waited with lgkmcnt(0); the sibling wave filled the gaps).  Here the A
// fragments go through a ring of P registers quads: the read of fragment i+P-1 is issued right
// after MFMA i, into the slot MFMA i-1 used, and MFMA i waits with a COUNTED lgkmcnt (P-2 younger
// reads stay in flight: ~7 MFMAs = 220+ cycles of cover).  hipcc does not keep that order (it re-sorts
// the ds_reads of a sched_group and then needs lgkmcnt(0)), so the LDS reads and their waits are
// inline asm, tied to the MFMA builtins through register dependencies and pinned with
// sched_barrier(0).  LDS reads return in order, so extra older LGKM operations only make a counted
// wait conservative, never wrong.
#pragma once
#include <utility>

#include "pnr_mlp_core.h"

template <int... I, class F>
__device__ __forceinline__ void pp_static_for_impl(std::integer_sequence<int, I...>, F&& f)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void pp_static_for(F&& f)
{
    pp_static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// LDS read of one 16-byte fragment share / bias quad: address VGPR + immediate offset (< 64 KiB)
template <int OFF>
__device__ __forceinline__ void pp_lds_read(u32x4& dst, uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void pp_lds_read(f32x4& dst, uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void pp_lds_read_b32(float& dst, uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// wait until at most N younger LGKM operations are outstanding; `a` (the register the oldest
// still-needed read fills) is an in/out operand so that its consumer cannot be scheduled above the wait
template <int N>
__device__ __forceinline__ void pp_wait(u32x4& a)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}

#ifndef PNR_PP_RING
#define PNR_PP_RING 4        /* measured: 4 = 5 > 6 > 8 > 10 (3 fragments = ~100 cycles in flight cover the LDS latency; fewer registers) */
#endif
#ifndef PNR_PP_PRIO
#define PNR_PP_PRIO 1        /* s_setprio of the M phase: +1 % measured */
#endif
// cache policy of the refill's LDS-DMA pieces (0 = default, 1 = sc0, 2 = nt).  Round 3, on the scalar-base pieces, same box,
// two passes: default 11.079 / 11.065 ms @1795-1805 MHz, sc0 11.071 / 11.072, nt 10.985 / 10.992 @1818-1840 (-0.75 %, and
// at a higher clock: less power) -> nt.  (Rounds 1-2, per-lane-pointer pieces: nt was +1 % -- re-measured, not assumed.)
// sc0 nt / sc1 nt / sc0 sc1 nt: the same as nt (10.95-10.99 ms where nt gives 10.97); sc1 alone: 11.09.
#ifndef PNR_PP_DMA_AUX
#define PNR_PP_DMA_AUX 2
#endif
// timing-only ablations (A/B builds; results are wrong): 1 = no refill pieces, 2 = no fragment reads inside the MFMA loop
// 1: the first fragments of a layer's 2nd, 3rd, ... chunk are read right after the previous chunk's M -> L barrier
#ifndef PNR_PP_HEAD
#define PNR_PP_HEAD 0        /* MFMAs of a chunk issued before its L -> M barrier (soft hand-over of the matrix pipe) */
#endif
#ifndef PNR_PP_EARLY
#define PNR_PP_EARLY 1
#endif
#ifndef PNR_PP_EARLY_BIAS
#define PNR_PP_EARLY_BIAS 0
#endif
#ifndef PNR_PP_ABL
#define PNR_PP_ABL 0
#endif
#ifndef PNR_TRACE_MID
#define PNR_TRACE_MID 0
#endif
#ifndef PNR_ABL_INB
#define PNR_ABL_INB 0
#endif


template <int WAVES>
struct CtxPP {
    static constexpr int P = PNR_PP_RING;
    const MlpArgs& a;
    char* smem;
    int lane, wave, hi, grp;           // grp 0: waves 0..WAVES/2-1 (P), 1: the others (Q)
    int ci;                            // chunk whose L / M comes next
    int slot_off;                      // byte offset of its LDS slot (0, slot_bytes, 2*slot_bytes)
    uint32_t lds_frag, lds_bias;       // LDS byte addresses of smem + lane*16 / smem + hi*16
    pnr_chunk_entry en;                // table entry of the chunk this wave refills in its next L: ci + 1 + grp
#if PNR_TRACE
    unsigned long long* tr;
    int titer;
    __device__ __forceinline__ void stamp(int k)
    {
        if (!((PNR_TRACE_MASK >> k) & 1)) return;
        if (blockIdx.x == PNR_TRACE_WG && titer == PNR_TRACE_ITER && ci < PNR_TRACE_CHUNKS) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) tr[ci * PNR_TRACE_STAMPS + k] = t;
        }
    }
#else
    __device__ __forceinline__ void stamp(int) {}
#endif

    __device__ __forceinline__ int wrap(int i) const { return i >= a.n_chunks ? i - a.n_chunks : i; }
    __device__ __forceinline__ int wrap_slot(int o) const { return o >= 3 * a.slot_bytes ? o - 3 * a.slot_bytes : o; }
    __device__ __forceinline__ pnr_chunk_entry entry(int idx) const
    {
        table_ptr t = (table_ptr)(uintptr_t)a.table;
        pnr_chunk_entry e;
        e.off_frag = t[idx].off_frag;
        e.nfrag = t[idx].nfrag;
        return e;
    }
    // this wave's share (fragments wave, wave + WAVES, ...) of a chunk: L2 -> LDS, asynchronous
    __device__ __forceinline__ void issue(const pnr_chunk_entry& e, int off) const
    {
        const uint8_t* src = a.data + (size_t)e.off_frag * PNR_FRAG_BYTES;
        char* dst = smem + off;
        for (int f = wave; f < (int)e.nfrag; f += WAVES) pnr_dma_piece(src + (size_t)f * PNR_FRAG_BYTES, dst + f * PNR_FRAG_BYTES, lane * 16);
    }
    __device__ __forceinline__ void barrier() const
    {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void start()
    {
        ci = 0;
        slot_off = 0;
        lds_frag = (uint32_t)(uintptr_t)(lds_void*)smem + lane * 16;
        lds_bias = (uint32_t)(uintptr_t)(lds_void*)smem + hi * 16;
        issue(entry(0), 0);
        issue(entry(wrap(1)), a.slot_bytes);
        if (grp) issue(entry(wrap(2)), 2 * a.slot_bytes);      // what Q's L(0) refills (P's L(0) share, chunk 1, is in already)
        en = entry(wrap(2 + grp));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier();
        if (grp) barrier();                 // Q runs one phase behind P
    }
    __device__ __forceinline__ void end()
    {
        if (!grp) barrier();                // pairs with Q's last phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the refills issued past the last chunk
#if PNR_PP_ABL & 16
        asm volatile("" :: "v"(abl_sink));
#endif
    }
    __device__ __forceinline__ uint32_t frag_addr() const { return lds_frag + slot_off; }
    __device__ __forceinline__ uint32_t bias_addr() const { return lds_bias + slot_off; }
    __device__ __forceinline__ uint32_t next_frag_addr() const { return lds_frag + wrap_slot(slot_off + a.slot_bytes); }
    __device__ __forceinline__ uint32_t next_bias_addr() const { return lds_bias + wrap_slot(slot_off + a.slot_bytes); }
    // After the chunk's MFMAs: close the M phase.  BEFORE the barrier this wave waits for its own refill pieces (issued at
    // the start of its previous L, ~2 phases ago: landed long since, the wait is free) -- so that every piece is covered by
    // its issuer's vmcnt(0) AND a barrier before anybody reads it, including the other waves of the issuer's own group,
    // whose early fragment reads follow this barrier directly.  (Waiting after the barrier covers only the reader's own
    // pieces: a race that showed as run-to-run differences in a ragged last sample group of the training forward, where
    // the pieces queue behind activation stores.)
    __device__ __forceinline__ void m_done()
    {
        stamp(3);
        wait_pieces();
        stamp(5);
        barrier();
        stamp(4);
    }
    // vmcnt wait that covers every refill piece of this wave but not the `pending_stores` raw-output stores it issued
    // AFTER them (pp_layer_out): s_waitcnt takes an immediate, hence the ladder (uniform branches, output chunks only)
    int pending_stores;
    __device__ __forceinline__ void wait_pieces()
    {
        const int n = pending_stores;
        pending_stores = 0;
        if (n == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
#define PNR_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        switch (n) {
            PNR_VM(1) PNR_VM(2) PNR_VM(3) PNR_VM(4) PNR_VM(5) PNR_VM(6) PNR_VM(7) PNR_VM(8)
            PNR_VM(9) PNR_VM(10) PNR_VM(11) PNR_VM(12) PNR_VM(13) PNR_VM(14) PNR_VM(15) PNR_VM(16)
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
#undef PNR_VM
    }
    // Refill of the next L -- this wave's share (fragments wave, wave + WAVES, ...) of chunk ci+2+grp, i.e. of chunk
    // (k+1)+1+grp for the L(k+1) that starts when M(k) ends -- issued piece by piece so that the caller can put the
    // epilogue's VALU work between the pieces: a wave blocks while the CU's LDS-DMA queue is full (100-200 cycles per
    // 1 KiB piece beside the partner's MFMA/ds_read stream), and the VALU work runs in exactly those gaps.
    const uint8_t* rf_src;
    const uint8_t* rf_base;        // the wave-uniform part of rf_src: what the pieces address (pnr_dma_piece)
    char* rf_dst;
    int rf_f, rf_n;
    __device__ __forceinline__ void refill_begin()
    {
#if !PNR_TRACE_MID          /* -DPNR_TRACE_MID=1: stamps 0 / 7 / 1 mark the quarters of the M phase instead (tools/mlp_trace_m.py) */
        stamp(0);
#endif
        rf_base = a.data + (size_t)en.off_frag * PNR_FRAG_BYTES;
        rf_src = rf_base + lane * 16;
        rf_dst = smem + wrap_slot(slot_off + (2 + grp) * a.slot_bytes);     // Q: chunk ci+3 takes over chunk ci's own slot
        rf_f = wave;
        rf_n = (int)en.nfrag;
#ifdef PNR_ABL_SKIP_MASK
        // ablation (results invalid): the chunks of the mask are never refilled -- what a chunk RESIDENT in the LDS would save
        if ((PNR_ABL_SKIP_MASK >> wrap(ci + 2 + grp)) & 1ull) rf_n = 0;
#endif
        en = entry(wrap(ci + 3 + grp));
    }
    u32x4 abl_sink;        // PNR_PP_ABL & 16 only
    __device__ __forceinline__ void piece(int f)
    {
#if PNR_PP_ABL & 16
        // ablation (results invalid): the same L2 reads, issued at the same places, but into a register nobody reads -- no LDS write
        const uint8_t* src = rf_src + (size_t)f * PNR_FRAG_BYTES;
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(abl_sink) : "v"(src) : "memory");
#else
        pnr_dma_piece<PNR_PP_DMA_AUX>(rf_base + (size_t)f * PNR_FRAG_BYTES, rf_dst + f * PNR_FRAG_BYTES, lane * 16);
#endif
    }
    __device__ __forceinline__ void refill_one()
    {
        if (!(PNR_PP_ABL & 1) && rf_f < rf_n) {
            piece(rf_f);
            rf_f += WAVES;
        }
    }
    __device__ __forceinline__ void refill_rest()
    {
        for (; !(PNR_PP_ABL & 1) && rf_f < rf_n; rf_f += WAVES) piece(rf_f);
#if !PNR_TRACE_MID
        stamp(1);
#endif
    }
    __device__ __forceinline__ void advance()
    {
        ci = wrap(ci + 1);
        slot_off = wrap_slot(slot_off + a.slot_bytes);
    }
};

// One chunk = FBC 32-row output blocks x KS k-steps; fragments in CONSUMPTION order i = ks*FBC + b.
template <int FBC, int NA, int NB>
struct PPChunk {
    static constexpr int P = PNR_PP_RING;
    static constexpr int KSA = NA / 4, KSB = NB / 4, KS = KSA + KSB, NF = KS * FBC;
    static constexpr int frag_off(int i) { return ((i % FBC) * KS + (i / FBC)) * PNR_FRAG_BYTES; }
    static constexpr int BIAS_OFF = FBC * KS * PNR_FRAG_BYTES;

    // L: the first P-1 fragments of the chunk into the ring (asynchronous; issued EARLY in the L phase, right after the
    // M -> L barrier of the previous chunk, so that they land under the refill / epilogue work)
    __device__ __forceinline__ static void first_frags(uint32_t fa, u32x4 (&A)[P])
    {
        pp_static_for<(P - 1 < NF ? P - 1 : NF)>([&](auto I) {
            constexpr int i = I;
            pp_lds_read<frag_off(i)>(A[i % P], fa);
        });
    }
    // bias quads of the chunk at `ba` (asynchronous) ...
    __device__ __forceinline__ static void bias_issue(uint32_t ba, f32x4 (&q)[FBC][4])
    {
        pp_static_for<FBC * 4>([&](auto J) {
            constexpr int b = J / 4, m = J % 4;
            if constexpr (PNR_PP_ABL & 8) { q[b][m] = f32x4{0, 0, 0, 0}; (void)ba; }      // ablation: no bias reads (results invalid)
            else pp_lds_read<BIAS_OFF + b * 128 + m * 32>(q[b][m], ba);
        });
    }
    // ... and, at the end of L: drain every LDS read of the phase (this is the memory half of the pairing: the partner
    // wave owns the matrix pipe meanwhile), accumulators <- bias.  The quads are in/out operands of the wait, so
    // whatever the compiler does with them (ideally nothing: they coalesce into the accumulator tuples) happens after
    // the data has arrived.
    __device__ __forceinline__ static void bias_finish(f32x4 (&q)[FBC][4], f32x16 (&acc)[FBC])
    {
#pragma unroll
        for (int b = 0; b < FBC; ++b)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[b][0]), "+v"(q[b][1]), "+v"(q[b][2]), "+v"(q[b][3]));
#pragma unroll
        for (int b = 0; b < FBC; ++b)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[b][4 * m + 0] = q[b][m][0]; acc[b][4 * m + 1] = q[b][m][1];
                acc[b][4 * m + 2] = q[b][m][2]; acc[b][4 * m + 3] = q[b][m][3];
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    // transposed product (SWAP): every register of lane j starts at the bias of output channel j = lane & 31.
    // ba0: LDS address of the chunk's slot (no lane part); FBC == 1.
    __device__ __forceinline__ static void prologue_swapped(uint32_t fa, uint32_t ba0, int lane, u32x4 (&A)[P], f32x16 (&acc)[FBC])
    {
        static_assert(FBC == 1, "transposed output blocks are single-block chunks");
        first_frags(fa, A);
        float bj;
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bj) : "v"(ba0 + (uint32_t)(lane & 31) * 4u), "n"(BIAS_OFF));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bj));
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = bj;
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ static void prologue(uint32_t fa, uint32_t ba, u32x4 (&A)[P], f32x16 (&acc)[FBC])
    {
        f32x4 q[FBC][4];
        first_frags(fa, A);
        bias_issue(ba, q);
        bias_finish(q, acc);
    }

    // M: the chunk's MFMAs, one wave alone on its SIMD's matrix pipe.
    // PNR_PP_HEAD > 0 (soft hand-over): the first HEAD MFMAs are issued BEFORE the L -> M barrier, at low priority.  The chunk's
    // weights landed two phases ago and were covered by a barrier then (the L phase already reads its bias and first
    // fragments), so only the pipe hand-over hangs on this barrier: the partner group's last MFMAs keep priority, and these
    // fill its gaps and the ~100 cycles the barrier itself takes.
    // SWAP: the fragment is the MFMA's B operand and the activations its A operand.  A weight fragment (lane = output row, 8
    // consecutive k) IS the B-operand image of the transposed weights (lane = output column, 8 consecutive k), and the
    // activation registers (lane = sample, 8 consecutive k) ARE an A operand with rows = samples: the product comes out
    // transposed -- lane = output channel, register r = sample row(r, hi) -- which is what the fused compositing epilogue
    // wants for the logit blocks (the weighted sum over the samples becomes 16 FMAs per lane instead of 16 butterflies).
    template <bool SWAP = false, class BARRIER>
    __device__ __forceinline__ static void mma(uint32_t fa, u32x4 (&A)[P], const uint32_t (&inA)[NA],
                                               const uint32_t (&inB)[NB > 0 ? NB : 1], f32x16 (&acc)[FBC], BARRIER&& barrier)
    {
        constexpr int HEAD = PNR_PP_HEAD < NF / 2 ? PNR_PP_HEAD : NF / 2;
        if constexpr (HEAD == 0) {
            barrier();
#if PNR_PP_PRIO
            __builtin_amdgcn_s_setprio(PNR_PP_PRIO);
#endif
        }
        pp_static_for<NF>([&](auto I) {
            constexpr int i = I;
            constexpr int ks = i / FBC, b = i % FBC;
            constexpr int younger = (NF - 1 - i) < (P - 2) ? (NF - 1 - i) : (P - 2);
            if constexpr (HEAD > 0 && i == HEAD) {
                barrier();
#if PNR_PP_PRIO
                __builtin_amdgcn_s_setprio(PNR_PP_PRIO);
#endif
            }
#if PNR_TRACE_MID
            if constexpr (i == NF / 4) barrier(0);          // trace builds: the callers' functor stamps when given a number
            if constexpr (i == NF / 2) barrier(7);
            if constexpr (i == 3 * NF / 4) barrier(1);
#endif
            pp_wait<younger>(A[i % P]);
            if constexpr (SWAP) {
                const uint32_t* act = ks < KSA ? &inA[4 * (ks < KSA ? ks : 0)] : &inB[4 * (ks >= KSA ? ks - KSA : 0)];
                u32x4 av;
                av[0] = act[0]; av[1] = act[1]; av[2] = act[2]; av[3] = act[3];
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, A[i % P]),
                                                                  acc[b], 0, 0, 0);
            } else if constexpr (ks < KSA) acc[b] = kstep<PNR_PREC_BF16>(A[i % P], &inA[4 * ks], acc[b]);
#if PNR_ABL_INB
            else acc[b] = kstep<PNR_PREC_BF16>(A[i % P], &inA[4 * (ks - KSA)], acc[b]);      // ablation (results invalid): no second operand segment
#else
            else acc[b] = kstep<PNR_PREC_BF16>(A[i % P], &inB[4 * (ks - KSA)], acc[b]);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i + P - 1 < NF && !(PNR_PP_ABL & 2)) {
                pp_lds_read<frag_off(i + P - 1)>(A[(i + P - 1) % P], fa);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
#if PNR_PP_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
};
