// K3 (+K2 fused): the NeRF MLP with semantic / instance heads as ONE fused gfx950 kernel.
// Reference: Network / NeRF forward + Embedder (branch not in the
// mount, see include/pnr.h).  Design: pnr_mlp_layout.h.
//
//  * Every layer is evaluated transposed (H^T = W * H_in^T) so the 32x32 MFMA accumulator of
//    one layer is, after bias/ReLU/convert, directly the B operand of the next: activations
//    stay in VGPRs/AGPRs from gamma(x) to the raw outputs, never touching LDS or HBM.
//  * Weights are the A operand.  They are pre-permuted on the host into 1 KiB fragments in
//    consumption order and streamed L2 -> LDS with global_load_lds (16 B/lane, lane-linear,
//    conflict-free ds_read_b128) into a ring of LDS slots, two (layer, 32-row block) chunks ahead
//    of the MFMAs; the 8 waves of a workgroup share every fragment and synchronise through LDS
//    counters instead of s_barrier (struct Ctx).
//  * gamma(x), gamma(d) are computed in registers by the lanes that need them (the two
//    half-waves split the frequency bands), so the 63/27-wide encodings never exist in memory.
//  * bf16 path: v_mfma_f32_32x32x16_bf16, fp32 accumulate, RNE conversion of activations.
//    fp32 path (parity mode): v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain.
//
// HBM traffic per sample: 4 B of z (+32 B/ray) in, 4*(4+C+K) B of raw out; the kernel is
// MFMA-bound (1.19 MFLOP/sample trunk + heads).
#include "pnr_common.h"
#include "pnr_mlp_layout.h"
#include "pnr_mlp_plan.h"

int pnr_mlp_validate(const pnr_mlp_desc* d);

// PNR_FLOW 1: free-running waves + LDS-counter flow control; 0: one s_barrier per chunk (see struct Ctx)
#ifndef PNR_FLOW
#define PNR_FLOW 1
#endif
// ablation switches: timing experiments only (results are wrong by construction), never set in the product build
// PNR_TRACE: per-chunk s_memtime stamps of one wave into a debug buffer (tools/mlp_trace.py); trace builds only
#ifndef PNR_TRACE
#define PNR_TRACE 0
#endif
#ifndef PNR_ABL_NOSYNC
#define PNR_ABL_NOSYNC 0
#endif
#ifndef PNR_ABL_NODMA
#define PNR_ABL_NODMA 0
#endif
#ifndef PNR_ABL_NOMFMA
#define PNR_ABL_NOMFMA 0
#endif
// optimisation switches (A/B builds: make EXTRA="-DPNR_OPT_...=0")
#ifndef PNR_OPT_STORE_AFTER_BARRIER
#define PNR_OPT_STORE_AFTER_BARRIER 1
#endif
#ifndef PNR_OPT_PREFETCH_INPUTS
#define PNR_OPT_PREFETCH_INPUTS 1
#endif
#ifndef PNR_OPT_FAST_EMBED
#define PNR_OPT_FAST_EMBED 1
#endif
#ifndef PNR_OPT_PINGPONG
#define PNR_OPT_PINGPONG 0
#endif
#ifndef PNR_MLP_DEFAULT_VARIANT
#define PNR_MLP_DEFAULT_VARIANT 3
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

struct MlpArgs {
    const uint8_t* data;            // fragment stream (device)
    const pnr_chunk_entry* table;   // chunk table (device)
    int n_chunks, slot_bytes;
    const float* rays; const float* z;
    int S, N, n_groups;
    float* raw; int64_t ss, sc;
    int D, skip, n_sem, n_inst;
    unsigned long long* trace;      // PNR_TRACE builds: [iter][chunk][8] cycle stamps of (block 0, wave trace_wave)
    int trace_wave;
};

enum { MODE_RELU = 0, MODE_LINEAR = 1 };

template <int PREC> struct PrecT;
template <> struct PrecT<PNR_PREC_BF16> { static constexpr int RPB = 8; };    // B regs per 32 input features
template <> struct PrecT<PNR_PREC_FP32> { static constexpr int RPB = 16; };

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)
{
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

// One k-step: 16 bytes of A per lane against 4 B registers.
template <int PREC>
__device__ __forceinline__ f32x16 kstep(const u32x4& a, const uint32_t* b, f32x16 acc)
{
    if constexpr (PREC == PNR_PREC_BF16) {
        u32x4 bv;
        bv[0] = b[0]; bv[1] = b[1]; bv[2] = b[2]; bv[3] = b[3];
#if PNR_ABL_NOMFMA
        acc[0] += __uint_as_float(a[0] ^ bv[0]); acc[5] += __uint_as_float(a[1] ^ bv[1]);
        acc[10] += __uint_as_float(a[2] ^ bv[2]); acc[15] += __uint_as_float(a[3] ^ bv[3]);
        return acc;
#endif
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bv),
                                                        acc, 0, 0, 0);
    } else {
        // NB: __builtin_bit_cast(float, a[j]) on an ext-vector ELEMENT miscompiles with ROCm 7.2's
        // clang (every j reads element 0); copy the element to a scalar first.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t av = a[j], bv = b[j];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av), __uint_as_float(bv), acc, 0, 0, 0);
        }
        return acc;
    }
}

// The chunk table is read through the constant address space so that hipcc emits scalar
// (s_load) instead of vector loads: a vector load here costs an L2 round trip AND a vmcnt(0)
// that drains the in-flight LDS-DMA, once per chunk.
typedef const __attribute__((address_space(4))) pnr_chunk_entry* table_ptr;

// ---- weight-stream flow control -------------------------------------------------------------
// The packed weight stream is consumed chunk by chunk from a ring of NS LDS slots that the
// workgroup's waves fill cooperatively by LDS-DMA, DIST chunks ahead.  Two schemes:
//
//  FLOW (default): free-running waves.  Per slot two monotonic LDS counters: ready[s] counts the
//    waves whose DMA share of the slot's chunk has landed (vmcnt), done[s] the waves that have
//    issued their last ds_read of it.  A wave reads a chunk when ready == WAVES*(uses+1) and
//    refills a slot when done == WAVES*uses.  There is no s_barrier in steady state, so the two
//    waves of a SIMD drift out of phase (<= NS-DIST-1 chunks) and one wave's non-MFMA work (DMA
//    issue, waits, epilogue VALU, first LDS reads) overlaps its partner's MFMAs.  With a per-chunk
//    s_barrier the 8 waves run those phases in lockstep and the matrix pipe idles half the time
//    (profiles/r01*: 2000 cycles per 1024 MFMA-cycles).
//  !FLOW: one s_barrier per chunk, two slots (the first working version; kept for A/B).
template <int WAVES, int NS_, int DIST_, int GDB_, bool FLOW>
struct Ctx {
    static constexpr int GDB = GDB_;   // A-fragment read-ahead (k-steps)
    static constexpr int NS = NS_, DIST = DIST_;
    const MlpArgs& a;
    char* smem;                        // slot 0
    volatile lds_u32* cnt;             // ready[NS] | done[NS]  (LDS address space: ds_read / ds_add, never flat)
    int lane, wave, hi;
    int ci;                            // table index of the current chunk
    int slot, uses;                    // slot of the current chunk, times that slot was used before
    int pslot, puses;                  // same for the chunk DIST ahead (the DMA target)
    int sig_slot;                      // slot whose DMA this wave issued last (-1: none pending)
    pnr_chunk_entry eD, eD1;           // table entries of chunks ci+DIST, ci+DIST+1 (fetched early, scalar)
    int iter;

    __device__ __forceinline__ void stamp(int k) const
    {
#if PNR_TRACE
        if (a.trace && blockIdx.x == 0 && wave == a.trace_wave && iter < 4) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) a.trace[((size_t)iter * a.n_chunks + ci) * 8 + k] = t;
        }
#endif
    }
    __device__ __forceinline__ int wrap(int i) const { return i >= a.n_chunks ? i - a.n_chunks : i; }
    __device__ __forceinline__ pnr_chunk_entry entry(int idx) const
    {
        table_ptr t = (table_ptr)(uintptr_t)a.table;
        pnr_chunk_entry e;
        e.off_frag = t[idx].off_frag;
        e.nfrag = t[idx].nfrag;
        return e;
    }
    // Issue the L2 -> LDS copy of a chunk into slot `sl` (asynchronous; LDS-DMA, 1 KiB per wave-instruction).
    __device__ __forceinline__ void issue(const pnr_chunk_entry& e, int sl) const
    {
#if PNR_ABL_NODMA
        if (iter > 0) return;
#endif
        const uint8_t* src = a.data + (size_t)e.off_frag * PNR_FRAG_BYTES + lane * 16;
        char* dst = smem + sl * a.slot_bytes;
        for (int f = wave; f < (int)e.nfrag; f += WAVES)
            __builtin_amdgcn_global_load_lds((const void*)(src + (size_t)f * PNR_FRAG_BYTES),
                                             (lds_void*)(dst + f * PNR_FRAG_BYTES), 16, 0, 0);
    }
    __device__ __forceinline__ void wait_ge(int idx, uint32_t target) const
    {
#if !PNR_ABL_NOSYNC
        while (cnt[idx] < target) __builtin_amdgcn_s_sleep(1);
#endif
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ void signal(int idx) const
    {
        asm volatile("" ::: "memory");
        if (lane == 0)
            __hip_atomic_fetch_add(const_cast<lds_u32*>(cnt) + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    static __device__ __forceinline__ void bump(int& sl, int& us) { if (++sl == NS) { sl = 0; ++us; } }

    __device__ __forceinline__ void start()
    {
        if (threadIdx.x < 2 * NS) cnt[threadIdx.x] = 0;
        __syncthreads();
        ci = 0; slot = 0; uses = 0; iter = 0; sig_slot = -1;
#pragma unroll
        for (int k = 0; k < DIST; ++k) issue(entry(wrap(k)), k);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (FLOW) {
#pragma unroll
            for (int k = 0; k < DIST; ++k) signal(k);
        } else {
            __syncthreads();
        }
        pslot = DIST % NS; puses = DIST / NS;
        eD = entry(wrap(DIST));
        eD1 = entry(wrap(DIST + 1));
    }
    // Top of a chunk: refill the ring DIST chunks ahead, then make sure this chunk's weights are in LDS.
    __device__ __forceinline__ void begin()
    {
        stamp(0);
        if constexpr (FLOW) {
            if (puses > 0) wait_ge(NS + pslot, (uint32_t)(WAVES * puses));   // every wave is done reading the slot
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // my DMA of the previous chunk landed
            if (sig_slot >= 0) signal(sig_slot);
            issue(eD, pslot);
            sig_slot = pslot;
            stamp(1);
            wait_ge(slot, (uint32_t)(WAVES * (uses + 1)));                    // all 8 shares of THIS chunk landed
        } else {
            issue(eD, pslot);
            stamp(1);
        }
    }
    __device__ __forceinline__ const char* base() const { return smem + slot * a.slot_bytes; }
    // All ds_reads of the current chunk have been issued (LDS executes a wave's DS ops in order).
    __device__ __forceinline__ void reads_done() const
    {
        stamp(2);
        if constexpr (FLOW) signal(NS + slot);
    }
    __device__ __forceinline__ void finish()
    {
        stamp(3);
        if constexpr (!FLOW) {
#if !PNR_ABL_NOSYNC
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#endif
        }
        stamp(5);
        bump(slot, uses);
        bump(pslot, puses);
        ci = wrap(ci + 1);
        eD = eD1;
        int nx = ci + DIST + 1;
        if (nx >= a.n_chunks) nx -= a.n_chunks;
        if (nx >= a.n_chunks) nx -= a.n_chunks;
        eD1 = entry(nx);
    }
};

__device__ __forceinline__ void load_bias(const char* bias_frag, int hi, f32x16& acc)
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias_frag + (8 * m + 4 * hi) * 4);
        acc[4 * m + 0] = b[0]; acc[4 * m + 1] = b[1]; acc[4 * m + 2] = b[2]; acc[4 * m + 3] = b[3];
    }
}


// All MFMAs of one chunk: FBC output blocks x KS k-steps x TILES sample tiles.  The FBC blocks are
// FBC independent accumulator chains issued round-robin per k-step, so consecutive MFMAs never
// share an accumulator (FBC >= 2) and the ds_reads slotted between them cost ~6 cycles instead
// of ~43 (pnr_mlp_layout.h).  A fragments are read from LDS G k-steps ahead of their MFMAs; the
// sched_group_barrier sequence pins that interleave (hipcc otherwise either sinks every read to
// just before its use or hoists all of them to the chunk top, +64-96 VGPRs).
template <int PREC, int TILES, int FBC, int G, int NA, int NB>
__device__ __forceinline__ void mma_chunk(const char* frag, const uint32_t (&inA)[TILES][NA],
                                          const uint32_t (&inB)[TILES][NB > 0 ? NB : 1], f32x16 (&acc)[FBC][TILES])
{
    constexpr int KSA = NA / 4, KSB = NB / 4, KS = KSA + KSB, NG = (KS + G - 1) / G;
    constexpr int MPK = PREC == PNR_PREC_BF16 ? 1 : 4;   // MFMAs per k-step per tile per block
    u32x4 A[2][G][FBC];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
        for (int b = 0; b < FBC; ++b)
            if (j < KS) A[0][j][b] = *reinterpret_cast<const u32x4*>(frag + (b * KS + j) * PNR_FRAG_BYTES);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int ks = (g + 1) * G + j;
#pragma unroll
                for (int b = 0; b < FBC; ++b)
                    if (ks < KS) A[(g + 1) & 1][j][b] = *reinterpret_cast<const u32x4*>(frag + (b * KS + ks) * PNR_FRAG_BYTES);
            }
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int ks = g * G + j;
            if (ks < KS) {
#pragma unroll
                for (int b = 0; b < FBC; ++b)
#pragma unroll
                    for (int t = 0; t < TILES; ++t) {
                        if (ks < KSA) acc[b][t] = kstep<PREC>(A[g & 1][j][b], &inA[t][4 * (ks < KSA ? ks : 0)], acc[b][t]);
                        else if constexpr (NB > 0) acc[b][t] = kstep<PREC>(A[g & 1][j][b], &inB[t][4 * (ks >= KSA ? ks - KSA : 0)], acc[b][t]);
                    }
            }
        }
        // the next group's G*FBC ds_reads go out during the FIRST half of this group's MFMAs
        if (g + 1 < NG) {
#pragma unroll
            for (int j = 0; j < (G + 1) / 2; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, FBC * TILES * MPK, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * FBC, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Hidden layer: inputs = up to two register segments, output -> registers (next B operand).
template <int PREC, int TILES, class CTX, int KIND, int NA, int NB, int NFB_OUT, int MODE, int NOUT>
__device__ __forceinline__ void layer_regs(CTX& c, const uint32_t (&inA)[TILES][NA],
                                           const uint32_t (&inB)[TILES][NB > 0 ? NB : 1],
                                           uint32_t (&out)[TILES][NOUT])
{
    constexpr int RPB = PrecT<PREC>::RPB;
    constexpr int KS = NA / 4 + NB / 4;
    constexpr int FBC0 = pnr_layer_fbc(KIND, PREC);
    constexpr int FBC = (NFB_OUT % FBC0 == 0) ? FBC0 : 1;      // must mirror pnr_build_plan
    // read-ahead depth in k-steps: the window costs 2*G*FBC*4 registers
    constexpr int G = (CTX::GDB / FBC) < 1 ? 1 : (CTX::GDB / FBC);
    static_assert(NOUT >= NFB_OUT * RPB, "output register array too small");
#pragma unroll
    for (int cb = 0; cb < NFB_OUT / FBC; ++cb) {
        c.begin();
        const char* base = c.base();
        const char* frag = base + c.lane * 16;
        f32x16 acc[FBC][TILES];
#pragma unroll
        for (int b = 0; b < FBC; ++b) {
            load_bias(base + FBC * KS * PNR_FRAG_BYTES + b * 128, c.hi, acc[b][0]);
#pragma unroll
            for (int t = 1; t < TILES; ++t) acc[b][t] = acc[b][0];
        }
        mma_chunk<PREC, TILES, FBC, G, NA, NB>(frag, inA, inB, acc);
        c.reads_done();
#pragma unroll
        for (int b = 0; b < FBC; ++b) {
            const int fb = cb * FBC + b;
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                if constexpr (PREC == PNR_PREC_BF16) {
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        float lo = acc[b][t][2 * p], hi = acc[b][t][2 * p + 1];
                        if (MODE == MODE_RELU) { lo = fmaxf(lo, 0.0f); hi = fmaxf(hi, 0.0f); }
                        out[t][fb * RPB + p] = pack_bf16(lo, hi);
                    }
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
        c.finish();
    }
}

// Output layer: rows [0, n_out) are stored to raw channels ch_base + row.
template <int PREC, int TILES, class CTX, int NA, int NB>
__device__ __forceinline__ void layer_out(CTX& c, const uint32_t (&inA)[TILES][NA],
                                          const uint32_t (&inB)[TILES][NB > 0 ? NB : 1], int n_out, int ch_base,
                                          const int (&samp)[TILES])
{
    constexpr int KSA = NA / 4, KSB = NB / 4;
    const int nfb = (n_out + 31) >> 5;
#pragma unroll 1
    for (int fb = 0; fb < nfb; ++fb) {
        c.begin();
        const char* base = c.base();
        const char* frag = base + c.lane * 16;
        f32x16 acc1[1][TILES];
        load_bias(base + (KSA + KSB) * PNR_FRAG_BYTES, c.hi, acc1[0][0]);
#pragma unroll
        for (int t = 1; t < TILES; ++t) acc1[0][t] = acc1[0][0];
        mma_chunk<PREC, TILES, 1, CTX::GDB, NA, NB>(frag, inA, inB, acc1);
        f32x16 (&acc)[TILES] = acc1[0];
        c.reads_done();
#if PNR_OPT_STORE_AFTER_BARRIER
        // Chunk hand-over first, stores second: finish()'s vmcnt(0) must cover only the LDS-DMA issued a
        // chunk ago, not the raw stores below (an HBM write round trip per output block otherwise).
        c.finish();
#endif
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            if (samp[t] >= 0) {
                float* dst = c.a.raw + (int64_t)samp[t] * c.a.ss;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = fb * 32 + (r & 3) + 8 * (r >> 2) + 4 * c.hi;
                    if (row < n_out) dst[(int64_t)(ch_base + row) * c.a.sc] = acc[t][r];
                }
            }
        }
#if !PNR_OPT_STORE_AFTER_BARRIER
        c.finish();
#endif
    }
}

// gamma() of one 3-vector into this lane's share of the lane vector (pnr_mlp_layout.h).
// NF = frequency bands per half-wave (5 for xyz, 2 for view directions); NV = values per lane.
template <int PREC, int NF, int NV, int NREG>
__device__ __forceinline__ void embed_lane(float p0, float p1, float p2, int hi, uint32_t (&out)[NREG])
{
    float v[NV];
    v[0] = hi ? p2 : p0;
    v[1] = hi ? 0.0f : p1;
    if constexpr (PREC == PNR_PREC_BF16 && PNR_OPT_FAST_EMBED) {
        // One accurate sincos per coordinate at this half-wave's lowest band, then the double-angle
        // recurrences sin 2t = 2 s c, cos 2t = 1 - 2 s^2 for the NF-1 higher bands: the error doubles per
        // octave (<= 2^(NF-1) * 1e-7 ~ 2e-6), far below the bf16 rounding (4e-3) applied next.
        const float base = hi ? (float)(1 << NF) : 1.0f;
        const float pp[3] = {p0, p1, p2};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float s, co;
            sincosf(pp[a] * base, &s, &co);
            v[2 + a] = s; v[2 + 3 + a] = co;
#pragma unroll
            for (int fp = 1; fp < NF; ++fp) {
                const float s2 = 2.0f * s * co, c2 = fmaf(-2.0f * s, s, 1.0f);
                s = s2; co = c2;
                v[2 + 6 * fp + a] = s; v[2 + 6 * fp + 3 + a] = co;
            }
        }
    } else {
#pragma unroll
        for (int fp = 0; fp < NF; ++fp) {
            const float sc = hi ? (float)(1 << (NF + fp)) : (float)(1 << fp);
            float s, co;
            sincosf(p0 * sc, &s, &co); v[2 + 6 * fp + 0] = s; v[2 + 6 * fp + 3] = co;
            sincosf(p1 * sc, &s, &co); v[2 + 6 * fp + 1] = s; v[2 + 6 * fp + 4] = co;
            sincosf(p2 * sc, &s, &co); v[2 + 6 * fp + 2] = s; v[2 + 6 * fp + 5] = co;
        }
    }
#pragma unroll
    for (int i = 2 + 6 * NF; i < NV; ++i) v[i] = 0.0f;
    if constexpr (PREC == PNR_PREC_BF16) {
#pragma unroll
        for (int p = 0; p < NV / 2; ++p) out[p] = pack_bf16(v[2 * p], v[2 * p + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) out[i] = __float_as_uint(v[i]);
    }
}

// Ring geometry: bf16 chunks are <= 25 KiB, fp32 <= 49 KiB; LDS is 160 KiB per CU.
//   8 waves (1 workgroup/CU) or 1 wave/SIMD: 5 slots, DMA 2 chunks ahead (waves may drift 2 chunks);
//   4 waves at 2 workgroups/CU, fp32, or the barrier scheme: fewer slots, DMA 1 ahead.
template <int PREC, int WAVES, int MINW>
struct Ring {
    static constexpr bool FLOW = PNR_FLOW != 0;
    static constexpr int NS = !FLOW ? 2 : 3;      // bf16 chunks are <= 41 KiB, fp32 <= 49 KiB; LDS is 160 KiB
    static constexpr int DIST = 1;
    static constexpr int CNT_BYTES = 64;       // ready[NS] | done[NS], in front of the slots
};

template <int PREC, int W, int TILES, int WAVES, int MINW>
__global__ __launch_bounds__(64 * WAVES, MINW) void k_mlp_fused(const MlpArgs a)
{
    using RG = Ring<PREC, WAVES, MINW>;
    constexpr int GDB = MINW >= 2 ? 4 : 8;   // deeper read-ahead when a wave is alone on its SIMD
    using CTX = Ctx<WAVES, RG::NS, RG::DIST, (PREC == PNR_PREC_BF16 ? GDB : 4), RG::FLOW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RPB = PrecT<PREC>::RPB;
    constexpr int NFB = W / 32, HFB = W / 64;
    constexpr int HR = NFB * RPB, GR = HFB * RPB;
    constexpr int GXR = PREC == PNR_PREC_BF16 ? 16 : 32;
    constexpr int GDR = PREC == PNR_PREC_BF16 ? 8 : 16;

    CTX c{a, smem + RG::CNT_BYTES, (volatile lds_u32*)smem, (int)(threadIdx.x & 63),
          __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), (int)((threadIdx.x & 63) >> 5),
          0, 0, 0, 0, 0, -1, {0, 0}, {0, 0}, 0};
    const int n = c.lane & 31;
    c.start();

    uint32_t dummy[TILES][1];
#pragma unroll
    for (int t = 0; t < TILES; ++t) dummy[t][0] = 0;

    // per-sample inputs of one tile: o(3) dx | dy dz (near far unused) | z
    struct SampleIn { float4 o4, d4; float zz; };
    auto fetch = [&](int grp, int t) {
        const int s = ((grp * WAVES + c.wave) * TILES + t) * 32 + n;
        const int sl = s < a.S ? s : a.S - 1;
        const int ray = sl / a.N;
        SampleIn in;
        in.o4 = *reinterpret_cast<const float4*>(a.rays + (int64_t)ray * 8);
        in.d4 = *reinterpret_cast<const float4*>(a.rays + (int64_t)ray * 8 + 4);
        in.zz = a.z[sl];
        return in;
    };
    SampleIn nextin[TILES];
#if PNR_OPT_PREFETCH_INPUTS
#pragma unroll
    for (int t = 0; t < TILES; ++t) nextin[t] = fetch(blockIdx.x < a.n_groups ? blockIdx.x : 0, t);
#endif

    for (int grp = blockIdx.x; grp < a.n_groups; grp += gridDim.x) {
        int samp[TILES];
        float vd[TILES][3];
        uint32_t ex[TILES][GXR];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int s = ((grp * WAVES + c.wave) * TILES + t) * 32 + n;
            samp[t] = s < a.S ? s : -1;
#if PNR_OPT_PREFETCH_INPUTS
            const SampleIn in = nextin[t];
#else
            const SampleIn in = fetch(grp, t);
#endif
            const float4 o4 = in.o4, d4 = in.d4;
            const float zz = in.zz;
            const float dx = o4.w, dy = d4.x, dz = d4.y;
            // pts = o + d*z: separate multiply and add, as the sampler's pnr_points does
            const float px = __fadd_rn(o4.x, __fmul_rn(dx, zz));
            const float py = __fadd_rn(o4.y, __fmul_rn(dy, zz));
            const float pz = __fadd_rn(o4.z, __fmul_rn(dz, zz));
            const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)));
            vd[t][0] = dx / nrm; vd[t][1] = dy / nrm; vd[t][2] = dz / nrm;
            embed_lane<PREC, 5, 32, GXR>(px, py, pz, c.hi, ex[t]);
        }

        // Trunk.  The activations ping-pong between two register arrays, two layers per loop trip, so
        // that no layer ends with a 64-register copy (a measured ~2k cycles per layer when the loop
        // carried `cur = nxt`).  Layer 0 writes whichever array makes the LAST trunk layer land in `cur`.
        uint32_t cur[TILES][HR], nxt[TILES][HR];
        auto trunk = [&](int l, const uint32_t (&src)[TILES][HR], uint32_t (&dst)[TILES][HR]) {
            if (l - 1 == a.skip)
                layer_regs<PREC, TILES, CTX, PNR_L_TRUNK, GXR, HR, NFB, MODE_RELU, HR>(c, ex, src, dst);
            else
                layer_regs<PREC, TILES, CTX, PNR_L_TRUNK, HR, 0, NFB, MODE_RELU, HR>(c, src, dummy, dst);
        };
#if PNR_OPT_PINGPONG
        int l = 1;
        if ((a.D - 1) & 1) {
            layer_regs<PREC, TILES, CTX, PNR_L_TRUNK0, GXR, 0, NFB, MODE_RELU, HR>(c, ex, dummy, nxt);
            trunk(1, nxt, cur);
            l = 2;
        } else {
            layer_regs<PREC, TILES, CTX, PNR_L_TRUNK0, GXR, 0, NFB, MODE_RELU, HR>(c, ex, dummy, cur);
        }
#pragma unroll 1
        for (; l < a.D; l += 2) {
            trunk(l, cur, nxt);
            trunk(l + 1, nxt, cur);
        }
#else
        layer_regs<PREC, TILES, CTX, PNR_L_TRUNK0, GXR, 0, NFB, MODE_RELU, HR>(c, ex, dummy, cur);
#pragma unroll 1
        for (int l = 1; l < a.D; ++l) {
            trunk(l, cur, nxt);
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int i = 0; i < HR; ++i) cur[t][i] = nxt[t][i];
        }
#endif
        if (a.n_sem) {
            uint32_t sh[TILES][GR];
            layer_regs<PREC, TILES, CTX, PNR_L_SEM0, HR, 0, HFB, MODE_RELU, GR>(c, cur, dummy, sh);
            layer_out<PREC, TILES, CTX, GR, 0>(c, sh, dummy, a.n_sem, 4, samp);
        }
        if (a.n_inst) {
            uint32_t sh[TILES][GR];
            layer_regs<PREC, TILES, CTX, PNR_L_INST0, HR, 0, HFB, MODE_RELU, GR>(c, cur, dummy, sh);
            layer_out<PREC, TILES, CTX, GR, 0>(c, sh, dummy, a.n_inst, 4 + a.n_sem, samp);
        }
#if PNR_OPT_PREFETCH_INPUTS
        // next sample group's inputs: issued here so their HBM latency hides under the feature/views layers
        {
            const int g2 = grp + (int)gridDim.x < a.n_groups ? grp + (int)gridDim.x : grp;
#pragma unroll
            for (int t = 0; t < TILES; ++t) nextin[t] = fetch(g2, t);
        }
#endif
        layer_regs<PREC, TILES, CTX, PNR_L_FEATURE, HR, 0, NFB, MODE_LINEAR, HR>(c, cur, dummy, nxt);
        uint32_t ed[TILES][GDR];
#pragma unroll
        for (int t = 0; t < TILES; ++t) embed_lane<PREC, 2, 16, GDR>(vd[t][0], vd[t][1], vd[t][2], c.hi, ed[t]);
        uint32_t g[TILES][GR];
        layer_regs<PREC, TILES, CTX, PNR_L_VIEWS, HR, GDR, HFB, MODE_RELU, GR>(c, nxt, ed, g);
        layer_out<PREC, TILES, CTX, GR, HR>(c, g, cur, 4, 0, samp);
        ++c.iter;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ring refills issued past the last chunk
}

// ------------------------------------------------------------------------------- launcher
template <int PREC, int W, int TILES, int WAVES, int MINW>
static int launch_mlp(const MlpArgs& a0, hipStream_t stream)
{
    MlpArgs a = a0;
    using RG = Ring<PREC, WAVES, MINW>;
    const int lds_bytes = RG::CNT_BYTES + RG::NS * a.slot_bytes;
    PNR_REQUIRE(lds_bytes <= 163840, "pnr_mlp_forward: weight ring of %d bytes exceeds the 160 KiB LDS", lds_bytes);
    PNR_REQUIRE(a.n_chunks >= RG::DIST + 2, "pnr_mlp_forward: network too small for the weight ring");
    const int per_group = 32 * TILES * WAVES;
    a.n_groups = (a.S + per_group - 1) / per_group;
    auto kern = k_mlp_fused<PREC, W, TILES, WAVES, MINW>;
    static thread_local int wg_per_cu = 0;
    if (wg_per_cu == 0) {
        PNR_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        int nb = 0;
        PNR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 64 * WAVES, lds_bytes));
        wg_per_cu = nb < 1 ? 1 : (nb > 4 ? 4 : nb);
    }
    // persistent grid: every resident workgroup slot of the 256 CUs, grid-stride over sample groups
    const int cap = 256 * wg_per_cu;
    const int grid = a.n_groups < cap ? a.n_groups : cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds_bytes, stream, a);
    PNR_CHECK_LAUNCH("pnr_mlp_forward");
    return PNR_OK;
}

// Kernel variant for the bf16 path, selectable at run time for A/B measurements
// (PNR_MLP_VARIANT): 0 = 1 tile/wave, 4 waves, 1 wave/SIMD;  1 = 1 tile/wave, 4 waves,
// registers capped for 2 workgroups per CU;  2 = 2 tiles/wave, 4 waves;  3 = 1 tile/wave, 8 waves.
static int mlp_variant()
{
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("PNR_MLP_VARIANT");
        v = e ? atoi(e) : PNR_MLP_DEFAULT_VARIANT;
        if (v < 0 || v > 3) v = PNR_MLP_DEFAULT_VARIANT;
    }
    return v;
}

template <int W>
static int launch_bf16(const MlpArgs& a, hipStream_t st)
{
    switch (mlp_variant()) {
    case 1: return launch_mlp<PNR_PREC_BF16, W, 1, 4, 2>(a, st);
    case 2: return launch_mlp<PNR_PREC_BF16, W, 2, 4, 1>(a, st);
    case 3: return launch_mlp<PNR_PREC_BF16, W, 1, 8, 2>(a, st);
    default: return launch_mlp<PNR_PREC_BF16, W, 1, 4, 1>(a, st);
    }
}

PNR_EXPORT int pnr_mlp_forward(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z,
                               int64_t n_rays, int n_samples, float* raw, int64_t raw_stride_s,
                               int64_t raw_stride_c, void* stream)
{
    int rc = pnr_mlp_validate(desc);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(n_rays >= 0 && n_samples >= 1, "pnr_mlp_forward: bad size");
    if (n_rays == 0) return PNR_OK;
    PNR_REQUIRE(packed && rays && z && raw, "pnr_mlp_forward: null pointer");
    PNR_REQUIRE(n_rays * (int64_t)n_samples < ((int64_t)1 << 31) - 4096, "pnr_mlp_forward: R*N=%lld exceeds 2^31",
                (long long)(n_rays * n_samples));
    PNR_REQUIRE((((uintptr_t)rays) & 15) == 0 && (((uintptr_t)packed) & 15) == 0,
                "pnr_mlp_forward: rays / packed must be 16-byte aligned");
    PnrPlan plan;
    pnr_build_plan(*desc, plan);
    MlpArgs a;
    a.data = (const uint8_t*)packed + plan.data_off;
    a.table = (const pnr_chunk_entry*)((const uint8_t*)packed + plan.table_off);
    a.n_chunks = (int)plan.chunks.size();
    a.slot_bytes = plan.max_chunk_frags * PNR_FRAG_BYTES;
    a.rays = rays; a.z = z; a.S = (int)(n_rays * n_samples); a.N = n_samples; a.n_groups = 0;
    a.raw = raw; a.ss = raw_stride_s; a.sc = raw_stride_c;
    a.D = desc->D; a.skip = desc->skip; a.n_sem = desc->n_sem; a.n_inst = desc->n_inst;
    a.trace = nullptr; a.trace_wave = 0;
#if PNR_TRACE
    if (const char* e = getenv("PNR_TRACE_PTR")) a.trace = (unsigned long long*)strtoull(e, nullptr, 0);
    if (const char* e = getenv("PNR_TRACE_WAVE")) a.trace_wave = atoi(e);
#endif
    hipStream_t st = (hipStream_t)stream;
    if (desc->precision == PNR_PREC_BF16) return desc->W == 256 ? launch_bf16<256>(a, st) : launch_bf16<128>(a, st);
    if (desc->W == 256) return launch_mlp<PNR_PREC_FP32, 256, 1, 4, 1>(a, st);
    return launch_mlp<PNR_PREC_FP32, 128, 1, 4, 1>(a, st);
}

PNR_EXPORT int pnr_time_mlp_forward(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z,
                                    int64_t n_rays, int n_samples, float* raw, int64_t raw_stride_s,
                                    int64_t raw_stride_c, int iters, float* ms_out_host, void* stream)
{
    PNR_REQUIRE(iters >= 1 && ms_out_host, "pnr_time_mlp_forward: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    PNR_HIP(hipEventCreate(&e0));
    PNR_HIP(hipEventCreate(&e1));
    PNR_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) {
        int rc = pnr_mlp_forward(desc, packed, rays, z, n_rays, n_samples, raw, raw_stride_s, raw_stride_c, stream);
        if (rc != PNR_OK) return rc;
    }
    PNR_HIP(hipEventRecord(e1, st));
    PNR_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    PNR_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out_host = ms / (float)iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return PNR_OK;
}
