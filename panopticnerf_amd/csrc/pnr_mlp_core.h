// Device-side core shared by the fused MLP forward (pnr_mlp.hip) and backward (pnr_mlp_bwd.hip)
// kernels: MFMA k-step, the double-buffered LDS weight stream, and the chunk MFMA loop.
// See pnr_mlp.hip's header for the design and pnr_mlp_layout.h for the packed layout.
#pragma once
#include "pnr_common.h"
#include "pnr_mlp_layout.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) void lds_void;
// The chunk table is read through the constant address space so that hipcc emits scalar
// (s_load) instead of vector loads: a vector load here costs an L2 round trip AND a vmcnt(0)
// that drains the in-flight LDS-DMA, once per chunk.
typedef const __attribute__((address_space(4))) pnr_chunk_entry* table_ptr;

struct MlpArgs {
    const uint8_t* data;            // fragment stream (device)
    const pnr_chunk_entry* table;   // chunk table (device)
    int n_chunks, slot_bytes;
    const float* rays; const float* z;
    int S, N, n_groups;
    uint32_t n_magic; int n_shift;  // x / N for 0 <= x < 2^31 as (x * n_magic) >> n_shift (pnr_set_div_magic)
    float* raw; int64_t ss, sc;
    int D, skip, n_sem, n_inst;
    int head_tap, head_depth;       // pnr_mlp_desc: 0 / 1 = the heads read h / the feature; 1 / 2 Linear layers per head
    // training only (null otherwise).  acts: activations saved by the forward for the backward,
    // bf16, one [S][width] slot-ordered region per tensor (pnr_train_layout); d_raw: upstream gradient
    // of raw, (ch, S) channel-major fp32; dys: pre-activation gradients written by the backward.
    uint16_t* acts; const float* d_raw; uint16_t* dys;
    // fused compositing epilogue (inference, k_mlp_pp<.., FUSE>; pnr_mlp_fuse.h): one record of `rec_floats` floats per 32-sample
    // tile and one (lw, r, g, b) quadruple per sample
    float* rec; int rec_floats; float4* ps;
    unsigned long long* trace;      // PNR_TRACE builds: [8 waves][PNR_TRACE_CHUNKS][PNR_TRACE_STAMPS]
    unsigned long long* clk;        // optional (bench): {shader cycles, 100 MHz ticks} of workgroup 0's first wave
    int64_t acts_off[24], dys_off[24], gate_off[24];     // pnr_train_layout (bf16 units)
};

// x / d for 0 <= x < 2^31, d >= 1, without an integer division on the device: m = ceil(2^(31+s) / d), s = ceil(log2 d);
// then m * d - 2^(31+s) <= 2^s, which makes floor(x * m / 2^(31+s)) exact for every x < 2^31 (Granlund-Montgomery).
static inline void pnr_set_div_magic(int d, uint32_t& magic, int& shift)
{
    int s = 0;
    while ((1ll << s) < (long long)d) ++s;
    shift = 31 + s;
    magic = (uint32_t)(((1ull << shift) + (unsigned long long)d - 1) / (unsigned long long)d);
}
__device__ __forceinline__ int pnr_div_magic(int x, uint32_t magic, int shift)
{
    return (int)(((unsigned long long)(uint32_t)x * magic) >> shift);
}

#ifndef PNR_ABL_STORE
#define PNR_ABL_STORE 0
#endif
#ifndef PNR_ABL_NODMA
#define PNR_ABL_NODMA 0
#endif
#ifndef PNR_TRAIN_FWD_ISSUERS
#define PNR_TRAIN_FWD_ISSUERS 4 /* waves that copy the weight pieces in the TRAINING forward (Ctx::issue); 8 = every wave (round 3) */
#endif
#ifndef PNR_CORE_DMA_AUX
#define PNR_CORE_DMA_AUX 0      /* cache policy of the lock-step kernels' weight pieces (training forward, data-gradient pass): nt, which the
                                   inference kernel gains 0.9 % from, costs the training forward +15 % (1.58 -> 1.82 ms) -- its write stream
                                   sweeps the L2 and non-temporal weight lines go first */
#endif
// One 32-row block's share of lane (n, hi) -- 16 slots = chunks fb*4 + hi*2 + {0, 1} -- of padded sample row s into a
// saved region (pnr_mlp_layout.h: the two chunks sit in neighbouring lines at the same position).  Unmasked: rows
// S..S_pad are written as well.
__device__ __forceinline__ void store_slots(uint16_t* base, int width, int s, int fb, int hi, const uint32_t* r8)
{
#if PNR_ABL_STORE == 3          /* ablation: every workgroup writes the same 256 rows (L2-resident target, results invalid) */
    s &= 255;
#endif
    u32x4* p = reinterpret_cast<u32x4*>(base + pnr_saved_chunk(width >> 3, s, fb * 4 + hi * 2));
    u32x4 a, b;
    a[0] = r8[0]; a[1] = r8[1]; a[2] = r8[2]; a[3] = r8[3];
    b[0] = r8[4]; b[1] = r8[5]; b[2] = r8[6]; b[3] = r8[7];
#if PNR_ABL_STORE == 4 || PNR_ABL_STORE == 5   /* ablation (results invalid): 4 = waves 0-3 (the piece issuers of the training forward) do not
                                                  store, 5 = waves 4-7 do not: is it the ISSUERS' stores that the piece waits hang on? */
    if (((int)(threadIdx.x >> 6) < 4) == (PNR_ABL_STORE == 4)) { asm volatile("" :: "v"(a), "v"(b), "v"(p)); return; }
    p[0] = a; p[8] = b;
#elif PNR_ABL_STORE == 1          /* ablation: no activation / gradient stores (results invalid) */
    asm volatile("" :: "v"(a), "v"(b), "v"(p));
#elif PNR_ABL_STORE == 2        /* nontemporal: 1.8 -> 3.0 ms (sc0 / sc1 / sc0 sc1 scopes: +-0 / +-0 / +8 %) */
    __builtin_nontemporal_store(a, p); __builtin_nontemporal_store(b, p + 8);
#else
    p[0] = a; p[8] = b;
#endif
}

// gamma(x) of lane (n, hi): 32 slots hi*32 + v = chunks hi*4 + k of a 64-slot saved region
template <int NR>
__device__ __forceinline__ void store_ex(uint16_t* base, int s, int hi, const uint32_t (&ex)[NR])
{
    static_assert(NR == 16, "bf16 gamma(x): 16 packed registers per lane");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        u32x4 v; v[0] = ex[4 * k]; v[1] = ex[4 * k + 1]; v[2] = ex[4 * k + 2]; v[3] = ex[4 * k + 3];
        *reinterpret_cast<u32x4*>(base + pnr_saved_chunk(8, s, hi * 4 + k)) = v;
    }
}

// ReLU gate bits of one 32-row block (8 packed bf16x2 registers, post-ReLU) into the lane's gate word of the block PAIR:
// min(x, 1) per half leaves bit 0 / bit 16, shifted to bit 8*(fb&1) + p / 16 + 8*(fb&1) + p (pnr_train_layout).
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
__device__ __forceinline__ uint32_t gate_bits_or(uint32_t word, const uint32_t* out8, int fb)
{
    const u16x2 one = {1, 1};
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint32_t y = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, out8[p]), one));
        word |= y << (8 * (fb & 1) + p);
    }
    return word;
}
// this lane's NW = width/64 gate dwords of sample s in a gate region of `width` features
template <int NW>
__device__ __forceinline__ void store_gates(uint16_t* region, int width, int s, int hi, const uint32_t (&w)[NW])
{
    uint32_t* p = reinterpret_cast<uint32_t*>(region)      // unmasked: padded sample rows S..S_pad are written too (the whole buffer is defined)
        + (size_t)s * (width / 32) + hi * NW;
    if constexpr (NW == 4) { u32x4 v; v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3]; *reinterpret_cast<u32x4*>(p) = v; }
    else {
#pragma unroll
        for (int i = 0; i < NW; ++i) p[i] = w[i];
    }
}
template <int NW>
__device__ __forceinline__ void load_gates(const uint16_t* region, int width, int s, int hi, uint32_t (&w)[NW])
{
    const uint32_t* p = reinterpret_cast<const uint32_t*>(region) + (size_t)(s < 0 ? 0 : s) * (width / 32) + hi * NW;
    if constexpr (NW == 4) { const u32x4 v = *reinterpret_cast<const u32x4*>(p); w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; w[3] = v[3]; }
    else {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = p[i];
    }
}
// gate bits of a whole (packed bf16, post-ReLU) layer output of NR = 8 * blocks registers
template <int NR>
__device__ __forceinline__ void save_gates(uint16_t* region, int s, int hi, const uint32_t (&regs)[NR])
{
    constexpr int NW = NR / 16;
    uint32_t w[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) w[j] = 0;
#pragma unroll
    for (int fb = 0; fb < NR / 8; ++fb) w[fb / 2] = gate_bits_or(w[fb / 2], &regs[fb * 8], fb);
    store_gates<NW>(region, NR * 4, s, hi, w);
}
// packed bf16x2 gradient gated by the two gate bits of (block fb, register p) of gate word g: each half times its bit
// (shift, and, v_pk_mul_lo_u16 -- two instructions fewer per register than building an and-mask from sign-extended bits)
__device__ __forceinline__ uint32_t gate_apply(uint32_t packed, uint32_t g, int fb, int p)
{
    const uint32_t bits = (g >> (8 * (fb & 1) + p)) & 0x00010001u;
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, packed) * __builtin_bit_cast(u16x2, bits));
}

enum { MODE_RELU = 0, MODE_LINEAR = 1 };

template <int PREC> struct PrecT;
template <> struct PrecT<PNR_PREC_BF16> { static constexpr int RPB = 8; };    // B regs per 32 input features
template <> struct PrecT<PNR_PREC_FP32> { static constexpr int RPB = 16; };

// Two fp32 -> packed bf16x2 (RNE), lo in bits [15:0].  Written as ONE 2-vector conversion so that hipcc selects a
// single v_cvt_pk_bf16_f32; two scalar (__bf16) casts become two half-empty v_cvt_pk_bf16_f32 plus a v_perm_b32.
// (Not inline asm: asm reading MFMA results bypasses the compiler's MFMA->VALU hazard handling.)
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi)
{
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
}

typedef __attribute__((ext_vector_type(2))) short i16x2;
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t v)
{
    const i16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, v), z));
}

// One k-step: 16 bytes of A per lane against 4 B registers.
template <int PREC>
__device__ __forceinline__ f32x16 kstep(const u32x4& a, const uint32_t* b, f32x16 acc)
{
    if constexpr (PREC == PNR_PREC_BF16) {
        u32x4 bv;
        bv[0] = b[0]; bv[1] = b[1]; bv[2] = b[2]; bv[3] = b[3];
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

// PNR_TRACE (debug builds only): per-chunk s_memtime stamps of every wave of workgroup 0, kept in LDS (no
// vector-memory traffic, so vmcnt waits are not perturbed) and dumped to MlpArgs::trace at kernel end.
#ifndef PNR_TRACE
#define PNR_TRACE 0
#endif
#ifndef PNR_TRACE_MASK
#define PNR_TRACE_MASK 0xff
#endif
#ifndef PNR_TRACE_WG
#define PNR_TRACE_WG 0          /* the traced workgroup and its traced iteration (sample group) */
#endif
#ifndef PNR_TRACE_ITER
#define PNR_TRACE_ITER 2
#endif
#define PNR_TRACE_CHUNKS 48
#define PNR_TRACE_STAMPS 8

// ---- weight stream: NSLOT LDS slots; chunk c + NSLOT - 1 is copied in (LDS-DMA) while chunk c feeds the MFMAs.
// Two slots everywhere.  Three (two chunks ahead) were tried for the data-gradient pass on the suspicion that its chunk
// hand-overs wait for weight pieces delayed behind the gradient stores: +-0 (profiles/README.md, round 2 training notes).
template <int WAVES, int GDB_, int NSLOT = 2, int ISSUERS_ = 0>
struct Ctx {
    static constexpr int GDB = GDB_;   // A-fragment read-ahead (fragments per tile in flight)
    static constexpr int DIST = NSLOT - 1;
    const MlpArgs& a;
    char* smem;
    int lane, wave, hi;
    int ci, slot;
    pnr_chunk_entry e1, e2;            // table entries of chunks ci+DIST, ci+DIST+1 (scalar loads, fetched a chunk early)
    bool st_full = false;
    int n_last = 0;                    // NSLOT = 3: this wave's LDS-DMA pieces of the chunk requested in begin()              // every lane of this wave holds a valid sample: its store count per chunk is exact
#if PNR_TRACE
    unsigned long long* tr;            // LDS trace area of this wave: [PNR_TRACE_CHUNKS][PNR_TRACE_STAMPS]
    int titer;
    __device__ __forceinline__ void stamp(int k)
    {
        if (!((PNR_TRACE_MASK >> k) & 1)) return;       // single-stamp builds: a stamp costs an lgkmcnt(0) wait
        if (blockIdx.x == PNR_TRACE_WG && titer == PNR_TRACE_ITER && ci < PNR_TRACE_CHUNKS) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) tr[ci * PNR_TRACE_STAMPS + k] = t;
        }
    }
#else
    __device__ __forceinline__ void stamp(int) {}
#endif

    __device__ __forceinline__ int wrap(int i) const { return i >= a.n_chunks ? i - a.n_chunks : i; }
    __device__ __forceinline__ pnr_chunk_entry entry(int idx) const
    {
        table_ptr t = (table_ptr)(uintptr_t)a.table;
        pnr_chunk_entry e;
        e.off_frag = t[idx].off_frag;
        e.nfrag = t[idx].nfrag;
        return e;
    }
    // L2 -> LDS copy of a chunk into slot `sl` (asynchronous LDS-DMA, 1 KiB per wave-instruction)
    // ISSUERS_ > 0: only waves < ISSUERS_ copy weight pieces.  vmcnt is ONE in-order counter for LDS-DMA loads and stores, so a
    // wave that waits for its pieces also waits for every older store of its own: with all waves issuing pieces, each wave keeps
    // at most ~2 chunks of activation stores in flight.  A wave that never issues a piece never has to wait on vmcnt at a chunk
    // hand-over (the pieces are covered by their issuers' waits plus the barrier), so its stores stay in flight as long as they
    // need.  Training forward, 4 of 8 waves (same box, tools/train_kernels_time.py, outputs bit-identical): 1.3205 -> 1.2681 ms at 786 K
    // samples (-4.0 %), 0.450 -> 0.434 at 262 K; 2 of 8: -1.9 % (the two issuers' own 16 pieces per chunk get long).  The
    // data-gradient pass loads gate words in every layer -- the compiler's wait for those loads drains the stores anyway -- and
    // measured slower with 4 issuers: it keeps all 8.
    static constexpr int ISSUERS = (ISSUERS_ > 0 && ISSUERS_ < WAVES) ? ISSUERS_ : WAVES;
    __device__ __forceinline__ void issue(const pnr_chunk_entry& e, int sl) const
    {
        const uint8_t* src = a.data + (size_t)e.off_frag * PNR_FRAG_BYTES;
        char* dst = smem + sl * a.slot_bytes;
        if (ISSUERS < WAVES && wave >= ISSUERS) return;
        for (int f = wave; f < (int)e.nfrag; f += ISSUERS) pnr_dma_piece<PNR_CORE_DMA_AUX>(src + (size_t)f * PNR_FRAG_BYTES, dst + f * PNR_FRAG_BYTES, lane * 16);
    }
    __device__ __forceinline__ void start()
    {
        ci = 0; slot = 0;
#pragma unroll
        for (int k = 0; k < DIST; ++k) issue(entry(wrap(k)), k);
        e1 = entry(wrap(DIST));
        e2 = entry(wrap(wrap(DIST) + 1));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    __device__ __forceinline__ const char* base() const { return smem + slot * a.slot_bytes; }
    __device__ __forceinline__ void begin()
    {
        stamp(0);
        int sl = slot + DIST;
        if (sl >= NSLOT) sl -= NSLOT;
#if !PNR_ABL_NODMA              /* ablation: the weight stream stops after start() (results invalid) */
        issue(e1, sl);
#endif
        if constexpr (NSLOT > 2) n_last = (wave < ISSUERS && wave < (int)e1.nfrag) ? ((int)e1.nfrag - wave + ISSUERS - 1) / ISSUERS : 0;
        stamp(1);
    }
    // Chunk hand-over: this wave's share of the next chunk has landed, every wave is done reading this one.
    // nst: activation / gradient store INSTRUCTIONS this wave issued since begin().  Vector-memory operations of a wave
    // complete in order, so "at most nst outstanding" means the LDS-DMA pieces (older) have landed while the stores
    // drain under the next chunk's MFMAs -- without it every chunk pays an HBM write acknowledgement (measured: 0.6 ms
    // of the 1.8 ms training forward, 0.7 ms of the data-gradient pass).  A wave with invalid lanes may have skipped
    // stores (execz), so it waits for everything.
    __device__ __forceinline__ void finish(int nst = 0)
    {
        stamp(4);
        if constexpr (NSLOT > 2) {
            // chunk ci+1 must have landed; the pieces of chunk ci+2 (requested in this chunk's begin()) and the stores may fly on
            const int allow = n_last + (st_full ? nst : 0);
#define PNR_VM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
            switch (allow) {
                PNR_VM(1) PNR_VM(2) PNR_VM(3) PNR_VM(4) PNR_VM(5) PNR_VM(6) PNR_VM(7) PNR_VM(8) PNR_VM(9) PNR_VM(10) PNR_VM(11) PNR_VM(12)
                PNR_VM(13) PNR_VM(14) PNR_VM(15) PNR_VM(16)
                default: if (allow > 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
#undef PNR_VM
        } else if (ISSUERS < WAVES && wave >= ISSUERS) {
            // this wave issued no piece: nothing of its own to wait for -- its stores fly on
        } else {
            if (st_full && nst == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (st_full && nst == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        stamp(5);
        __syncthreads();
        stamp(6);
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
        ci = wrap(ci + 1);
        e1 = e2;
        e2 = entry(wrap(wrap(wrap(ci + DIST)) + 1));
    }
};

__device__ __forceinline__ void load_bias(const char* bias, int hi, f32x16& acc)
{
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + (8 * m + 4 * hi) * 4);
        acc[4 * m + 0] = b[0]; acc[4 * m + 1] = b[1]; acc[4 * m + 2] = b[2]; acc[4 * m + 3] = b[3];
    }
}

// All MFMAs of one chunk: FBC output blocks x KS k-steps x TILES sample tiles.  The FBC blocks are
// FBC independent accumulator chains issued round-robin per k-step.  A fragments are read from LDS
// G k-steps ahead of their MFMAs; the sched_group_barrier sequence pins that interleave (hipcc
// otherwise either sinks every read to just before its use or hoists all of them to the chunk top,
// +64-96 VGPRs and spills at 2 waves/SIMD).
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

