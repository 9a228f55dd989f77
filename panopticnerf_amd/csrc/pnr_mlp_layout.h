// Layout of the packed NeRF-MLP parameter image shared by the host packer (pnr_mlp_pack.cpp)
// and the fused kernel (pnr_mlp.hip).  SURVEY.md 8a row a5 gives the network; this header
// fixes how it is laid out for v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32.
//
// The kernel evaluates every layer TRANSPOSED:  H_out^T (features x samples) = W * H_in^T,
// with W as the MFMA A operand and the activations as the B operand.  One wave owns 32
// samples per tile (B/C/D column n = lane & 31); the two half-waves hi = lane >> 5 split
// the k (input-feature) dimension.  The 32x32 accumulator holds, for lane (n, hi),
// rows row(r,hi) = (r&3) + 8*(r>>2) + 4*hi, r = 0..15 (cdna_hip_programming.md section 3) --
// exactly the shape a B operand needs, so a layer's output registers ARE the next layer's
// B operand and activations never leave the register file.  The price is that input feature
// order inside the k dimension is the accumulator's row order; the packer pays it once by
// permuting W's columns ("slot maps" below).
//
// Lane vector.  For a layer input of 2*VL k-slots each lane (n,hi) holds VL values V[v].
//   hidden features : V[v] = feature (v>>4)*32 + row(v&15, hi)           (VL = width/2)
//   gamma(x), 64 slots (VL=32): v=0: hi?z:x  v=1: hi?pad:y  v=2+6f'+j: freq f=5*hi+f',
//                               j = sin x,y,z, cos x,y,z  -> canonical column 3+6f+j
//   gamma(d), 32 slots (VL=16): v=0: hi?dz:dx v=1: hi?pad:dy v=2+6f'+j: f=2*hi+f' (f'<2); v=14,15 pad
// A "k-step" consumes KPL consecutive V values per lane = 4 B registers = one 16-byte A read:
//   bf16: KPL=8 (one 32x32x16 MFMA);  fp32: KPL=4 (four 32x32x2 MFMAs, one value each).
//
// Fragment = 1 KiB: lane l = (i = l&31, hi = l>>5) owns bytes [16 l, 16 l + 16): the KPL
// values W[fb*32 + i][col(hi, ks*KPL + j)], j < KPL, 0 where col is a pad or the row is
// beyond the layer's outputs.  A chunk = FBC consecutive 32-row output blocks of one layer
// (pnr_layer_fbc): block 0's k-step fragments in order, block 1's, ..., then ONE bias fragment
// (FBC x 32 fp32 biases, rest 0).  The kernel runs the FBC blocks of a chunk as FBC interleaved
// accumulator chains: an instruction issued between two MFMAs on the SAME accumulator costs
// ~43 cycles on gfx950, between different accumulators ~6 (MI355X_MICROARCH.md cycle table).
// The image is the chunks in execution order, preceded by a header and a chunk table.
#pragma once
#include <stdint.h>

#include "pnr.h"

#if defined(__HIPCC__)
#define PNR_HD __host__ __device__
#else
#define PNR_HD
#endif

#define PNR_PACK_MAGIC 0x504e5231u /* "PNR1" */
#define PNR_FRAG_BYTES 1024

struct pnr_pack_header {
    uint32_t magic, version;
    uint32_t n_chunks, max_chunk_frags;
    uint32_t table_off, data_off;   // bytes from image start
    uint64_t total_bytes;
    int32_t desc[16];               // copy of pnr_mlp_desc
    uint32_t pad[8];
};                                   // 128 bytes
struct pnr_chunk_entry { uint32_t off_frag, nfrag; };   // offset from data_off in fragments

PNR_HD static inline int pnr_row_of(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
PNR_HD static inline int pnr_kpl(int precision) { return precision == 0 ? 8 : 4; }

enum { PNR_SEG_GX = 0, PNR_SEG_GD = 1, PNR_SEG_FEAT = 2 };

// layer kinds in execution order
enum { PNR_L_TRUNK0 = 0, PNR_L_TRUNK, PNR_L_SEM0, PNR_L_SEM1, PNR_L_INST0, PNR_L_INST1, PNR_L_FEATURE, PNR_L_VIEWS,
       PNR_L_RGBSIGMA, PNR_L_LOGITS /* plan 1: semantic and instance logit layers as one chunk */ };

// 32-row output blocks per chunk.  fp32 (parity mode) keeps 1: its fragments are twice as many.
static constexpr int pnr_layer_fbc(int kind, int precision)
{
    if (precision != 0) return 1;
    switch (kind) {
    case PNR_L_TRUNK0: return 4;                       // 4 k-steps per block: 4 blocks make a 16-MFMA chunk
    // 2 blocks per hidden chunk: 4 (half the barriers, 65 KiB slots) measured 2 % slower
    case PNR_L_TRUNK: case PNR_L_FEATURE: case PNR_L_SEM0: case PNR_L_INST0: case PNR_L_VIEWS: return 2;
    default: return 1;                                 // output layers: run-time block count
    }
}

// canonical column (within the segment's own canonical vector) of lane-vector slot (hi, v); -1 = pad
PNR_HD static inline int pnr_seg_col(int kind, int L, int hi, int v)
{
    if (kind == PNR_SEG_FEAT) return (v >> 4) * 32 + pnr_row_of(v & 15, hi);
    const int nfh = (kind == PNR_SEG_GX) ? 5 : 2;   // frequencies per half-wave
    if (v == 0) return hi ? 2 : 0;
    if (v == 1) return hi ? -1 : 1;
    const int fp = (v - 2) / 6, j = (v - 2) % 6;
    if (fp >= nfh) return -1;
    const int f = hi * nfh + fp;
    if (f >= L) return -1;
    return 3 + 6 * f + j;
}
// values per lane of a segment
static inline int pnr_seg_vl(int kind, int nfeat) { return kind == PNR_SEG_GX ? 32 : kind == PNR_SEG_GD ? 16 : nfeat / 2; }

// Training buffers (bf16 elements; every region is a slot-ordered [S_pad][width] tensor in the SAVED-TENSOR layout below).
// Saved-tensor layout: the unit is the 16-byte chunk c = slot / 8 of sample s; a 128-byte line holds chunk c of the 8
// samples of group s >> 3, lines are ordered [s >> 3][c], and inside a line sample s sits at position
// (s & 7) ^ (4 * ((c >> 1) & 1)).  Why: (1) a wave's store instruction (32 samples x 2 half-waves, one chunk each) then
// writes 8 FULL lines instead of touching 32 -- at [S][width] rows the training forward and the data-gradient pass spent
// a third of their time in the texture-address unit (no-store ablation: 1.82 -> 1.25 ms and 1.81 -> 1.13 ms);
// (2) k_wgrad's LDS-DMA copies 1 KiB pieces verbatim (8 lines, contiguous on both sides) and its transposing reads
// (ds_read_b64_tr_b16: 16 lanes = 4 samples x 2 chunks) hit all 64 banks once per half-wave thanks to the XOR.
// S_pad = S rounded up to 256 (a workgroup's samples): the kernels write the padding rows too (dY: zeros).
//   acts_off: [0] EX (64)  [1] ED (32)  [2+l] X_{l+1}, l < D (W)  [2+D] F (W)  [3+D] G (W/2)
//             [4+D] SH_sem (W/2)  [5+D] SH_inst (W/2)  [6+D] total
//   dys_off : [0] DY_views (W/2)  [1] DY_feature (W)  [2] DY_sem0 (W/2)  [3] DY_inst0 (W/2)
//             [4+l] DY_l, l < D (W)  [4+D] total
// Slot order of a width-n feature tensor: slot fb*32 + hi*16 + r <-> feature fb*32 + row(r,hi); of EX: slot
// hi*32 + v <-> pnr_seg_col(GX, L, hi, v); of ED: slot hi*16 + v <-> pnr_seg_col(GD, L, hi, v).
//   gate_off (optional, same indexing as acts_off; entries of tensors without a ReLU are -1): ONE BIT per element of every
//             ReLU output -- X_1..X_D, G, SH_sem, SH_inst -- appended to the acts buffer (acts_off[6+D] = grand total).  The
//             data-gradient pass gates with these (32 B per sample and 256-wide layer) instead of re-reading the bf16
//             activations (512 B).  Per sample w/32 dwords: lane (n, hi) owns dwords hi*(w/64) .. ; dword j covers the
//             32-row blocks 2j, 2j+1; bit 8*(fb&1) + p <-> slot fb*32 + hi*16 + 2p, bit 16 + 8*(fb&1) + p <-> slot .. + 2p + 1.
PNR_HD static inline int64_t pnr_pad_samples(int64_t S) { return (S + 255) & ~(int64_t)255; }
// element offset of chunk c (slots 8c..8c+7) of sample s in a saved region of cpr = width / 8 chunks per sample
PNR_HD static inline int64_t pnr_saved_chunk(int cpr, int64_t s, int c)
{
    return (((s >> 3) * cpr + c) << 6) + ((((int)s & 7) ^ (((c >> 1) & 1) << 2)) << 3);
}
static inline void pnr_train_layout(const pnr_mlp_desc& d, int64_t S0, int64_t* acts_off, int64_t* dys_off, int64_t* gate_off = nullptr)
{
    const int64_t S = pnr_pad_samples(S0);
    int64_t o = 0;
    auto take = [&](int64_t w) { const int64_t r = o; o += w * S; o = (o + 63) & ~(int64_t)63; return r; };
    acts_off[0] = take(64);
    acts_off[1] = take(32);
    for (int l = 0; l < d.D; ++l) acts_off[2 + l] = take(d.W);
    acts_off[2 + d.D] = take(d.W);
    acts_off[3 + d.D] = take(d.W / 2);
    acts_off[4 + d.D] = take(d.W / 2);
    acts_off[5 + d.D] = take(d.W / 2);
    {
        int64_t g[24];
        for (int i = 0; i < 24; ++i) g[i] = -1;
        for (int l = 0; l < d.D; ++l) g[2 + l] = take(d.W / 16);          // bf16 units: w bits = w/16 units per sample
        g[3 + d.D] = take(d.W / 32);
        g[4 + d.D] = take(d.W / 32);
        g[5 + d.D] = take(d.W / 32);
        if (gate_off) for (int i = 0; i < 24; ++i) gate_off[i] = g[i];
    }
    acts_off[6 + d.D] = o;
    o = 0;
    dys_off[0] = take(d.W / 2);
    dys_off[1] = take(d.W);
    dys_off[2] = take(d.W / 2);
    dys_off[3] = take(d.W / 2);
    for (int l = 0; l < d.D; ++l) dys_off[4 + l] = take(d.W);
    // the output layers' dY = d_raw in bf16, FEAT slot order (k_mlp_bwd stores what it feeds the MFMAs):
    // [rgb, sigma] in 32 slots, semantic logits and instance logits in PNR_BWD_OUT_SLOTS = 64 slots each
    dys_off[4 + d.D] = take(32);
    dys_off[5 + d.D] = take(64);
    dys_off[6 + d.D] = take(64);
    dys_off[7 + d.D] = o;
}

// ---- fragment descriptors: ONE description of where every element of a packed image comes from, consumed by
// the host packer (pnr_mlp_pack / pnr_mlp_pack_bwd) and by the device packer (pnr_mlp_pack_device), so the two
// produce identical images by construction.
enum { PNR_F_WEIGHT = 0, PNR_F_WEIGHT_T = 1, PNR_F_BIAS = 2 };
struct PnrFragDesc {
    const float* src;       // weight matrix, row-major (out,in); bias vector for PNR_F_BIAS; null => zeros
    const float* src2;      // PNR_F_BIAS: second bias vector (rgb/sigma block), else unused
    int32_t kind;           // PNR_F_*
    int32_t ld, col_off;    // leading dimension, first column of this k-segment / of the h columns (transposed)
    int32_t row0;           // first row of the fragment's 32-row block (forward: output rows; transposed: input rows)
    int32_t lo, hi, off;    // forward: valid output-row window [lo,hi) and row offset into src
                            // transposed: valid window of the k index o (forward OUTPUT feature) and its offset into src
                            // bias: window / offset of src
    int32_t lo2, hi2, off2; // bias: window / offset of src2
    int32_t seg_kind, L, ks;// forward: slot map of the k dimension (pnr_seg_col) and the k-step
    int32_t nblk;           // bias: 32-row blocks in the fragment
    int32_t pad[2];
};

// value of element j (k-slot within the k-step) of lane `lane` of a weight fragment
PNR_HD static inline float pnr_frag_value(const PnrFragDesc& d, int kpl, int lane, int j)
{
    if (!d.src) return 0.0f;
    const int i = lane & 31, hi = lane >> 5;
    if (d.kind == PNR_F_WEIGHT) {
        const int row = d.row0 + i;
        const int col = pnr_seg_col(d.seg_kind, d.L, hi, d.ks * kpl + j);
        if (row < d.lo || row >= d.hi || col < 0) return 0.0f;
        return d.src[(int64_t)(row - d.off) * d.ld + d.col_off + col];
    }
    const int irow = d.row0 + i;                                   // transposed: rows are INPUT features
    const int o = pnr_seg_col(PNR_SEG_FEAT, 0, hi, d.ks * kpl + j);
    if (o < d.lo || o >= d.hi) return 0.0f;
    return d.src[(int64_t)(o - d.off) * d.ld + d.col_off + irow];
}
PNR_HD static inline float pnr_bias_value(const PnrFragDesc& d, int idx)      // idx = block*32 + row in block
{
    if (idx >= d.nblk * 32) return 0.0f;
    const int row = d.row0 + idx;
    if (d.src && row >= d.lo && row < d.hi) return d.src[row - d.off];
    if (d.src2 && row >= d.lo2 && row < d.hi2) return d.src2[row - d.off2];
    return 0.0f;
}
PNR_HD static inline uint16_t pnr_f32_to_bf16(float f)
{
    union { float f; uint32_t u; } c;
    c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
