// Execution plan of the fused MLP: the ordered list of layers and (layer, 32-row block)
// chunks.  Host-only; used by the packer and by the launcher (chunk count, LDS slot size).
// Order = the order the kernel consumes the weight stream:
//   plan 0 (classic):          trunk 0..D-1 | feature | views | rgb+sigma | [sem0, sem1] | [inst0, inst1]
//   plan 1 (fused inference):  trunk 0..D-1 | feature | views | rgb+sigma | [sem0] | [inst0] | logits = {sem1, inst1} in ONE chunk;
//                              layer 0 is one chunk of all W/32 blocks (32 MFMAs) instead of two of 16
//   plan 2 (two-tile kernel, csrc/asm/gen_mlp_tt.py): plan 1's order (sem0 | inst0 | sem1 | inst1) with layer 0, sem1 and inst1
//                              as ONE chunk EACH (all blocks of the layer), and no chunk above 33 fragments -- a layer whose 2-block chunk would be larger (the layer behind
//                              the skip: 2 x 20 + 1, views: 2 x 18 + 1) is cut into 1-block chunks -- so that FOUR weight slots fit
//                              the LDS; the number of chunks is then a multiple of 4 (slot = chunk % 4 is static)
#pragma once
#include <stddef.h>

#include <utility>
#include <vector>

#include "pnr.h"
#include "pnr_mlp_layout.h"

// plan 1 exists for: bf16, W = 256, 1..2 semantic and 0..1 instance logit blocks (the instantiated kernel tails)
static inline int pnr_head_depth(const pnr_mlp_desc& d) { return d.head_depth == 1 ? 1 : 2; }
static inline int pnr_plan1_supported(const pnr_mlp_desc& d)
{
    const int nbs = (d.n_sem + 31) / 32, nbi = (d.n_inst + 31) / 32;
    return d.precision == PNR_PREC_BF16 && d.W == 256 && nbs >= 1 && nbs <= 2 && nbi <= 1 && pnr_head_depth(d) == 2;
}
// plan 2 (the two-tile kernel's image): the geometries its generated kernels exist for -- the 8 x 256 network of BASELINE configs 2..5
// with no heads, a semantic head of 1..3 logit blocks, or a semantic head of 1..2 and a 1-block instance head (csrc/asm/gen_mlp_tt.py); heads of
// depth 2 (W -> W/2 -> n) or 1 (one Linear W -> n: the generic head_depth-1 branch of pnr_build_plan with the plan-2 chunk rules)
#define PNR_PLAN2_MAX_CHUNK_FRAGS 33
static inline int pnr_plan2_supported(const pnr_mlp_desc& d)
{
    const int nbs = (d.n_sem + 31) / 32, nbi = (d.n_inst + 31) / 32;
    return d.precision == PNR_PREC_BF16 && d.W == 256 && d.D == 8 && d.skip == 4 && d.xyz_L == 10 && d.dir_L == 4 &&      // head_tap 0 and (round 6) 1
           nbs <= 3 && nbi <= (nbs == 1 || nbs == 2 ? 1 : 0) && (nbs == 0 || d.head_W == 128);      // head_depth 2 and (round 6) 1;
           // a third semantic block (65..96 classes) without an instance head: six logit accumulators + two for the local weights
}

#ifndef PNR_PLAN1_TRUNK0_MERGE
#define PNR_PLAN1_TRUNK0_MERGE 1
#endif
struct PnrLayer {
    int kind, index;       // index: trunk layer number; PNR_L_LOGITS: number of semantic blocks
    int out_dim, n_fb;     // output rows, 32-row blocks
    int nseg;
    int seg_kind[2], seg_nfeat[2];
    int nks;               // k-steps (weight fragments) per 32-row block
    int fbc;               // blocks per chunk
};
struct PnrChunk { int layer, fb, nfb, off_frag, nfrag; };   // fb: first block, nfb: blocks in this chunk
struct PnrPlan {
    std::vector<PnrLayer> layers;
    std::vector<PnrChunk> chunks;
    int max_chunk_frags;
    size_t table_off, data_off, total_bytes;
};

static inline void pnr_build_plan(const pnr_mlp_desc& d, PnrPlan& plan)
{
    const int kpl = pnr_kpl(d.precision);
    auto add = [&](int kind, int index, int out_dim, int k0, int n0, int k1 = -1, int n1 = 0) {
        PnrLayer L;
        L.kind = kind; L.index = index; L.out_dim = out_dim; L.n_fb = (out_dim + 31) / 32;
        L.nseg = k1 >= 0 ? 2 : 1;
        L.seg_kind[0] = k0; L.seg_nfeat[0] = n0; L.seg_kind[1] = k1; L.seg_nfeat[1] = n1;
        L.nks = pnr_seg_vl(k0, n0) / kpl + (k1 >= 0 ? pnr_seg_vl(k1, n1) / kpl : 0);
        L.fbc = pnr_layer_fbc(kind, d.precision);
        if (L.n_fb % L.fbc) L.fbc = 1;
        if (PNR_PLAN1_TRUNK0_MERGE && d.plan >= 1 && kind == PNR_L_TRUNK0) L.fbc = L.n_fb;      // plan 1, 2: layer 0 is ONE chunk of W/32 blocks x 4 k-steps
        if (d.plan == 2 && (kind == PNR_L_SEM1 || kind == PNR_L_INST1)) L.fbc = L.n_fb;         // plan 2: a logit layer is one chunk
        if (d.plan == 2 && L.fbc * L.nks + 1 > PNR_PLAN2_MAX_CHUNK_FRAGS) L.fbc = 1;            // plan 2: four slots must fit the LDS
        plan.layers.push_back(L);
    };
    plan.layers.clear();
    plan.chunks.clear();
    for (int i = 0; i < d.D; ++i) {
        if (i == 0) add(PNR_L_TRUNK0, i, d.W, PNR_SEG_GX, 0);
        else if (i - 1 == d.skip) add(PNR_L_TRUNK, i, d.W, PNR_SEG_GX, 0, PNR_SEG_FEAT, d.W);
        else add(PNR_L_TRUNK, i, d.W, PNR_SEG_FEAT, d.W);
    }
    // appearance first, panoptic heads last: sigma (hence every sample's compositing weight) is known before the logit
    // blocks are produced, which is what lets the fused inference epilogue reduce them over the ray on the fly
    add(PNR_L_FEATURE, 0, d.W, PNR_SEG_FEAT, d.W);
    add(PNR_L_VIEWS, 0, d.W / 2, PNR_SEG_FEAT, d.W, PNR_SEG_GD, 0);
    add(PNR_L_RGBSIGMA, 0, 4, PNR_SEG_FEAT, d.W / 2, PNR_SEG_FEAT, d.W);
    if (d.plan == 1) {
        const int nbs = (d.n_sem + 31) / 32, nbi = (d.n_inst + 31) / 32;
        if (d.n_sem) add(PNR_L_SEM0, 0, d.head_W, PNR_SEG_FEAT, d.W);
        if (d.n_inst) add(PNR_L_INST0, 0, d.head_W, PNR_SEG_FEAT, d.W);
        add(PNR_L_LOGITS, nbs, (nbs + nbi) * 32, PNR_SEG_FEAT, d.head_W);
        plan.layers.back().fbc = nbs + nbi;         // every logit block in one chunk
    } else if (d.plan == 2 && pnr_head_depth(d) == 2) {
        if (d.n_sem) add(PNR_L_SEM0, 0, d.head_W, PNR_SEG_FEAT, d.W);
        if (d.n_inst) add(PNR_L_INST0, 0, d.head_W, PNR_SEG_FEAT, d.W);
        if (d.n_sem) add(PNR_L_SEM1, 0, d.n_sem, PNR_SEG_FEAT, d.head_W);
        if (d.n_inst) add(PNR_L_INST1, 0, d.n_inst, PNR_SEG_FEAT, d.head_W);
    } else if (pnr_head_depth(d) == 1) {      // one Linear per head, straight from the tap (W inputs)
        if (d.n_sem) add(PNR_L_SEM1, 0, d.n_sem, PNR_SEG_FEAT, d.W);
        if (d.n_inst) add(PNR_L_INST1, 0, d.n_inst, PNR_SEG_FEAT, d.W);
    } else {
        if (d.n_sem) {
            add(PNR_L_SEM0, 0, d.head_W, PNR_SEG_FEAT, d.W);
            add(PNR_L_SEM1, 0, d.n_sem, PNR_SEG_FEAT, d.head_W);
        }
        if (d.n_inst) {
            add(PNR_L_INST0, 0, d.head_W, PNR_SEG_FEAT, d.W);
            add(PNR_L_INST1, 0, d.n_inst, PNR_SEG_FEAT, d.head_W);
        }
    }
    int off = 0, mx = 0;
    for (size_t li = 0; li < plan.layers.size(); ++li) {
        const PnrLayer& L = plan.layers[li];
        for (int fb = 0; fb < L.n_fb; fb += L.fbc) {
            PnrChunk c;
            c.layer = (int)li; c.fb = fb; c.nfb = L.fbc; c.off_frag = off; c.nfrag = L.fbc * L.nks + 1;   // + bias fragment
            off += c.nfrag;
            if (c.nfrag > mx) mx = c.nfrag;
            plan.chunks.push_back(c);
        }
    }
    plan.max_chunk_frags = mx;
    plan.table_off = sizeof(pnr_pack_header);
    size_t t = plan.table_off + plan.chunks.size() * sizeof(pnr_chunk_entry);
    plan.data_off = (t + 1023) & ~(size_t)1023;
    plan.total_bytes = plan.data_off + (size_t)off * PNR_FRAG_BYTES;
}

// ---- backward (dgrad) plan, bf16 only.  Every backward layer computes  dX^T = W^T * dY^T  for one tensor X:
// rows = X's features (32-row blocks), k = the forward layer's OUTPUT features, supplied by up to four
// k-segments in FEAT slot order (pnr_seg_col).  Chunks hold 2 row blocks and no bias fragment.
// Execution order: DG | DF | [DSHS] | [DSHI] | DH | DX(D-1) ... DX(1)                     (head_tap 0: the heads read h)
//                  DG | [DSHS] | [DSHI] | DF (views + head contributions) | DH | DX ...     (head_tap 1: the heads read the feature)
enum { PNR_B_DG = 0, PNR_B_DF, PNR_B_DSHS, PNR_B_DSHI, PNR_B_DH, PNR_B_DX };
enum { PNR_K_RGBS = 0, PNR_K_VIEWS, PNR_K_SEM1, PNR_K_INST1, PNR_K_FEATURE, PNR_K_SEM0, PNR_K_INST0, PNR_K_TRUNK };
#define PNR_BWD_OUT_SLOTS 64      /* k-slots reserved for a head's outputs (n_sem, n_inst <= 64 when training) */

struct PnrBLayer {
    int kind, index;        // index: trunk layer l for PNR_B_DX
    int rows, n_fb;         // X's width, 32-row blocks
    int nseg;
    int seg_kind[4], seg_slots[4];
    int nks;                // k-steps per 32-row block
    int fbc;                // 32-row blocks per chunk
};
// blocks per chunk of a backward layer: 2, except the 34-k-step dH layer (its two-block chunk would be 68 KiB and
// three weight slots -- k_mlp_bwd prefetches two chunks ahead -- would not fit the LDS)
#define PNR_BWD_FBC 2
#define PNR_BWD_FBC_DH 1
struct PnrBPlan {
    std::vector<PnrBLayer> layers;
    std::vector<PnrChunk> chunks;
    int max_chunk_frags;
    size_t table_off, data_off, total_bytes;
};

static inline void pnr_build_bwd_plan(const pnr_mlp_desc& d, PnrBPlan& plan)
{
    const int W = d.W, H = d.W / 2;
    auto add = [&](int kind, int index, int rows, std::initializer_list<std::pair<int, int>> segs) {
        PnrBLayer L;
        L.kind = kind; L.index = index; L.rows = rows; L.n_fb = rows / 32; L.nseg = 0; L.nks = 0;
        L.fbc = kind == PNR_B_DH ? PNR_BWD_FBC_DH : PNR_BWD_FBC;
        for (auto& sg : segs) { L.seg_kind[L.nseg] = sg.first; L.seg_slots[L.nseg] = sg.second; L.nks += sg.second / 16; ++L.nseg; }
        plan.layers.push_back(L);
    };
    plan.layers.clear();
    plan.chunks.clear();
    // head_depth 1 (one Linear W -> n per head): no hidden head layer, so no DSHS / DSHI step; the logit gradients themselves
    // (64 slots, zero-extended to H) take the place of dY_sem0 / dY_inst0 in the chain that reaches the tap, and the packer fills
    // the PNR_K_SEM0 / PNR_K_INST0 segments from the single Linear's transposed weights (describe_backward)
    const bool deep = pnr_head_depth(d) == 2;
    add(PNR_B_DG, 0, H, {{PNR_K_RGBS, 32}});
    if (d.head_tap == 1) {
        // the head gradients reach the FEATURE: d F = views^T dY_views + sem0^T dY_sem0 + inst0^T dY_inst0 (all three segments
        // always present: an absent head's fragments are zeros), then d h = feature^T d F + alpha^T d sigma
        if (d.n_sem && deep) add(PNR_B_DSHS, 0, H, {{PNR_K_SEM1, PNR_BWD_OUT_SLOTS}});
        if (d.n_inst && deep) add(PNR_B_DSHI, 0, H, {{PNR_K_INST1, PNR_BWD_OUT_SLOTS}});
        add(PNR_B_DF, 0, W, {{PNR_K_VIEWS, H}, {PNR_K_SEM0, H}, {PNR_K_INST0, H}});
        add(PNR_B_DH, 0, W, {{PNR_K_FEATURE, W}, {PNR_K_RGBS, 32}});
    } else {
        add(PNR_B_DF, 0, W, {{PNR_K_VIEWS, H}});
        if (d.n_sem && deep) add(PNR_B_DSHS, 0, H, {{PNR_K_SEM1, PNR_BWD_OUT_SLOTS}});
        if (d.n_inst && deep) add(PNR_B_DSHI, 0, H, {{PNR_K_INST1, PNR_BWD_OUT_SLOTS}});
        add(PNR_B_DH, 0, W, {{PNR_K_FEATURE, W}, {PNR_K_RGBS, 32}, {PNR_K_SEM0, H}, {PNR_K_INST0, H}});
    }
    for (int l = d.D - 1; l >= 1; --l) add(PNR_B_DX, l, W, {{PNR_K_TRUNK, W}});
    int off = 0, mx = 0;
    for (size_t li = 0; li < plan.layers.size(); ++li) {
        const PnrBLayer& L = plan.layers[li];
        for (int fb = 0; fb < L.n_fb; fb += L.fbc) {
            PnrChunk c;
            c.layer = (int)li; c.fb = fb; c.nfb = L.fbc; c.off_frag = off; c.nfrag = L.fbc * L.nks;
            off += c.nfrag;
            if (c.nfrag > mx) mx = c.nfrag;
            plan.chunks.push_back(c);
        }
    }
    plan.max_chunk_frags = mx;
    plan.table_off = sizeof(pnr_pack_header);
    size_t t = plan.table_off + plan.chunks.size() * sizeof(pnr_chunk_entry);
    plan.data_off = (t + 1023) & ~(size_t)1023;
    plan.total_bytes = plan.data_off + (size_t)off * PNR_FRAG_BYTES;
}
