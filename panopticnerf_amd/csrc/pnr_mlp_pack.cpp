// Host-side packer: dense nn.Linear parameters -> MFMA-fragment-ordered image
// (layout: pnr_mlp_layout.h).  Pure CPU code; replaces nothing in the reference -- it is the
// load-time counterpart of Network.__init__ / load_state_dict.
#include <math.h>
#include <string.h>

#include <vector>

#include "pnr_common.h"
#include "pnr_mlp_layout.h"
#include "pnr_mlp_plan.h"

static uint16_t f32_to_bf16_rne(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // quiet NaN
    const uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (uint16_t)(u >> 16);
}

int pnr_mlp_validate(const pnr_mlp_desc* d)
{
    PNR_REQUIRE(d, "pnr_mlp: null desc");
    PNR_REQUIRE(d->W == 256 || d->W == 128, "pnr_mlp: W=%d unsupported (128 or 256)", d->W);
    PNR_REQUIRE(d->D >= 2 && d->D <= 16, "pnr_mlp: D=%d outside [2,16]", d->D);
    PNR_REQUIRE(d->skip >= -1 && d->skip < d->D - 1, "pnr_mlp: skip=%d must be -1 or < D-1", d->skip);
    PNR_REQUIRE(d->xyz_L >= 0 && d->xyz_L <= 10, "pnr_mlp: xyz_L=%d outside [0,10]", d->xyz_L);
    PNR_REQUIRE(d->dir_L >= 0 && d->dir_L <= 4, "pnr_mlp: dir_L=%d outside [0,4]", d->dir_L);
    PNR_REQUIRE(d->n_sem >= 0 && d->n_sem <= 256 && d->n_inst >= 0 && d->n_inst <= 256,
                "pnr_mlp: n_sem/n_inst outside [0,256]");
    PNR_REQUIRE((d->n_sem == 0 && d->n_inst == 0) || d->head_W == d->W / 2,
                "pnr_mlp: head_W=%d must equal W/2=%d", d->head_W, d->W / 2);
    PNR_REQUIRE(d->precision == PNR_PREC_BF16 || d->precision == PNR_PREC_FP32, "pnr_mlp: bad precision %d",
                d->precision);
    return PNR_OK;
}

PNR_EXPORT int64_t pnr_mlp_packed_bytes(const pnr_mlp_desc* desc)
{
    if (pnr_mlp_validate(desc) != PNR_OK) return PNR_EINVAL;
    PnrPlan plan;
    pnr_build_plan(*desc, plan);
    return (int64_t)plan.total_bytes;
}

PNR_EXPORT int pnr_mlp_pack(const pnr_mlp_desc* desc, const pnr_mlp_params_host* p, void* packed_host)
{
    int rc = pnr_mlp_validate(desc);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(p && packed_host, "pnr_mlp_pack: null pointer");
    PNR_REQUIRE(p->pts_w && p->pts_b && p->alpha_w && p->alpha_b && p->feature_w && p->feature_b && p->views_w &&
                    p->views_b && p->rgb_w && p->rgb_b,
                "pnr_mlp_pack: missing trunk parameter");
    if (desc->n_sem) PNR_REQUIRE(p->sem0_w && p->sem0_b && p->sem1_w && p->sem1_b, "pnr_mlp_pack: missing semantic head");
    if (desc->n_inst) PNR_REQUIRE(p->inst0_w && p->inst0_b && p->inst1_w && p->inst1_b, "pnr_mlp_pack: missing instance head");

    PnrPlan plan;
    pnr_build_plan(*desc, plan);
    uint8_t* img = (uint8_t*)packed_host;
    memset(img, 0, plan.total_bytes);

    pnr_pack_header hdr;
    memset(&hdr, 0, sizeof(hdr));
    hdr.magic = PNR_PACK_MAGIC;
    hdr.version = 1;
    hdr.n_chunks = (uint32_t)plan.chunks.size();
    hdr.max_chunk_frags = (uint32_t)plan.max_chunk_frags;
    hdr.table_off = (uint32_t)plan.table_off;
    hdr.data_off = (uint32_t)plan.data_off;
    hdr.total_bytes = plan.total_bytes;
    memcpy(hdr.desc, desc, sizeof(pnr_mlp_desc));
    memcpy(img, &hdr, sizeof(hdr));
    pnr_chunk_entry* table = (pnr_chunk_entry*)(img + plan.table_off);

    const int W = desc->W, EX = 3 + 6 * desc->xyz_L, ED = 3 + 6 * desc->dir_L;
    const int kpl = pnr_kpl(desc->precision);
    const bool bf16 = desc->precision == PNR_PREC_BF16;

    for (size_t ci = 0; ci < plan.chunks.size(); ++ci) {
        const PnrChunk& ch = plan.chunks[ci];
        const PnrLayer& L = plan.layers[ch.layer];
        table[ci].off_frag = (uint32_t)ch.off_frag;
        table[ci].nfrag = (uint32_t)ch.nfrag;
        uint8_t* base = img + plan.data_off + (size_t)ch.off_frag * PNR_FRAG_BYTES;
        // Resolve (out row, segment, canonical column) -> weight value for this layer.
        auto weight = [&](int row, int seg, int col) -> float {
            if (row >= L.out_dim || col < 0) return 0.0f;
            switch (L.kind) {
            case PNR_L_TRUNK0:
            case PNR_L_TRUNK: {
                const int i = L.index;
                const float* Wm = p->pts_w[i];
                if (i == 0) return Wm[(size_t)row * EX + col];                      // [gamma(x)]
                if (L.nseg == 2)                                                     // [gamma(x), h]
                    return Wm[(size_t)row * (EX + W) + (seg == 0 ? col : EX + col)];
                return Wm[(size_t)row * W + col];                                    // [h]
            }
            case PNR_L_SEM0: return p->sem0_w[(size_t)row * W + col];
            case PNR_L_SEM1: return p->sem1_w[(size_t)row * desc->head_W + col];
            case PNR_L_INST0: return p->inst0_w[(size_t)row * W + col];
            case PNR_L_INST1: return p->inst1_w[(size_t)row * desc->head_W + col];
            case PNR_L_FEATURE: return p->feature_w[(size_t)row * W + col];
            case PNR_L_VIEWS: {
                if (seg == 1 && col >= ED) return 0.0f;
                const int c = seg == 0 ? col : W + col;            // [feature, gamma(d)]
                return p->views_w[(size_t)row * (W + ED) + c];
            }
            case PNR_L_RGBSIGMA:
                if (seg == 0) return row < 3 ? p->rgb_w[(size_t)row * (W / 2) + col] : 0.0f;
                return row == 3 ? p->alpha_w[col] : 0.0f;
            }
            return 0.0f;
        };
        auto bias = [&](int row) -> float {
            if (row >= L.out_dim) return 0.0f;
            switch (L.kind) {
            case PNR_L_TRUNK0:
            case PNR_L_TRUNK: return p->pts_b[L.index][row];
            case PNR_L_SEM0: return p->sem0_b[row];
            case PNR_L_SEM1: return p->sem1_b[row];
            case PNR_L_INST0: return p->inst0_b[row];
            case PNR_L_INST1: return p->inst1_b[row];
            case PNR_L_FEATURE: return p->feature_b[row];
            case PNR_L_VIEWS: return p->views_b[row];
            case PNR_L_RGBSIGMA: return row < 3 ? p->rgb_b[row] : p->alpha_b[0];
            }
            return 0.0f;
        };
        int frag_idx = 0;
        for (int fbl = 0; fbl < ch.nfb; ++fbl) {
            const int fb = ch.fb + fbl;
            for (int seg = 0; seg < L.nseg; ++seg) {
                const int kind = L.seg_kind[seg];
                const int vl = pnr_seg_vl(kind, L.seg_nfeat[seg]);
                const int Lf = kind == PNR_SEG_GX ? desc->xyz_L : desc->dir_L;
                for (int ks = 0; ks < vl / kpl; ++ks, ++frag_idx) {
                    uint8_t* frag = base + (size_t)frag_idx * PNR_FRAG_BYTES;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int i = lane & 31, hi = lane >> 5;
                        const int row = fb * 32 + i;
                        for (int j = 0; j < kpl; ++j) {
                            const int col = pnr_seg_col(kind, Lf, hi, ks * kpl + j);
                            const float w = weight(row, seg, col);
                            if (bf16) {
                                const uint16_t h = f32_to_bf16_rne(w);
                                memcpy(frag + lane * 16 + j * 2, &h, 2);
                            } else {
                                memcpy(frag + lane * 16 + j * 4, &w, 4);
                            }
                        }
                    }
                }
            }
        }
        float* bfrag = (float*)(base + (size_t)frag_idx * PNR_FRAG_BYTES);
        for (int fbl = 0; fbl < ch.nfb; ++fbl)
            for (int r = 0; r < 32; ++r) bfrag[fbl * 32 + r] = bias((ch.fb + fbl) * 32 + r);
    }
    return PNR_OK;
}

// ---- backward image: the transposed weights, same fragment format (rows = input features i of the
// forward layer, k = its output features o in FEAT slot order).  bf16.
static int bwd_validate(const pnr_mlp_desc* d)
{
    int rc = pnr_mlp_validate(d);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(d->precision == PNR_PREC_BF16, "pnr_mlp backward: bf16 only");
    PNR_REQUIRE(d->n_sem <= PNR_BWD_OUT_SLOTS && d->n_inst <= PNR_BWD_OUT_SLOTS,
                "pnr_mlp backward: n_sem / n_inst must be <= %d", PNR_BWD_OUT_SLOTS);
    return PNR_OK;
}

PNR_EXPORT int64_t pnr_mlp_bwd_packed_bytes(const pnr_mlp_desc* desc)
{
    if (bwd_validate(desc) != PNR_OK) return PNR_EINVAL;
    PnrBPlan plan;
    pnr_build_bwd_plan(*desc, plan);
    return (int64_t)plan.total_bytes;
}

PNR_EXPORT int pnr_mlp_pack_bwd(const pnr_mlp_desc* desc, const pnr_mlp_params_host* p, void* packed_host)
{
    int rc = bwd_validate(desc);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(p && packed_host, "pnr_mlp_pack_bwd: null pointer");
    PNR_REQUIRE(p->pts_w && p->alpha_w && p->feature_w && p->views_w && p->rgb_w, "pnr_mlp_pack_bwd: missing trunk parameter");
    if (desc->n_sem) PNR_REQUIRE(p->sem0_w && p->sem1_w, "pnr_mlp_pack_bwd: missing semantic head");
    if (desc->n_inst) PNR_REQUIRE(p->inst0_w && p->inst1_w, "pnr_mlp_pack_bwd: missing instance head");
    PnrBPlan plan;
    pnr_build_bwd_plan(*desc, plan);
    uint8_t* img = (uint8_t*)packed_host;
    memset(img, 0, plan.total_bytes);
    pnr_pack_header hdr;
    memset(&hdr, 0, sizeof(hdr));
    hdr.magic = PNR_PACK_MAGIC;
    hdr.version = 2;
    hdr.n_chunks = (uint32_t)plan.chunks.size();
    hdr.max_chunk_frags = (uint32_t)plan.max_chunk_frags;
    hdr.table_off = (uint32_t)plan.table_off;
    hdr.data_off = (uint32_t)plan.data_off;
    hdr.total_bytes = plan.total_bytes;
    memcpy(hdr.desc, desc, sizeof(pnr_mlp_desc));
    memcpy(img, &hdr, sizeof(hdr));
    pnr_chunk_entry* table = (pnr_chunk_entry*)(img + plan.table_off);
    const int W = desc->W, H = W / 2, EX = 3 + 6 * desc->xyz_L, ED = 3 + 6 * desc->dir_L;

    for (size_t ci = 0; ci < plan.chunks.size(); ++ci) {
        const PnrChunk& ch = plan.chunks[ci];
        const PnrBLayer& L = plan.layers[ch.layer];
        table[ci].off_frag = (uint32_t)ch.off_frag;
        table[ci].nfrag = (uint32_t)ch.nfrag;
        uint8_t* base = img + plan.data_off + (size_t)ch.off_frag * PNR_FRAG_BYTES;
        // W_forward[o][i] of the k-segment's layer, 0 outside it
        auto wt = [&](int kseg, int o, int i) -> float {
            switch (kseg) {
            case PNR_K_RGBS:
                if (L.kind == PNR_B_DG) return o < 3 ? p->rgb_w[(size_t)o * H + i] : 0.0f;     // d g
                return o == 3 ? p->alpha_w[i] : 0.0f;                                          // d h (sigma row)
            case PNR_K_VIEWS: return o < H ? p->views_w[(size_t)o * (W + ED) + i] : 0.0f;      // feature columns
            case PNR_K_SEM1: return (desc->n_sem && o < desc->n_sem) ? p->sem1_w[(size_t)o * H + i] : 0.0f;
            case PNR_K_INST1: return (desc->n_inst && o < desc->n_inst) ? p->inst1_w[(size_t)o * H + i] : 0.0f;
            case PNR_K_FEATURE: return p->feature_w[(size_t)o * W + i];
            case PNR_K_SEM0: return desc->n_sem ? p->sem0_w[(size_t)o * W + i] : 0.0f;
            case PNR_K_INST0: return desc->n_inst ? p->inst0_w[(size_t)o * W + i] : 0.0f;
            case PNR_K_TRUNK: {
                const int l = L.index;
                const bool sk = (l - 1 == desc->skip);
                return p->pts_w[l][(size_t)o * (sk ? EX + W : W) + (sk ? EX + i : i)];
            }
            }
            return 0.0f;
        };
        int frag_idx = 0;
        for (int fbl = 0; fbl < ch.nfb; ++fbl) {
            const int fb = ch.fb + fbl;
            for (int seg = 0; seg < L.nseg; ++seg) {
                const int vl = L.seg_slots[seg] / 2;
                for (int ks = 0; ks < vl / 8; ++ks, ++frag_idx) {
                    uint8_t* frag = base + (size_t)frag_idx * PNR_FRAG_BYTES;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int irow = fb * 32 + (lane & 31), hi = lane >> 5;
                        for (int j = 0; j < 8; ++j) {
                            const int o = pnr_seg_col(PNR_SEG_FEAT, 0, hi, ks * 8 + j);
                            const uint16_t h = f32_to_bf16_rne(wt(L.seg_kind[seg], o, irow));
                            memcpy(frag + lane * 16 + j * 2, &h, 2);
                        }
                    }
                }
            }
        }
    }
    return PNR_OK;
}
