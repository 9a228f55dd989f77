// Packer: dense nn.Linear parameters -> MFMA-fragment-ordered image (layout: pnr_mlp_layout.h), forward and
// transposed (backward) images, on the host (pnr_mlp_pack, pnr_mlp_pack_bwd) or on the device
// (pnr_mlp_pack_device: reads the live parameter tensors, no host round trip per optimiser step).
// Both go through the same fragment descriptors.  Replaces nothing in the reference -- it is the load-time
// counterpart of Network.__init__ / load_state_dict.
#include <math.h>
#include <string.h>

#include <vector>

#include "pnr_common.h"
#include "pnr_mlp_layout.h"
#include "pnr_mlp_plan.h"
#include "pnr_mlp_tt.h"

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
    PNR_REQUIRE(d->head_tap == 0 || d->head_tap == 1, "pnr_mlp: head_tap=%d must be 0 (trunk output) or 1 (feature)", d->head_tap);
    PNR_REQUIRE(d->head_depth >= 0 && d->head_depth <= 2, "pnr_mlp: head_depth=%d must be 1 or 2", d->head_depth);
#ifndef PNR_TRAIN_TILES_EXPERIMENT
#define PNR_TRAIN_TILES_EXPERIMENT 0
#endif
    PNR_REQUIRE(d->schedule >= 0 && d->schedule <= (PNR_TRAIN_TILES_EXPERIMENT ? 6 : 2), "pnr_mlp: schedule=%d must be 0 (default), 1 (lock-step) or 2 (ping-pong)", d->schedule);
    PNR_REQUIRE(d->plan == 0 || (d->plan == 1 && pnr_plan1_supported(*d)) || (d->plan == 2 && pnr_plan2_supported(*d)),
                "pnr_mlp: plan=%d is not available for this geometry (ask pnr_mlp_fused_plan)", d->plan);
    // the descriptor must be zero-initialised (include/pnr.h): the diagnostic words are READ -- clk_probe is a device address the
    // forward kernels store 16 bytes to, flags select kernels -- so garbage there is rejected where it can be recognised
    const uint32_t trace = (uint32_t)d->flags & 0xFF00u;
    PNR_REQUIRE(((uint32_t)d->flags & ~(uint32_t)(PNR_MLP_SOFTMAX | 0xFF70u | PNR_MLP_WG_CAP(0x1FF))) == 0 && (trace == 0 || trace == PNR_MLP_TRACE) &&
                (trace != 0 || ((uint32_t)d->flags & 0x70u) == 0),
                "pnr_mlp: unknown bits in flags=0x%x (zero-initialise pnr_mlp_desc; PNR_MLP_* are the defined bits)", (unsigned)d->flags);
    PNR_REQUIRE(((uint32_t)d->clk_probe[0] & 15u) == 0, "pnr_mlp: clk_probe=0x%08x%08x is not a 16-byte aligned device address "
                "(zero-initialise pnr_mlp_desc; clk_probe is a diagnostics field, 0 = off)", (unsigned)d->clk_probe[1], (unsigned)d->clk_probe[0]);
    PNR_REQUIRE(trace == 0 || (d->clk_probe[0] | d->clk_probe[1]) != 0, "pnr_mlp: PNR_MLP_TRACE needs clk_probe");
    return PNR_OK;
}

PNR_EXPORT int pnr_mlp_fused_plan(const pnr_mlp_desc* desc)
{
    if (!desc) return 0;
    pnr_mlp_desc d = *desc;
    d.plan = 0;
    if (pnr_mlp_validate(&d) != PNR_OK) return 0;
    if ((d.flags & PNR_MLP_SOFTMAX) && d.n_sem + d.n_inst > 0)      // softmax compositing: the best plan that HAS a softmax kernel
        return pnr_plan2_supported(d) ? 2 : pnr_plan1_supported(d) ? 1 : 0;       // (k_mlp_tt_sm_* / _d1sm_*: both head depths)
    return pnr_plan2_supported(d) ? 2 : pnr_plan1_supported(d) ? 1 : 0;
}

static int bwd_validate(const pnr_mlp_desc* d)
{
    int rc = pnr_mlp_validate(d);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(d->plan == 0, "pnr_mlp backward: plan must be 0");
    PNR_REQUIRE(d->precision == PNR_PREC_BF16, "pnr_mlp backward: bf16 only");
    PNR_REQUIRE(d->n_sem <= PNR_BWD_OUT_SLOTS && d->n_inst <= PNR_BWD_OUT_SLOTS,
                "pnr_mlp backward: n_sem / n_inst must be <= %d", PNR_BWD_OUT_SLOTS);
    return PNR_OK;
}

static int check_params(const pnr_mlp_desc* desc, const pnr_mlp_params_host* p, bool bias)
{
    PNR_REQUIRE(p, "pnr_mlp_pack: null params");
    PNR_REQUIRE(p->pts_w && p->alpha_w && p->feature_w && p->views_w && p->rgb_w, "pnr_mlp_pack: missing trunk parameter");
    if (bias) PNR_REQUIRE(p->pts_b && p->alpha_b && p->feature_b && p->views_b && p->rgb_b, "pnr_mlp_pack: missing trunk bias");
    const bool deep = pnr_head_depth(*desc) == 2;
    if (desc->n_sem) PNR_REQUIRE((!deep || p->sem0_w) && p->sem1_w && (!bias || ((!deep || p->sem0_b) && p->sem1_b)), "pnr_mlp_pack: missing semantic head");
    if (desc->n_inst) PNR_REQUIRE((!deep || p->inst0_w) && p->inst1_w && (!bias || ((!deep || p->inst0_b) && p->inst1_b)), "pnr_mlp_pack: missing instance head");
    return PNR_OK;
}

// ------------------------------------------------------------------------------- descriptors
struct Image {
    pnr_pack_header hdr;
    std::vector<pnr_chunk_entry> table;
    std::vector<PnrFragDesc> frags;      // one per fragment of the data area, in order
    size_t table_off, data_off, total_bytes;
};

static PnrFragDesc zero_desc()
{
    PnrFragDesc f;
    memset(&f, 0, sizeof(f));
    return f;
}

static void fill_header(Image& im, const pnr_mlp_desc& d, int version, size_t n_chunks, int max_frags, size_t table_off,
                        size_t data_off, size_t total)
{
    memset(&im.hdr, 0, sizeof(im.hdr));
    im.hdr.magic = PNR_PACK_MAGIC;
    im.hdr.version = (uint32_t)version;
    im.hdr.n_chunks = (uint32_t)n_chunks;
    im.hdr.max_chunk_frags = (uint32_t)max_frags;
    im.hdr.table_off = (uint32_t)table_off;
    im.hdr.data_off = (uint32_t)data_off;
    im.hdr.total_bytes = total;
    memcpy(im.hdr.desc, &d, sizeof(pnr_mlp_desc));
    im.table_off = table_off; im.data_off = data_off; im.total_bytes = total;
}

static void describe_forward(const pnr_mlp_desc& d, const pnr_mlp_params_host& p, Image& im)
{
    PnrPlan plan;
    pnr_build_plan(d, plan);
    fill_header(im, d, 1, plan.chunks.size(), plan.max_chunk_frags, plan.table_off, plan.data_off, plan.total_bytes);
    const int W = d.W, H = W / 2, EX = 3 + 6 * d.xyz_L, ED = 3 + 6 * d.dir_L;
    const int kpl = pnr_kpl(d.precision);
    for (const PnrChunk& ch : plan.chunks) {
        const PnrLayer& L = plan.layers[ch.layer];
        im.table.push_back({(uint32_t)ch.off_frag, (uint32_t)ch.nfrag});
        for (int fbl = 0; fbl < ch.nfb; ++fbl) {
            for (int seg = 0; seg < L.nseg; ++seg) {
                const int kind = L.seg_kind[seg];
                const int nks = pnr_seg_vl(kind, L.seg_nfeat[seg]) / kpl;
                for (int ks = 0; ks < nks; ++ks) {
                    PnrFragDesc f = zero_desc();
                    f.kind = PNR_F_WEIGHT;
                    f.row0 = (ch.fb + fbl) * 32;
                    f.lo = 0; f.hi = L.out_dim; f.off = 0;
                    if (L.kind == PNR_L_LOGITS) {               // blocks [0, index): sem1's rows, the rest: inst1's rows
                        const bool sem = ch.fb + fbl < L.index;
                        f.src = sem ? p.sem1_w : p.inst1_w; f.ld = H;
                        f.row0 = (sem ? ch.fb + fbl : ch.fb + fbl - L.index) * 32;
                        f.hi = sem ? d.n_sem : d.n_inst;
                    }
                    f.seg_kind = kind; f.L = kind == PNR_SEG_GX ? d.xyz_L : d.dir_L; f.ks = ks;
                    switch (L.kind) {
                    case PNR_L_TRUNK0: f.src = p.pts_w[0]; f.ld = EX; break;
                    case PNR_L_TRUNK:
                        f.src = p.pts_w[L.index];
                        if (L.nseg == 2) { f.ld = EX + W; f.col_off = seg == 0 ? 0 : EX; }      // [gamma(x), h]
                        else f.ld = W;
                        break;
                    case PNR_L_SEM0: f.src = p.sem0_w; f.ld = W; break;
                    case PNR_L_SEM1: f.src = p.sem1_w; f.ld = pnr_head_depth(d) == 1 ? W : H; break;
                    case PNR_L_INST0: f.src = p.inst0_w; f.ld = W; break;
                    case PNR_L_INST1: f.src = p.inst1_w; f.ld = pnr_head_depth(d) == 1 ? W : H; break;
                    case PNR_L_FEATURE: f.src = p.feature_w; f.ld = W; break;
                    case PNR_L_VIEWS: f.src = p.views_w; f.ld = W + ED; f.col_off = seg == 0 ? 0 : W; break;   // [feature, gamma(d)]
                    case PNR_L_LOGITS: break;                  // set above
                    case PNR_L_RGBSIGMA:
                        if (seg == 0) { f.src = p.rgb_w; f.ld = H; f.lo = 0; f.hi = 3; f.off = 0; }             // rows 0..2 <- g
                        else { f.src = p.alpha_w; f.ld = W; f.lo = 3; f.hi = 4; f.off = 3; }                    // row 3 <- h
                        break;
                    }
                    im.frags.push_back(f);
                }
            }
        }
        PnrFragDesc b = zero_desc();
        b.kind = PNR_F_BIAS;
        b.row0 = ch.fb * 32; b.nblk = ch.nfb;
        b.lo = 0; b.hi = L.out_dim; b.off = 0;
        switch (L.kind) {
        case PNR_L_TRUNK0: case PNR_L_TRUNK: b.src = p.pts_b[L.index]; break;
        case PNR_L_SEM0: b.src = p.sem0_b; break;
        case PNR_L_SEM1: b.src = p.sem1_b; break;
        case PNR_L_INST0: b.src = p.inst0_b; break;
        case PNR_L_INST1: b.src = p.inst1_b; break;
        case PNR_L_FEATURE: b.src = p.feature_b; break;
        case PNR_L_VIEWS: b.src = p.views_b; break;
        case PNR_L_RGBSIGMA: b.src = p.rgb_b; b.lo = 0; b.hi = 3; b.src2 = p.alpha_b; b.lo2 = 3; b.hi2 = 4; b.off2 = 3; break;
        case PNR_L_LOGITS:      // rows [0, 32 nbs): sem1's bias; rows 32 nbs + [0, n_inst): inst1's
            b.src = p.sem1_b; b.lo = 0; b.hi = d.n_sem; b.off = 0;
            b.src2 = p.inst1_b; b.lo2 = L.index * 32; b.hi2 = L.index * 32 + d.n_inst; b.off2 = L.index * 32;
            break;
        }
        im.frags.push_back(b);
    }
}

static void describe_backward(const pnr_mlp_desc& d, const pnr_mlp_params_host& p, Image& im)
{
    PnrBPlan plan;
    pnr_build_bwd_plan(d, plan);
    fill_header(im, d, 2, plan.chunks.size(), plan.max_chunk_frags, plan.table_off, plan.data_off, plan.total_bytes);
    const int W = d.W, H = W / 2, EX = 3 + 6 * d.xyz_L, ED = 3 + 6 * d.dir_L;
    for (const PnrChunk& ch : plan.chunks) {
        const PnrBLayer& L = plan.layers[ch.layer];
        im.table.push_back({(uint32_t)ch.off_frag, (uint32_t)ch.nfrag});
        for (int fbl = 0; fbl < ch.nfb; ++fbl) {
            for (int seg = 0; seg < L.nseg; ++seg) {
                for (int ks = 0; ks < L.seg_slots[seg] / 16; ++ks) {
                    PnrFragDesc f = zero_desc();
                    f.kind = PNR_F_WEIGHT_T;
                    f.row0 = (ch.fb + fbl) * 32;        // input-feature rows
                    f.ks = ks;
                    switch (L.seg_kind[seg]) {
                    case PNR_K_RGBS:
                        if (L.kind == PNR_B_DG) { f.src = p.rgb_w; f.ld = H; f.lo = 0; f.hi = 3; f.off = 0; }      // d g
                        else { f.src = p.alpha_w; f.ld = W; f.lo = 3; f.hi = 4; f.off = 3; }                        // d h (sigma)
                        break;
                    case PNR_K_VIEWS: f.src = p.views_w; f.ld = W + ED; f.lo = 0; f.hi = H; break;                   // feature columns
                    case PNR_K_SEM1: f.src = d.n_sem ? p.sem1_w : nullptr; f.ld = H; f.lo = 0; f.hi = d.n_sem; break;
                    case PNR_K_INST1: f.src = d.n_inst ? p.inst1_w : nullptr; f.ld = H; f.lo = 0; f.hi = d.n_inst; break;
                    case PNR_K_FEATURE: f.src = p.feature_w; f.ld = W; f.lo = 0; f.hi = W; break;
                    // head_depth 1: the segment carries the logit gradients themselves (n <= 64 valid k-slots, the rest zero)
                    // against the single Linear (n, W) of the head
                    case PNR_K_SEM0:
                        if (pnr_head_depth(d) == 1) { f.src = d.n_sem ? p.sem1_w : nullptr; f.ld = W; f.lo = 0; f.hi = d.n_sem; }
                        else { f.src = d.n_sem ? p.sem0_w : nullptr; f.ld = W; f.lo = 0; f.hi = H; }
                        break;
                    case PNR_K_INST0:
                        if (pnr_head_depth(d) == 1) { f.src = d.n_inst ? p.inst1_w : nullptr; f.ld = W; f.lo = 0; f.hi = d.n_inst; }
                        else { f.src = d.n_inst ? p.inst0_w : nullptr; f.ld = W; f.lo = 0; f.hi = H; }
                        break;
                    case PNR_K_TRUNK: {
                        const bool sk = (L.index - 1 == d.skip);
                        f.src = p.pts_w[L.index]; f.ld = sk ? EX + W : W; f.col_off = sk ? EX : 0; f.lo = 0; f.hi = W;
                        break;
                    }
                    }
                    im.frags.push_back(f);
                }
            }
        }
    }
}

static void fill_fragment_host(const PnrFragDesc& f, int precision, uint8_t* frag)
{
    const int kpl = pnr_kpl(precision);
    if (f.kind == PNR_F_BIAS) {
        float* b = (float*)frag;
        for (int i = 0; i < 256; ++i) b[i] = pnr_bias_value(f, i);
        return;
    }
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < kpl; ++j) {
            const float w = pnr_frag_value(f, kpl, lane, j);
            if (precision == PNR_PREC_BF16) {
                const uint16_t h = pnr_f32_to_bf16(w);
                memcpy(frag + lane * 16 + j * 2, &h, 2);
            } else {
                memcpy(frag + lane * 16 + j * 4, &w, 4);
            }
        }
}

static void write_host(const Image& im, int precision, void* out)
{
    uint8_t* img = (uint8_t*)out;
    memset(img, 0, im.total_bytes);
    memcpy(img, &im.hdr, sizeof(im.hdr));
    memcpy(img + im.table_off, im.table.data(), im.table.size() * sizeof(pnr_chunk_entry));
    for (size_t i = 0; i < im.frags.size(); ++i) fill_fragment_host(im.frags[i], precision, img + im.data_off + i * PNR_FRAG_BYTES);
}

// ------------------------------------------------------------------------------- host API
PNR_EXPORT int64_t pnr_mlp_packed_bytes(const pnr_mlp_desc* desc)
{
    if (pnr_mlp_validate(desc) != PNR_OK) return PNR_EINVAL;
    if (desc->plan == 2) pnr_mlp_tt_prepare_quiet();     // (a plan-2 image is about to exist: have its kernel loaded before any capture)
    PnrPlan plan;
    pnr_build_plan(*desc, plan);
    return (int64_t)plan.total_bytes;
}

PNR_EXPORT int pnr_mlp_pack(const pnr_mlp_desc* desc, const pnr_mlp_params_host* p, void* packed_host)
{
    int rc = pnr_mlp_validate(desc);
    if (rc != PNR_OK) return rc;
    PNR_REQUIRE(packed_host, "pnr_mlp_pack: null pointer");
    if ((rc = check_params(desc, p, true)) != PNR_OK) return rc;
    if (desc->plan == 2) pnr_mlp_tt_prepare_quiet();
    Image im;
    describe_forward(*desc, *p, im);
    write_host(im, desc->precision, packed_host);
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
    PNR_REQUIRE(packed_host, "pnr_mlp_pack_bwd: null pointer");
    if ((rc = check_params(desc, p, false)) != PNR_OK) return rc;
    Image im;
    describe_backward(*desc, *p, im);
    write_host(im, PNR_PREC_BF16, packed_host);
    return PNR_OK;
}

// ------------------------------------------------------------------------------- device packer
// One 64-lane workgroup per fragment; lane l fills its own 16 bytes (the kernel's ds_read_b128 unit).
__global__ __launch_bounds__(64) void k_pack_fragments(const PnrFragDesc* __restrict__ descs, int n_frags, int precision,
                                                       uint8_t* __restrict__ data)
{
    const int lane = threadIdx.x;
    for (int fi = blockIdx.x; fi < n_frags; fi += gridDim.x) {
        const PnrFragDesc f = descs[fi];
        uint32_t w[4];
        if (f.kind == PNR_F_BIAS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = __float_as_uint(pnr_bias_value(f, lane * 4 + j));
        } else if (precision == PNR_PREC_BF16) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = (uint32_t)pnr_f32_to_bf16(pnr_frag_value(f, 8, lane, 2 * j)) |
                       ((uint32_t)pnr_f32_to_bf16(pnr_frag_value(f, 8, lane, 2 * j + 1)) << 16);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = __float_as_uint(pnr_frag_value(f, 4, lane, j));
        }
        *reinterpret_cast<uint4*>(data + (size_t)fi * PNR_FRAG_BYTES + lane * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

static int describe(const pnr_mlp_desc* desc, const pnr_mlp_params_host* p, int backward, Image& im)
{
    int rc = backward ? bwd_validate(desc) : pnr_mlp_validate(desc);
    if (rc != PNR_OK) return rc;
    if ((rc = check_params(desc, p, !backward)) != PNR_OK) return rc;
    if (backward) describe_backward(*desc, *p, im); else describe_forward(*desc, *p, im);
    return PNR_OK;
}

PNR_EXPORT int64_t pnr_mlp_pack_workspace_bytes(const pnr_mlp_desc* desc, int backward)
{
    const int64_t total = backward ? pnr_mlp_bwd_packed_bytes(desc) : pnr_mlp_packed_bytes(desc);
    if (total < 0) return total;
    return (total / PNR_FRAG_BYTES + 1) * (int64_t)sizeof(PnrFragDesc);      // >= one descriptor per fragment
}

PNR_EXPORT int pnr_mlp_pack_device(const pnr_mlp_desc* desc, const pnr_mlp_params_host* params_dev, int backward,
                                   void* workspace, void* packed, void* stream)
{
    PNR_REQUIRE(workspace && packed, "pnr_mlp_pack_device: null pointer");
    if (desc && !backward && desc->plan == 2) {      // the two-tile kernel's code object: loaded here, outside any stream capture
        int rc0 = pnr_mlp_tt_prepare();
        if (rc0 != PNR_OK) return rc0;
    }
    Image im;
    int rc = describe(desc, params_dev, backward, im);
    if (rc != PNR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    // header + chunk table, then the descriptors: small host-to-device copies ordered on `st` before the kernel.
    // Their sources are this call's local vectors, so the stream is drained once before they go out of scope (the
    // runtime happens to stage pageable copies before returning, but that is not documented behaviour).  This is the
    // one entry point that synchronises -- setup, outside graph capture; pnr_mlp_repack_device is the sync-free form.
    PNR_HIP(hipMemcpyAsync(packed, &im.hdr, sizeof(im.hdr), hipMemcpyHostToDevice, st));
    PNR_HIP(hipMemcpyAsync((uint8_t*)packed + im.table_off, im.table.data(), im.table.size() * sizeof(pnr_chunk_entry),
                           hipMemcpyHostToDevice, st));
    PNR_HIP(hipMemcpyAsync(workspace, im.frags.data(), im.frags.size() * sizeof(PnrFragDesc), hipMemcpyHostToDevice, st));
    const int n = (int)im.frags.size();
    hipLaunchKernelGGL(k_pack_fragments, dim3(n < 2048 ? n : 2048), dim3(64), 0, st, (const PnrFragDesc*)workspace, n,
                       backward ? PNR_PREC_BF16 : desc->precision, (uint8_t*)packed + im.data_off);
    PNR_CHECK_LAUNCH("pnr_mlp_pack_device");
    PNR_HIP(hipStreamSynchronize(st));
    return PNR_OK;
}

// Re-run ONLY the packing kernel: the descriptors in `workspace` and the header / chunk table in `packed` are those a
// previous pnr_mlp_pack_device call with the SAME desc, parameter pointers and buffers left there.  No host-to-device
// copy, so this form is graph-capture safe -- it is what lets a whole training step (render, losses, backward,
// optimiser, repack) be captured into one HIP graph and replayed.
PNR_EXPORT int pnr_mlp_repack_device(const pnr_mlp_desc* desc, const pnr_mlp_params_host* params_dev, int backward,
                                     void* workspace, void* packed, void* stream)
{
    PNR_REQUIRE(workspace && packed, "pnr_mlp_repack_device: null pointer");
    Image im;
    int rc = describe(desc, params_dev, backward, im);
    if (rc != PNR_OK) return rc;
    const int n = (int)im.frags.size();
    hipLaunchKernelGGL(k_pack_fragments, dim3(n < 2048 ? n : 2048), dim3(64), 0, (hipStream_t)stream, (const PnrFragDesc*)workspace, n,
                       backward ? PNR_PREC_BF16 : desc->precision, (uint8_t*)packed + im.data_off);
    PNR_CHECK_LAUNCH("pnr_mlp_repack_device");
    return PNR_OK;
}
