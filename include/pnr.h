/*
 * pnr.h -- C-ABI of libpnr.so, the MI355X (gfx950) implementation of PanopticNeRF's
 * render_rays hot path (BASELINE.json north_star).
 *
 * Which reference interface each entry point replaces.  The mounted reference
 * (/root/reference) contains only README.md -- README.md:7 / README.md:13 name the code
 * branches (`panopticnerf360`, `panopticnerf`) that hold lib/networks/renderer and are NOT
 * in the mount -- so no file:line can be cited for the functions themselves.  Each entry
 * below names the reference function (as BASELINE.json's north_star names it) and the
 * SURVEY.md section-8a row that specifies its arithmetic:
 *
 *   pnr_stratified      render_rays' stratified z sampler                 (8a row a3)
 *   pnr_points          pts = o + d*z                                     (8a row a3)
 *   pnr_embed           Embedder / get_embedder                           (8a row a4)
 *   pnr_mlp_*           Network (NeRF 8x256 MLP + semantic/instance heads) (8a row a5)
 *   pnr_mlp_forward_train / pnr_mlp_backward   autograd of the Network   (8a row a9)
 *   pnr_composite       raw2outputs (+ panoptic logit / fixed-field maps) (8a row a6)
 *   pnr_composite_backward   autograd backward of raw2outputs            (8a row a9)
 *   pnr_sample_pdf      sample_pdf + sorted merge with the coarse z       (8a row a7)
 *   pnr_bbox_hits       ray / 3D-bbox intersection (bbox prior)           (8a row a8)
 *   pnr_sample_labels   per-sample fixed semantic / instance labels        (8a row a8)
 *   pnr_ray_setup       a8 + a3 + a8 of the coarse level in one launch    (8a rows a3, a8)
 *   pnr_sample_pdf_labels   a7 + a8 of the fine level in one launch       (8a rows a7, a8)
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - the caller owns every buffer; the library never allocates, frees or synchronises;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default);
 *   - int return: 0 = PNR_OK, negative = error; pnr_last_error() gives the text
 *     (thread-local);  no C++ exception crosses the ABI;
 *   - re-entrant; graph-capture safe (no hipMalloc / sync inside).
 * Arrays are dense row-major fp32 / int32 unless a stride is given.
 */
#ifndef PNR_H
#define PNR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNR_OK 0
#define PNR_EINVAL (-1)  /* bad argument (size, alignment, unsupported configuration) */
#define PNR_EHIP (-2)    /* a HIP runtime call failed; see pnr_last_error() */
#define PNR_ENODEV (-3)  /* no gfx950 device */

#define PNR_PREC_BF16 0  /* bf16 MFMA, fp32 accumulate (v_mfma_f32_32x32x16_bf16) */
#define PNR_PREC_FP32 1  /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32), parity mode */

int pnr_version(void);
const char* pnr_last_error(void);
/* Name of device `dev` copied to buf (host). Returns PNR_ENODEV if it is not gfx950. */
int pnr_device_check(int dev, char* buf_host, int buflen);

/* ---- a3: stratified sampler.  rays (R,8) = o(3) d(3) near far.  t_rand (R,N) or NULL
 * (perturb == 0).  z_out (R,N).  Bit-exact with oracle/pnr_oracle.c:pnro_stratified. */
int pnr_stratified(const float* rays, int64_t n_rays, int n_samples, int lindisp,
                   const float* t_rand, float* z_out, void* stream);

/* pts_out (R,N,3) = o + d*z.  Bit-exact with pnro_points. */
int pnr_points(const float* rays, const float* z, int64_t n_rays, int n_samples, float* pts_out,
               void* stream);

/* ---- a4: Embedder.  x (n,3) -> out (n, 3+6L), [x, sin(2^k x), cos(2^k x)]_k. */
int pnr_embed(const float* x, int64_t n, int L, float* out, void* stream);

/* ---- a5: fused NeRF MLP + heads.
 * Geometry of one network.  Trunk: D layers of width W (W = 128 or 256), skip-concat of
 * gamma(x) after layer `skip` (-1 = none).  sigma, feature(W), views(W+dir -> W/2), rgb.
 * Optional heads W -> head_W -> n_sem / n_inst (0 = absent). */
typedef struct pnr_mlp_desc {
    int32_t D, W, skip;
    int32_t xyz_L, dir_L;
    int32_t n_sem, n_inst, head_W;
    int32_t precision; /* PNR_PREC_* */
    int32_t plan;      /* chunk order of the packed image: 0 = classic (every kernel); 1 = fused-inference order, 2 = two-tile order,
                          see pnr_mlp_fused_plan (only pnr_mlp_forward_composite accepts them) */
    int32_t head_tap;  /* what the semantic / instance heads read: 0 = the trunk output h (default), 1 = the feature_linear
                          output (the reference's tap point cannot be checked here, so it is a switch) */
    int32_t head_depth;/* 0 or 2 = W -> head_W -> n (ReLU between; default), 1 = one Linear W -> n */
    int32_t schedule;  /* time structure of the bf16 weight stream (tests and A/B tools; the arithmetic, the packed image and the
                          results are identical bit for bit): 0 = default (ping-pong k_mlp_pp for inference launches, lock-step
                          k_mlp_fused for the training forward), 1 = lock-step everywhere, 2 = ping-pong everywhere */
    int32_t clk_probe[2]; /* diagnostics (libpnr_bench.so, tools/): low / high 32 bits of a 16-byte aligned DEVICE address of 16 bytes;
                             when non-zero the forward MLP kernels launched with this descriptor write {shader cycles, 100 MHz ticks}
                             of workgroup 0's first wave there (their ratio = the mean shader clock during the launch).  0 = off.
                             A descriptor field, not a setter: the library keeps no mutable state (round 5) */
    int32_t flags;     /* PNR_MLP_* bits, 0 by default */
} pnr_mlp_desc;
/* ZERO-INITIALISE the whole descriptor (memset / = {0}) before setting fields: clk_probe and flags are READ by every entry point
 * that takes a descriptor -- a stale clk_probe is a device address the forward kernels store to.  Every pnr_mlp_* call rejects
 * (PNR_EINVAL) flags with undefined bits and a clk_probe that is not 16-byte aligned. */

#define PNR_MLP_SOFTMAX 1      /* pnr_mlp_forward_composite / pnr_mlp_forward_tiles composite softmax(logits) over each learned field's
                                  channels instead of the logits (the reference's semantic_activation = softmax; pnr_composite's
                                  sem_mode 1).  Needs an image whose plan has a softmax kernel -- a head's logit blocks must be in
                                  registers together: plan 2 (the k_mlp_tt_*sm_* kernels, round 6) or plan 1; ask
                                  pnr_mlp_fused_plan WITH this flag set in desc.flags (0 = none: use pnr_mlp_forward + pnr_composite);
                                  PNR_EINVAL otherwise */
#define PNR_MLP_WG_CAP(n) (((n) & 0x1FF) << 16)  /* plan 2 (pnr_mlp_forward_composite / _tiles): launch on at most n workgroups (= compute units;
                                  0 = all of them).  Two launches with caps that add up to the device can run SIDE BY SIDE on two streams --
                                  one level of a chunk beside the other level of the next: Renderer's cfg.overlap_levels.  Same results. */
#define PNR_MLP_TRACE 0x7A00   /* diagnostics BUILDS of the library only (make EXTRA_TT=trace | abl; the shipped library refuses the
                                  flag), with plan 2: the trace build of k_mlp_tt -- clk_probe must then address (64 + workgroups) * 4
                                  bytes (workgroups <= number of CUs; tools/tt_trace.py allocates 1280): 64 per-unit s_memtime stamps
                                  of workgroup 0's first wave, then every workgroup's cycles.  + (a << 4), a in 1..7: its timing-only
                                  ablation a (bits 4..6 -- bit 0 stays PNR_MLP_SOFTMAX) */

/* Dense fp32 parameters in HOST memory, row-major (out,in), nn.Linear convention.
 * pts_w[i]/pts_b[i] for i < D.  Head pointers may be NULL when the head is absent; with head_depth = 1 a head is the single
 * Linear sem1_w / inst1_w of shape (n, W) and sem0_* / inst0_* are NULL. */
typedef struct pnr_mlp_params_host {
    const float* const* pts_w; const float* const* pts_b;
    const float* alpha_w; const float* alpha_b;
    const float* feature_w; const float* feature_b;
    const float* views_w; const float* views_b;
    const float* rgb_w; const float* rgb_b;
    const float* sem0_w; const float* sem0_b; const float* sem1_w; const float* sem1_b;
    const float* inst0_w; const float* inst0_b; const float* inst1_w; const float* inst1_b;
} pnr_mlp_params_host;

/* Bytes of the packed (MFMA-fragment-ordered) parameter image for `desc`; <0 on error. */
int64_t pnr_mlp_packed_bytes(const pnr_mlp_desc* desc);
/* Pack host parameters into packed_host (pnr_mlp_packed_bytes bytes, host). Pure CPU. */
int pnr_mlp_pack(const pnr_mlp_desc* desc, const pnr_mlp_params_host* params, void* packed_host);

/* Device-side packer: same image as pnr_mlp_pack (backward = 0) / pnr_mlp_pack_bwd (backward = 1), written
 * straight from the live parameter tensors on the GPU -- no host round trip per optimiser step.
 * params_dev: the struct itself (and its pts_w / pts_b pointer arrays) in HOST memory, every pointer in it a
 * DEVICE pointer.  workspace: device scratch of pnr_mlp_pack_workspace_bytes bytes (fragment descriptors).
 * packed: device buffer of pnr_mlp_packed_bytes / pnr_mlp_bwd_packed_bytes bytes.  Three small host->device
 * copies (header, chunk table, descriptors) and one kernel are enqueued on `stream`, and `stream` is synchronised
 * once before returning (the copies read call-local host staging memory): the ONE entry point that synchronises;
 * it is set-up work -- never call it inside graph capture, use pnr_mlp_repack_device there. */
int64_t pnr_mlp_pack_workspace_bytes(const pnr_mlp_desc* desc, int backward);
int pnr_mlp_pack_device(const pnr_mlp_desc* desc, const pnr_mlp_params_host* params_dev, int backward,
                        void* workspace, void* packed, void* stream);
/* The packing kernel alone, for parameters that CHANGED IN PLACE since a pnr_mlp_pack_device call with the same desc,
 * parameter pointers, workspace and packed buffer (an optimiser step): no host-to-device copy, graph-capture safe. */
int pnr_mlp_repack_device(const pnr_mlp_desc* desc, const pnr_mlp_params_host* params_dev, int backward,
                          void* workspace, void* packed, void* stream);

/* Evaluate the network on every sample of every ray:
 *   pts = o + d*z, viewdir = d/||d||, raw = MLP(gamma(pts), gamma(viewdir)).
 * packed: device copy of the pnr_mlp_pack image.  rays (R,8), z (R,N).
 * raw element (sample s = r*N+i, channel c) is written at raw[s*raw_stride_s + c*raw_stride_c],
 * channels = [r g b sigma | n_sem | n_inst].  Channel-major (stride_s=1, stride_c=R*N) is the
 * fast layout; sample-major (stride_s=4+n_sem+n_inst, stride_c=1) is the reference's. */
int pnr_mlp_forward(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z,
                    int64_t n_rays, int n_samples, float* raw, int64_t raw_stride_s,
                    int64_t raw_stride_c, void* stream);

/* ---- a5 + a6 fused (inference): evaluate the network and composite in ONE pass -- the raw image (4 + n_sem + n_inst floats per
 * sample) never goes to HBM.  The fused MLP's epilogue keeps per 32-sample tile one record (csrc/pnr_mlp_fuse.h: transmittance
 * factor and the weighted logit sums) and per sample (local weight, raw r, g, b): 26 B per sample instead of 324 at 45 / 32
 * heads; a second small kernel finishes each ray from them (acc / depth / rgb sums, the fixed fields, the logits).  Replaces the pair
 * pnr_mlp_forward + pnr_composite (same reference rows: render_rays' network call + raw2outputs, /root/reference/README.md:13
 * points to the branch that holds them) under: bf16, logits compositing (sem_mode 0), no sigma noise, n_samples a
 * multiple of 32 in [32, 256], n_sem + n_inst <= 128.  Results equal the two-kernel path to fp32 rounding (the sums are
 * associated per tile).  Outputs as pnr_composite's (any may be null; fix_* need their labels); weights (R,N) optional.
 * workspace: pnr_mlp_forward_composite_workspace_bytes(desc, n_rays, n_samples, weights != null) device bytes (tile records,
 * per-sample quadruples and 128 B per ray: |d| and gamma(d / |d|) once per ray, written by a pre-kernel for the plan-2 kernel --
 * which takes at most 2^24 rays per call). */
/* Chunk order for images that only pnr_mlp_forward_composite will consume: the BEST plan `desc`'s geometry has.
 *   1: the fused-inference plan (bf16, W = 256, 1..2 semantic and 0..1 instance logit blocks of 32): the appearance branch, then
 *      BOTH head hidden layers, then the two logit layers as ONE chunk (k_mlp_pp: 8 waves, one 32-sample tile per wave);
 *   2: the two-tile plan (the 8 x 256 network of the BASELINE configs: D = 8, skip = 4, L = 10 / 4, head_tap 0; no heads, a
 *      semantic head of up to 64 classes, that plus an instance head of up to 32, or a semantic head of 65..96 classes alone; any head_tap / head_depth): no chunk above 33 fragments, consumed by
 *      k_mlp_tt -- hand-placed gfx950 assembly, one wave per SIMD, two tiles per wave, every LDS weight fragment feeds two MFMAs
 *      (csrc/asm/gen_mlp_tt.py);
 *   0: the classic order, which every entry point accepts.
 * Set desc.plan to the returned value (or to a smaller supported one: plan 1 needs a semantic head) before pnr_mlp_packed_bytes / pnr_mlp_pack* and keep it
 * for the forward call.  Same arithmetic per layer under every plan: records and maps are bit-identical (head_depth 1: plan 2 against
 * plan 0 to fp32 rounding).  With PNR_MLP_SOFTMAX in desc.flags the answer is the best plan that has a SOFTMAX kernel (2, 1, or 0 = none). */
int pnr_mlp_fused_plan(const pnr_mlp_desc* desc);
int64_t pnr_mlp_forward_composite_workspace_bytes(const pnr_mlp_desc* desc, int64_t n_rays, int n_samples, int want_weights);
int pnr_mlp_forward_composite(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z,
                              int64_t n_rays, int n_samples, const int32_t* label_sem, const int32_t* label_inst,
                              int white_bkgd, float* rgb, float* depth, float* acc, float* weights, float* sem,
                              float* inst, float* fix_sem, float* fix_inst, void* workspace, void* stream);

/* The two halves of pnr_mlp_forward_composite as separate calls (it is exactly these two, in this order, on one workspace):
 * pnr_mlp_forward_tiles runs the fused MLP -- per 32-sample tile one record (transmittance factor, logit sums), per sample
 * (local weight, raw r, g, b) -- and pnr_composite_combine finishes every ray from them (k_composite_combine: one wave per ray).
 * Same arguments and conditions as pnr_mlp_forward_composite. */
int pnr_mlp_forward_tiles(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z, int64_t n_rays,
                          int n_samples, void* workspace, void* stream);
int pnr_composite_combine(const pnr_mlp_desc* desc, const void* workspace, const float* z, int64_t n_rays, int n_samples,
                          const int32_t* label_sem, const int32_t* label_inst, int white_bkgd, float* rgb, float* depth,
                          float* acc, float* weights, float* sem, float* inst, float* fix_sem, float* fix_inst, void* stream);

/* ---- a9 (training): forward that also saves what the backward needs, the data-gradient pass, and the
 * buffer layouts.  bf16 only; n_sem, n_inst <= 64.
 *   acts : bf16, pnr_mlp_train_layout's acts_off[D+6] elements -- gamma(x), gamma(d) and every layer's output, one
 *          slot-ordered region of S_pad x width per tensor (S = n_rays*n_samples, S_pad = S rounded up to 256), followed by
 *          one gate BIT per element of every ReLU output (what pnr_mlp_backward reads);
 *   dys  : bf16, dys_off[D+7] elements -- every layer's pre-activation gradient dY, same layout (plus the output
 *          layers' dY = d_raw in bf16: [rgb,sigma] in 32 slots, semantic and instance logits in 64 slots each), written by
 *          pnr_mlp_backward (rows S..S_pad: zeros) for pnr_mlp_wgrad's dW = dY^T X;
 *   d_raw: (4+n_sem+n_inst, S) channel-major fp32 (pnr_composite_backward's output).
 * Slot order (csrc/pnr_mlp_layout.h): slot fb*32 + hi*16 + r <-> feature fb*32 + (r&3) + 8*(r>>2) + 4*hi.
 * Saved-tensor layout of a region (csrc/pnr_mlp_layout.h, pnr_saved_chunk): the 16-byte chunk c = slot/8 of sample s lives
 * in the 128-byte line [s >> 3][c] at position (s & 7) ^ (4 * ((c >> 1) & 1)) -- a wave's store writes full lines and a
 * 64-sample tile is one contiguous block for the weight-gradient kernel.  The buffers are opaque to callers. */
int pnr_mlp_train_layout(const pnr_mlp_desc* desc, int64_t n_samples, int64_t* acts_off_host /* D+7 */,
                         int64_t* dys_off_host /* D+8 */);
int pnr_mlp_forward_train(const pnr_mlp_desc* desc, const void* packed, const float* rays, const float* z,
                          int64_t n_rays, int n_samples, float* raw, int64_t raw_stride_s,
                          int64_t raw_stride_c, void* acts, void* stream);
int64_t pnr_mlp_bwd_packed_bytes(const pnr_mlp_desc* desc);
int pnr_mlp_pack_bwd(const pnr_mlp_desc* desc, const pnr_mlp_params_host* params, void* packed_host);
int pnr_mlp_backward(const pnr_mlp_desc* desc, const void* packed_bwd, const float* d_raw, const void* acts,
                     void* dys, int64_t n_rays, int n_samples, void* stream);

/* Weight gradients (the third K3 kernel): for every Linear of the network
 *     dW = dY^T X   and   db = sum over samples of dY,
 * computed from `acts` (pnr_mlp_forward_train) and `dys` (pnr_mlp_backward) of the same n_samples = n_rays * n_samples.
 * grads_dev: a pnr_mlp_params_host whose pointers are DEVICE pointers to the fp32 GRADIENT buffers, one per parameter,
 *   nn.Linear layout ((out,in) row-major / (out)); every one of them is fully overwritten (not accumulated).  The struct
 *   and its pts_w / pts_b arrays live in host memory.
 * workspace: device scratch of pnr_mlp_wgrad_workspace_bytes bytes (per-slab partial sums; deterministic reduction).
 * Replaces autograd's per-layer  grad_output.t() @ input  and  grad_output.sum(0)  for nn.Linear. */
int64_t pnr_mlp_wgrad_workspace_bytes(const pnr_mlp_desc* desc, int64_t n_samples);
int pnr_mlp_wgrad(const pnr_mlp_desc* desc, const void* acts, const void* dys, int64_t n_samples,
                  const pnr_mlp_params_host* grads_dev, void* workspace, void* stream);

/* ---- a9, fp32 PARITY MODE of the training path (precision = "fp32" with autograd): the same three steps in plain fp32 -- the
 * mode in which a parameter gradient can be checked to 1e-4 against fp32 autograd of the reference arithmetic, and in which a
 * user can tell the precision of the bf16 path from a defect.  Not a performance path (one generic strided fp32 GEMM kernel per
 * Linear and direction, deterministic fixed-order reductions).  Parameters are NOT packed: params_dev is a pnr_mlp_params_host
 * in host memory whose pointers are DEVICE pointers to the dense (out,in) row-major fp32 parameters (the nn.Parameter tensors
 * themselves), as for pnr_mlp_pack_device.  head_depth 1 | 2 and head_tap 0 | 1 are supported; desc.precision is ignored.
 *   acts      : fp32, pnr_mlp_fp32_acts_floats(desc, n_rays * n_samples) floats -- gamma(x), gamma(d), every layer's output,
 *               dense [S][width] (opaque to callers);
 *   raw       : as pnr_mlp_forward (any strides);   d_raw: channel-major (4+n_sem+n_inst, >= S) fp32, channel stride given;
 *   grads_dev : like pnr_mlp_wgrad's -- DEVICE pointers to the fp32 gradient buffers, every one fully overwritten;
 *   workspace : pnr_mlp_backward_fp32_workspace_bytes device bytes.
 * Replaces torch autograd of the reference's Network for the fp32 case. */
int64_t pnr_mlp_fp32_acts_floats(const pnr_mlp_desc* desc, int64_t n_samples);
int64_t pnr_mlp_backward_fp32_workspace_bytes(const pnr_mlp_desc* desc, int64_t n_samples);
int pnr_mlp_forward_train_fp32(const pnr_mlp_desc* desc, const pnr_mlp_params_host* params_dev, const float* rays,
                               const float* z, int64_t n_rays, int n_samples, float* raw, int64_t raw_stride_s,
                               int64_t raw_stride_c, float* acts, void* stream);
int pnr_mlp_backward_fp32(const pnr_mlp_desc* desc, const pnr_mlp_params_host* params_dev, const float* d_raw,
                          int64_t d_raw_stride_c, const float* acts, int64_t n_rays, int n_samples,
                          const pnr_mlp_params_host* grads_dev, void* workspace, void* stream);

/* ---- a6: raw2outputs.  raw strides as above.  noise (R,N) or NULL; label_* (R,N) int32 or
 * NULL (fixed bbox-prior field, -1 = none).  sem_mode 0: composite logits; 1: softmax first.
 * Any output pointer may be NULL.  weights (R,N); rgb (R,3); depth (R); acc (R);
 * sem/fix_sem (R,n_sem); inst/fix_inst (R,n_inst).  n_samples % 4 == 0, n_samples <= 256. */
int pnr_composite(const float* raw, int64_t raw_stride_s, int64_t raw_stride_c, const float* z,
                  const float* rays, const float* noise, const int32_t* label_sem,
                  const int32_t* label_inst, int64_t n_rays, int n_samples, int n_sem, int n_inst,
                  int sem_mode, int white_bkgd, float* rgb, float* depth, float* acc, float* weights,
                  float* sem, float* inst, float* fix_sem, float* fix_inst, void* stream);

/* ---- a9 (backward of a6): gradient of the composited maps w.r.t. raw.  Channel-major images only
 * (raw_stride_s == 1).  g_* are the upstream gradients of the forward outputs of the same name
 * (any may be NULL = zero); g_weights (R,N) is the gradient of the weights output.  d_raw has raw's
 * shape and layout.  sem_mode 0 (logit compositing) only; no gradient flows into z (sample_pdf's
 * output is detached in the reference). */
int pnr_composite_backward(const float* raw, int64_t raw_stride_c, const float* z, const float* rays,
                           const float* noise, int64_t n_rays, int n_samples, int n_sem, int n_inst,
                           const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_sem,
                           const float* g_inst, const float* g_weights, float* d_raw, void* stream);

/* pnr_composite_backward plus the two gradient sources the trainer's loss wrapper adds:
 *   g_fix_sem (R,n_sem) / g_fix_inst (R,n_inst): gradients of the FIXED (bbox-prior) maps; since
 *     fix_x[c] = sum_i w_i [label_i == c], they reach the densities through the weights: dL/dw_i += g_fix_x[label_i];
 *   ce_sem / ce_inst: DEVICE scalars s; adds s * (softmax_c(raw logits of sample) - [c == label]) to d_raw for every
 *     sample with a label >= 0 -- the gradient of the per-sample 3D cross-entropy whose value pnr_ce3d computes
 *     (s = upstream gradient * loss weight / number of labelled samples).
 * label_sem / label_inst (R,N) int32 are the labels pnr_sample_labels produced for this level. */
int pnr_composite_backward2(const float* raw, int64_t raw_stride_c, const float* z, const float* rays,
                            const float* noise, int64_t n_rays, int n_samples, int n_sem, int n_inst,
                            const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_sem,
                            const float* g_inst, const float* g_weights, const int32_t* label_sem,
                            const int32_t* label_inst, const float* g_fix_sem, const float* g_fix_inst,
                            const float* ce_sem, const float* ce_inst, float* d_raw, void* stream);
/* ... and with pnr_composite's sem_mode: 1 = the semantic / instance maps composite softmax(logits) per sample
 * (g_sem / g_inst are then gradients of probability maps):  d x_{i,c} = w_i s_c (g_c - sum_k g_k s_k),
 * dL/dw_i += sum_k g_k s_k. */
int pnr_composite_backward3(const float* raw, int64_t raw_stride_c, const float* z, const float* rays,
                            const float* noise, int64_t n_rays, int n_samples, int n_sem, int n_inst, int sem_mode,
                            const float* g_rgb, const float* g_depth, const float* g_acc, const float* g_sem,
                            const float* g_inst, const float* g_weights, const int32_t* label_sem,
                            const int32_t* label_inst, const float* g_fix_sem, const float* g_fix_inst,
                            const float* ce_sem, const float* ce_inst, float* d_raw, void* stream);

/* ---- 8f-1: the trainer's loss wrapper (the reference's NetworkWrapper) on the maps of
 * one level, fused with the gradient of the weighted total w.r.t. every map.  All reductions are means:
 *   [0] rgb      mean over rays and channels of (rgb - rgb_gt)^2
 *   [1] depth    mean over rays with depth_gt > 0 of |depth - depth_gt|  (depth_l2: squared error)
 *   [2] sem      mean over rays with 0 <= sem_gt < n_sem of CE(softmax(sem), sem_gt)      (learned field, 2D pseudo label)
 *   [3] fix_sem  mean over the same rays of -log(fix_sem[sem_gt] + fix_eps)                (fixed bbox-prior field)
 *   [4] inst, [5] fix_inst: the same two terms for the instance field
 *   [6] total = sum of w_x * term_x;  losses_out is 8 floats on the DEVICE.
 * Any map / target / gradient pointer may be NULL (its term is skipped / its gradient not written).  g_x has the
 * shape of map x and receives d total / d x.  workspace: pnr_losses_workspace_bytes(n_rays) device bytes.
 * Replaces ~30 eager torch launches per level (mse_loss, l1_loss, cross_entropy, nll_loss and their backwards). */
typedef struct pnr_loss_cfg {
    float w_rgb, w_depth, w_sem, w_fix_sem, w_inst, w_fix_inst;
    int32_t depth_l2;
    float fix_eps;
    int32_t maps_are_prob;   /* 1: sem / inst are composited PROBABILITIES (pnr_composite sem_mode 1): their 2D term is
                                -log(map[label] + fix_eps) like the fixed field's, not a softmax cross-entropy */
} pnr_loss_cfg;
int64_t pnr_losses_workspace_bytes(int64_t n_rays);
int pnr_losses(const pnr_loss_cfg* cfg, int64_t n_rays, int n_sem, int n_inst, const float* rgb, const float* depth,
               const float* sem, const float* fix_sem, const float* inst, const float* fix_inst, const float* rgb_gt,
               const float* depth_gt, const int32_t* sem_gt, const int32_t* inst_gt, float* losses_out, float* g_rgb,
               float* g_depth, float* g_sem, float* g_fix_sem, float* g_inst, float* g_fix_inst, void* workspace, void* stream);

/* Per-sample 3D cross-entropy (forward value) of the learned logits raw[first_channel .. +n_classes) (channel-major,
 * sample stride 1) against label (n_samples) int32, -1 = unlabelled: out2 (device) = {mean CE over labelled samples,
 * their count}.  Gradient: pnr_composite_backward2's ce_sem / ce_inst. */
int64_t pnr_ce3d_workspace_bytes(int64_t n_samples);
int pnr_ce3d(const float* raw, int64_t raw_stride_c, int first_channel, int n_classes, const int32_t* label,
             int64_t n_samples, float* out2, void* workspace, void* stream);

/* ---- 8f-2: ray generation, the dataset-side producer of batch['rays'] (the reference builds rays from the KITTI-360
 * intrinsics and poses in lib/datasets/kitti360/, not in the mount).  intr4_host = {fx, fy, cx, cy} and c2w12_host =
 * 3x4 row-major camera-to-world [R | t] (x right, y down, z forward) are HOST arrays (copied into the launch).
 * Pixel (i = column, j = row): d = R * ((i - cx)/fx, (j - cy)/fy, 1), o = t; d is not normalised.
 * pix (n_rays) int32 linear pixel indices j*width + i on the device, or NULL = the whole frame (n_rays = width*height).
 * rays (n_rays, 8) = o d near far.  Bit-exact with pnro_gen_rays. */
int pnr_gen_rays(const float* intr4_host, const float* c2w12_host, int width, int height, float near_, float far_,
                 const int32_t* pix, int64_t n_rays, float* rays, void* stream);

/* ---- 8f-4: label-map post-processing and evaluator counters (what follows the path in the reference's evaluate loop;
 * its evaluator is not in the mount, conventions are this build's -- DESIGN.md 8).
 * pnr_panoptic_labels: sem_label = argmax_c sem (lowest index on ties); inst_label = argmax_k inst where is_thing[sem_label]
 *   != 0 (is_thing NULL: every class), else -1; panoptic = class*1000 + instance on things, class on stuff.  Any output
 *   may be NULL.  sem (R,n_sem), inst (R,n_inst) or NULL, is_thing (n_sem) int32 device or NULL.
 * pnr_confusion: conf[gt*n_classes + pred] += 1 over pixels with both labels in [0, n_classes) (gt < 0 = ignore); conf is a
 *   device (n_classes^2) int64 array the caller zeroes once and accumulates into over frames.  Integer atomics: exact.
 *   n_classes <= 8192 (above 128 without the per-block LDS histogram: used for the segment-pair counts of PQ).
 *   mIoU / accuracy are a handful of flops on that matrix (host side); PSNR = -10 log10 of pnr_losses' rgb term. */
int pnr_panoptic_labels(const float* sem, const float* inst, const int32_t* is_thing, int64_t n_rays, int n_sem, int n_inst,
                        int32_t* sem_label, int32_t* inst_label, int32_t* panoptic, void* stream);
int pnr_confusion(const int32_t* pred, const int32_t* gt, int64_t n, int n_classes, int64_t* conf, void* stream);

/* ---- a7: sample_pdf + merge.  z (R,Nc), weights (R,Nc) coarse; u (R,Nf) or NULL (det).
 * z_samples (R,Nf) and inds (R,Nf) int32 may be NULL; z_fine (R,Nc+Nf) sorted union or NULL.
 * Indices / z_samples bit-exact with pnro_sample_pdf.  Nc <= 256, Nc+Nf <= 512. */
int pnr_sample_pdf(const float* z, const float* weights, const float* u, int64_t n_rays, int n_coarse,
                   int n_fine, float* z_samples, int32_t* inds, float* z_fine, void* stream);

/* ---- a8: bbox prior.  box (M,15) = centre(3) rotation rows(9) half extents(3).
 * Per ray the max_hits NEAREST intersected boxes (smallest t_in; ties: lower box index), stored in ascending
 * (t_in, box index) order: hit_t (R,max_hits,2), hit_box (R,max_hits) int32 (-1 pad).  hit_count (R) int32 is the
 * TRUE number of intersected boxes: a value > max_hits reports that the farthest ones were dropped (grow max_hits);
 * pnr_sample_labels uses min(hit_count, max_hits) entries.  Bit-exact with pnro_bbox_hits. */
int pnr_bbox_hits(const float* rays, int64_t n_rays, const float* box, int n_box, int max_hits,
                  float* hit_t, int32_t* hit_box, int32_t* hit_count, void* stream);

/* Sampling restricted to the bbox prior (cfg.bbox_sampling = "hull"; SURVEY.md 9 item 2 -- whether the reference samples
 * [near, far] or the hit intervals cannot be checked here, so it is a switch): rays_out = rays with near / far replaced by the
 * hull [min t_in, max t_out] of the ray's kept intervals (min(hit_count, max_hits) entries of hit_t); rays without a hit are
 * copied unchanged.  rays_out may not alias rays.  Bit-exact with pnro_restrict_rays. */
int pnr_restrict_rays(const float* rays, int64_t n_rays, const float* hit_t, const int32_t* hit_count, int max_hits,
                      float* rays_out, void* stream);

/* box_ids (M,2) int32 = (semantic id, instance id).  label_* (R,N) int32. */
int pnr_sample_labels(const float* z, int64_t n_rays, int n_samples, const float* hit_t,
                      const int32_t* hit_box, const int32_t* hit_count, int max_hits,
                      const int32_t* box_ids, int32_t* label_sem, int32_t* label_inst, void* stream);

/* a8 + a3 (+ a8) in one launch -- the coarse level's per-ray preamble: hit lists exactly as pnr_bbox_hits writes them, z_out (R,N)
 * exactly as pnr_stratified (over the hull of each ray's kept intervals with hull != 0, as pnr_restrict_rays + pnr_stratified),
 * and, when label_sem / label_inst are given, the labels pnr_sample_labels would produce for z_out.  max_hits in [1, 8]
 * (larger lists: the separate entry points).  Replaces, with pnr_sample_pdf_labels, four of the reference render_rays' per-chunk
 * steps by one launch each. */
int pnr_ray_setup(const float* rays, int64_t n_rays, const float* box, int n_box, int max_hits, const int32_t* box_ids,
                  int n_samples, int lindisp, const float* t_rand, int hull, float* hit_t, int32_t* hit_box,
                  int32_t* hit_count, float* z_out, int32_t* label_sem, int32_t* label_inst, void* stream);

/* a7 + a8 in one launch: z_fine (R, n_coarse + n_fine) exactly as pnr_sample_pdf, and label_sem / label_inst (same shape)
 * exactly as pnr_sample_labels(z_fine, ...) -- the wave that merged a ray's samples labels them. */
int pnr_sample_pdf_labels(const float* z, const float* weights, const float* u, int64_t n_rays, int n_coarse, int n_fine,
                          float* z_fine, const float* hit_t, const int32_t* hit_box, const int32_t* hit_count,
                          int max_hits, const int32_t* box_ids, int32_t* label_sem, int32_t* label_inst, void* stream);

/* ---- diagnostics.  Measurement helpers (hipEvent timing, MFMA / HBM ceilings of the device) live in libpnr_bench.so
 * (include/pnr_bench.h), not here: every export of this library is stream-ordered, never synchronises and keeps no mutable
 * state -- the one diagnostic hook of the MLP kernels is a descriptor field (pnr_mlp_desc.clk_probe). */

#ifdef __cplusplus
}
#endif
#endif /* PNR_H */
