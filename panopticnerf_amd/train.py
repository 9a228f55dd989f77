"""Training-side glue of the render path: the autograd Function that
puts the HIP forward / backward kernels behind `Renderer.render`, and the flat-bucket gradient
all-reduce.

What runs where: forward (pnr_mlp_forward_train + pnr_composite), compositing backward
(pnr_composite_backward), the MLP data-gradient pass (pnr_mlp_backward) and the weight gradients
(pnr_mlp_wgrad: dW_l = dY_l^T X_l and the bias sums, MFMA over LDS transpose reads) are hand-written
HIP.  `weight_grads()` below is the same computation with library GEMMs (torch.bmm over sample slabs)
on the kernels' slot-ordered bf16 buffers; it is kept as the cross-check the GPU tests compare
pnr_mlp_wgrad against (tests/test_gpu_backward.py) and is not on the training path.
"""
import functools

import torch
import torch.distributed as dist

from . import ops


def _row_of(r, hi):
    return (r & 3) + 8 * (r >> 2) + 4 * hi


@functools.lru_cache(maxsize=None)
def feat_slots(width, device):
    """slot -> feature index of a slot-ordered width-`width` tensor (csrc/pnr_mlp_layout.h).  Cached: built with
    Python loops and one host->device copy, once per (width, device)."""
    idx = torch.empty(width, dtype=torch.long)
    for fb in range(width // 32):
        for hi in (0, 1):
            for r in range(16):
                idx[fb * 32 + hi * 16 + r] = fb * 32 + _row_of(r, hi)
    return idx.to(device)


@functools.lru_cache(maxsize=None)
def embed_slots(n_half_freq, L, device):
    """slot -> canonical gamma() column (or -1): EX (n_half_freq=5, 64 slots), ED (2, 32 slots).  Cached."""
    nv = 32 if n_half_freq == 5 else 16
    idx = torch.full((2 * nv,), -1, dtype=torch.long)
    for hi in (0, 1):
        for v in range(nv):
            if v == 0:
                c = 2 if hi else 0
            elif v == 1:
                c = -1 if hi else 1
            else:
                fp, j = divmod(v - 2, 6)
                f = hi * n_half_freq + fp
                c = 3 + 6 * f + j if (fp < n_half_freq and f < L) else -1
            idx[hi * nv + v] = c
    return idx.to(device)


_INV = {}


def _inverse(idx, n):
    """inv[c] = slot s with idx[s] == c, for c < n (every canonical column has exactly one slot).  Cached by identity."""
    key = (id(idx), n)
    inv = _INV.get(key)
    if inv is None:
        inv = torch.empty(n, dtype=torch.long, device=idx.device)
        ok = idx >= 0
        inv[idx[ok]] = torch.nonzero(ok).squeeze(1)
        _INV[key] = inv
    return inv


_OUT_DTYPE_OK = None      # does this torch build take out_dtype= on bmm (bf16 in, fp32 out)?


def _mm(a_t, b):
    """a_t (S, m), b (S, n), both bf16 -> a_t^T b (m, n) in fp32.
    The reduction dimension is S (10^5..10^6) while m, n <= 320: fed to the library as ONE GEMM it runs at
    ~80 TFLOP/s (no split-K); as a batched GEMM over S-slabs plus a sum it runs at ~500 TFLOP/s on MI355X
    (tools/wgrad_probe.py).  fp32 slab outputs when the build supports out_dtype, else bf16 slabs summed in fp32."""
    global _OUT_DTYPE_OK
    S, m = a_t.shape
    n = b.shape[1]
    nb = 1
    for cand in (256, 128, 64, 32, 16, 8, 4, 2):
        if S % cand == 0 and S // cand >= 512:
            nb = cand
            break
    A = a_t.reshape(nb, S // nb, m).transpose(1, 2)
    B = b.reshape(nb, S // nb, n)
    if _OUT_DTYPE_OK is not False:
        try:
            r = torch.bmm(A, B, out_dtype=torch.float32)
            _OUT_DTYPE_OK = True
            return r.sum(0)
        except (TypeError, RuntimeError):
            _OUT_DTYPE_OK = False
    return torch.bmm(A, B).float().sum(0)


def _sum0(x):
    return torch.sum(x, 0, dtype=torch.float32)


def weight_grads(nerf, desc, acts, dys, d_raw, S):
    """dict name -> fp32 gradient (nn.Linear layout) from the kernels' buffers."""
    dev = str(d_raw.device)
    D, W, H, C, K = nerf.D, nerf.W, nerf.W // 2, nerf.n_sem, nerf.n_inst
    ao, do = ops.train_layout(desc, S)
    A = lambda i, w: acts[ao[i]: ao[i] + S * w].view(S, w)
    Y = lambda i, w: dys[do[i]: do[i] + S * w].view(S, w)
    fW, fH = feat_slots(W, dev), feat_slots(H, dev)
    ex_idx, ed_idx = embed_slots(5, nerf.xyz_L, dev), embed_slots(2, nerf.dir_L, dev)
    EXn, EDn = 3 + 6 * nerf.xyz_L, 3 + 6 * nerf.dir_L

    def unperm_cols(g_slot, idx, n_cols):           # (m, slots) -> (m, n_cols): column c <- slot with idx == c
        return g_slot.index_select(1, _inverse(idx, n_cols))

    def unperm_rows(g, idx):                        # rows in slot order -> feature order
        return g.index_select(0, _inverse(idx, idx.numel()))

    X_h = A(1 + D, W)                               # h = X_D
    EX, ED = A(0, 64), A(1, 32)
    F_, G_, SHS, SHI = A(2 + D, W), A(3 + D, H), A(4 + D, H), A(5 + D, H)
    g = {}
    dr = d_raw.view(-1, S)
    drb = dr.t().to(torch.bfloat16).contiguous()    # (S, ch): the output layers' dY (already feature-ordered)
    # output layers
    g["rgb_linear.weight"] = unperm_cols(_mm(drb[:, 0:3].contiguous(), G_), fH, H)
    g["rgb_linear.bias"] = dr[0:3].sum(1)
    g["alpha_linear.weight"] = unperm_cols(_mm(drb[:, 3:4].contiguous(), X_h), fW, W)
    g["alpha_linear.bias"] = dr[3:4].sum(1)
    if C:
        g["semantic_linears.1.weight"] = unperm_cols(_mm(drb[:, 4:4 + C].contiguous(), SHS), fH, H)
        g["semantic_linears.1.bias"] = dr[4:4 + C].sum(1)
        dy = Y(2, H)
        g["semantic_linears.0.weight"] = unperm_rows(unperm_cols(_mm(dy, X_h), fW, W), fH)
        g["semantic_linears.0.bias"] = unperm_rows(_sum0(dy), fH)
    if K:
        g["instance_linears.1.weight"] = unperm_cols(_mm(drb[:, 4 + C:4 + C + K].contiguous(), SHI), fH, H)
        g["instance_linears.1.bias"] = dr[4 + C:4 + C + K].sum(1)
        dy = Y(3, H)
        g["instance_linears.0.weight"] = unperm_rows(unperm_cols(_mm(dy, X_h), fW, W), fH)
        g["instance_linears.0.bias"] = unperm_rows(_sum0(dy), fH)
    dy = Y(0, H)                                    # views: input [feature, gamma(d)]
    g["views_linears.0.weight"] = unperm_rows(torch.cat([unperm_cols(_mm(dy, F_), fW, W),
                                                         unperm_cols(_mm(dy, ED), ed_idx, EDn)], 1), fH)
    g["views_linears.0.bias"] = unperm_rows(_sum0(dy), fH)
    dy = Y(1, W)
    g["feature_linear.weight"] = unperm_rows(unperm_cols(_mm(dy, X_h), fW, W), fW)
    g["feature_linear.bias"] = unperm_rows(_sum0(dy), fW)
    for l in range(D):
        dy = Y(4 + l, W)
        if l == 0:
            gw = unperm_cols(_mm(dy, EX), ex_idx, EXn)
        elif l - 1 == nerf.skip:
            gw = torch.cat([unperm_cols(_mm(dy, EX), ex_idx, EXn), unperm_cols(_mm(dy, A(1 + l, W)), fW, W)], 1)
        else:
            gw = unperm_cols(_mm(dy, A(1 + l, W)), fW, W)
        g[f"pts_linears.{l}.weight"] = unperm_rows(gw, fW)
        g[f"pts_linears.{l}.bias"] = unperm_rows(_sum0(dy), fW)
    return g


class LevelFn(torch.autograd.Function):
    """One level (coarse or fine) of render_rays: NeRF MLP on every sample + raw2outputs.
    Differentiable w.r.t. the NeRF's parameters only (z comes from the detached sampler).  Besides the maps it
    returns, when the bbox-prior labels are present, the fixed-field maps (differentiable through the weights) and the
    per-sample 3D cross-entropies of the learned fields against those labels."""

    OUT = ("rgb", "depth", "acc", "weights", "semantic", "instance", "fix_semantic", "fix_instance",
           "ce3d_semantic", "ce3d_instance")

    @staticmethod
    def forward(ctx, rend, lv, rays, z, ls, li, noise, names, *params):
        net = rend.net
        nerf = net.nerf(lv)
        dev = rays.device
        desc, img = net.packed(lv, dev, "bf16")
        C, K = nerf.n_sem, nerf.n_inst
        raw, acts = ops.mlp_forward_train(desc, img, rays, z)
        out = ops.composite(raw, z, rays, C, K, True, noise, ls, li, rend.sem_mode, rend.white_bkgd, True)
        empty = torch.zeros(0, device=dev)
        ce_s = ops.ce3d(raw, 4, C, ls) if (C and ls is not None) else None
        ce_i = ops.ce3d(raw, 4 + C, K, li) if (K and li is not None) else None
        ctx.rend, ctx.lv, ctx.names = rend, lv, names
        ctx.set_materialize_grads(False)            # unused outputs arrive as None, not as zero tensors
        ctx.save_for_backward(raw, acts, z, rays, noise if noise is not None else empty,
                              ls if ls is not None else empty, li if li is not None else empty,
                              ce_s if ce_s is not None else empty, ce_i if ce_i is not None else empty)
        ctx.has_noise, ctx.has_ls, ctx.has_li = noise is not None, ls is not None, li is not None
        out["ce3d_semantic"] = ce_s[0].clone() if ce_s is not None else None
        out["ce3d_instance"] = ce_i[0].clone() if ce_i is not None else None
        res = [out.get(k) for k in LevelFn.OUT]
        return tuple(r if r is not None else empty for r in res)

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_acc, g_w, g_sem, g_inst, g_fs, g_fi, g_ces, g_cei):
        raw, acts, z, rays, noise, ls, li, ce_s, ce_i = ctx.saved_tensors
        rend, lv = ctx.rend, ctx.lv
        net = rend.net
        nerf = net.nerf(lv)
        C, K = nerf.n_sem, nerf.n_inst
        R, N = z.shape
        grads = {"rgb": g_rgb, "depth": g_depth, "acc": g_acc, "weights": g_w,
                 "semantic": g_sem if C else None, "instance": g_inst if K else None,
                 "fix_semantic": g_fs if (C and ctx.has_ls) else None, "fix_instance": g_fi if (K and ctx.has_li) else None}
        if rend.white_bkgd and g_rgb is not None:       # rgb += 1 - acc
            grads["acc"] = (g_acc if g_acc is not None else 0) - g_rgb.sum(-1)
        grads = {k: v for k, v in grads.items() if v is not None and v.numel()}
        # d(mean CE)/d logits = (softmax - onehot) / count, times the upstream gradient of the scalar
        sc_s = (g_ces / ce_s[1].clamp(min=1.0)) if (g_ces is not None and ce_s.numel()) else None
        sc_i = (g_cei / ce_i[1].clamp(min=1.0)) if (g_cei is not None and ce_i.numel()) else None
        d_raw = ops.composite_backward(raw, z, rays, C, K, grads, noise if ctx.has_noise else None,
                                       ls if ctx.has_ls else None, li if ctx.has_li else None, sc_s, sc_i, rend.sem_mode)
        desc, img_b = net.packed_bwd(lv, rays.device)
        dys = ops.mlp_backward(desc, img_b, d_raw, acts, R, N)
        shapes = {n: p.shape for n, p in nerf.named_parameters()}
        wg = ops.mlp_wgrad(desc, acts, dys, R * N, shapes)          # pnr_mlp_wgrad (weight_grads() above = torch cross-check)
        return (None,) * 8 + tuple(wg[n].to(p_dtype) for n, p_dtype in ctx.names)


def level_train(rend, lv, rays, z, ls, li, noise):
    """Differentiable level: returns the same dict as ops.composite()."""
    nerf = rend.net.nerf(lv)
    named = list(nerf.named_parameters())
    names = tuple((n, p.dtype) for n, p in named)
    res = LevelFn.apply(rend, lv, rays, z, ls, li, noise, names, *[p for _, p in named])
    out = {k: v for k, v in zip(LevelFn.OUT, res) if v.numel()}
    return out


def allreduce_grads(module, world=None, group=None):
    """Average the gradients of `module` across ranks with ONE flat bucket (~1.3 M fp32
    = 5 MB per step is latency-bound on xGMI, so one all-reduce beats DDP's many buckets).
    RCCL when the process group is 'nccl' (ROCm), gloo in the CPU tests."""
    world = world or dist.get_world_size(group)
    if world == 1:
        return
    ps = [p for p in module.parameters() if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    o = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
