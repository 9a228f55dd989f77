"""Test support: the weight gradients of the NeRF computed with LIBRARY GEMMs (torch.bmm over sample slabs, fp32 out)
on the kernels' slot-ordered bf16 buffers, and the slot maps that un-permute them.  This was the training path's
implementation before pnr_mlp_wgrad existed; it stays as the independent cross-check of that kernel
(tests/test_gpu_backward.py) and as the comparison line of tools/train_profile.py.  Not part of the product package."""
import functools

import torch

from panopticnerf_amd import ops


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


def pad_samples(S):
    return (S + 255) // 256 * 256


def saved_rows(buf, off, S, w):
    """(S, w) slot-ordered rows of a region in the saved-tensor layout (pnr_mlp_layout.h: lines [s >> 3][chunk] of 8 samples x
    16 B, sample s at position (s & 7) ^ (4 * ((chunk >> 1) & 1)))."""
    Sp, cpr = pad_samples(S), w // 8
    v = buf[off: off + Sp * w].view(Sp // 8, cpr, 8, 8)               # [group][chunk][position][element]
    c = torch.arange(cpr, device=buf.device)
    pos = torch.arange(8, device=buf.device)[None, :] ^ (((c >> 1) & 1) << 2)[:, None]      # [chunk][s & 7] -> position
    rows = v[:, c[:, None], pos, :]                                     # [group][chunk][s & 7][element]
    return rows.permute(0, 2, 1, 3).reshape(Sp, w)[:S]


def fill_saved_rows(buf, off, S, w, rows):
    """inverse of saved_rows; the padding rows S..S_pad are zeroed (the kernels write zeros / finite values there)."""
    Sp, cpr = pad_samples(S), w // 8
    full = torch.zeros((Sp, w), device=buf.device, dtype=buf.dtype)
    full[:S] = rows
    r = full.view(Sp // 8, 8, cpr, 8).permute(0, 2, 1, 3)               # [group][chunk][s & 7][element]
    c = torch.arange(cpr, device=buf.device)
    pos = torch.arange(8, device=buf.device)[None, :] ^ (((c >> 1) & 1) << 2)[:, None]
    v = buf[off: off + Sp * w].view(Sp // 8, cpr, 8, 8)
    v[:, c[:, None], pos, :] = r
    return buf


def weight_grads(nerf, desc, acts, dys, d_raw, S):
    """dict name -> fp32 gradient (nn.Linear layout) from the kernels' buffers."""
    dev = str(d_raw.device)
    D, W, H, C, K = nerf.D, nerf.W, nerf.W // 2, nerf.n_sem, nerf.n_inst
    ao, do = ops.train_layout(desc, S)
    A = lambda i, w: saved_rows(acts, ao[i], S, w)
    Y = lambda i, w: saved_rows(dys, do[i], S, w)
    fW, fH = feat_slots(W, dev), feat_slots(H, dev)
    ex_idx, ed_idx = embed_slots(5, nerf.xyz_L, dev), embed_slots(2, nerf.dir_L, dev)
    EXn, EDn = 3 + 6 * nerf.xyz_L, 3 + 6 * nerf.dir_L

    def unperm_cols(g_slot, idx, n_cols):           # (m, slots) -> (m, n_cols): column c <- slot with idx == c
        return g_slot.index_select(1, _inverse(idx, n_cols))

    def unperm_rows(g, idx):                        # rows in slot order -> feature order
        return g.index_select(0, _inverse(idx, idx.numel()))

    X_h = A(1 + D, W)                               # h = X_D
    X_tap = A(2 + D, W) if getattr(desc, "head_tap", 0) else X_h      # what the heads read: the feature (head_tap 1) or h
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
        g["semantic_linears.0.weight"] = unperm_rows(unperm_cols(_mm(dy, X_tap), fW, W), fH)
        g["semantic_linears.0.bias"] = unperm_rows(_sum0(dy), fH)
    if K:
        g["instance_linears.1.weight"] = unperm_cols(_mm(drb[:, 4 + C:4 + C + K].contiguous(), SHI), fH, H)
        g["instance_linears.1.bias"] = dr[4 + C:4 + C + K].sum(1)
        dy = Y(3, H)
        g["instance_linears.0.weight"] = unperm_rows(unperm_cols(_mm(dy, X_tap), fW, W), fH)
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
