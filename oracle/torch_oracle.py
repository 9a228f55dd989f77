"""Vectorised PyTorch (CPU, fp32) restatement of the PanopticNeRF render_rays path.

TEST INFRASTRUCTURE ONLY.  Nothing under panopticnerf_amd/ may import this module; it is
used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the checker
and as the timed "reference PyTorch CPU path" BASELINE.json asks to report.

PARITY UNPINNED.  /root/reference holds only README.md (README.md:7, README.md:13 point at
the un-mounted code branches `panopticnerf360` / `panopticnerf`), so nothing here can cite
a reference file:line.  Every function follows SURVEY.md section 8a (rows a3..a8): the
canonical NeRF formulation denoted by the names in BASELINE.json's north_star
(render_rays, sample_pdf, raw2outputs, Embedder, coarse/fine NeRF MLP with semantic and
instance heads).  Constants marked (!) are the parity-critical ones SURVEY.md section 9
lists for re-verification once the code branch is mounted.

Weights are passed as a plain dict name -> tensor with the canonical NeRF names:
  pts_linears.{i}.weight/bias (i < D), alpha_linear, feature_linear, views_linears.0,
  rgb_linear, semantic_linears.{0,1}, instance_linears.{0,1}
"""
import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F


def mlp_config(D=8, W=256, skips=(4,), xyz_L=10, dir_L=4, n_sem=0, n_inst=0, head_W=128, head_tap="trunk", head_depth=2):
    """head_tap / head_depth: SURVEY.md 9 item 4 as switches -- the heads read the trunk output h ('trunk') or the
    feature_linear output ('feature'); they are W -> head_W -> n (2) or one Linear W -> n (1)."""
    return SimpleNamespace(D=D, W=W, skips=tuple(skips), xyz_L=xyz_L, dir_L=dir_L,
                           n_sem=n_sem, n_inst=n_inst, head_W=head_W, head_tap=head_tap, head_depth=int(head_depth))


def init_params(cfg, seed=0, sigma_bias=None):
    """nn.Linear default init (kaiming_uniform a=sqrt(5)) for every layer, seeded."""
    g = torch.Generator().manual_seed(seed)
    ex, ed = 3 + 6 * cfg.xyz_L, 3 + 6 * cfg.dir_L
    p = {}

    def lin(name, fin, fout):
        bound = 1.0 / math.sqrt(fin)
        p[name + ".weight"] = (torch.rand(fout, fin, generator=g) * 2 - 1) * bound
        p[name + ".bias"] = (torch.rand(fout, generator=g) * 2 - 1) * bound

    for i in range(cfg.D):
        fin = ex if i == 0 else (cfg.W + ex if (i - 1) in cfg.skips else cfg.W)
        lin(f"pts_linears.{i}", fin, cfg.W)
    lin("alpha_linear", cfg.W, 1)
    lin("feature_linear", cfg.W, cfg.W)
    lin("views_linears.0", cfg.W + ed, cfg.W // 2)
    lin("rgb_linear", cfg.W // 2, 3)
    deep = getattr(cfg, "head_depth", 2) == 2
    for name, n in (("semantic_linears", cfg.n_sem), ("instance_linears", cfg.n_inst)):
        if not n:
            continue
        if deep:
            lin(name + ".0", cfg.W, cfg.head_W)
            lin(name + ".1", cfg.head_W, n)
        else:
            lin(name + ".0", cfg.W, n)
    if sigma_bias is not None:
        p["alpha_linear.bias"] = torch.full((1,), float(sigma_bias))
    return p


def bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


# ------------------------------------------------------------------ a3 stratified sampler
def stratified(rays, n_samples, lindisp=False, t_rand=None):
    near, far = rays[..., 6:7], rays[..., 7:8]
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    if not lindisp:
        z = near * (1.0 - t) + far * t
    else:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


def points(rays, z):
    return rays[..., None, 0:3] + rays[..., None, 3:6] * z[..., :, None]


# ------------------------------------------------------------------ a4 Embedder
def embed(x, L):
    out = [x]
    for k in range(L):
        f = 2.0 ** k
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, -1)


# ------------------------------------------------------------------ a5 NeRF MLP + heads
class _LinearBf16(torch.autograd.Function):
    """One Linear of the bf16 MFMA path WITH its backward (emulate_bf16 = "bwd"): the arithmetic of k_mlp_fused<TRAIN>,
    k_mlp_bwd and k_wgrad up to accumulation order.  Forward: bf16(x) bf16(W)^T + b, fp32 accumulate.  Backward: the incoming
    gradient dY (already summed over its consumers in fp32 and ReLU-gated by autograd, as the d h chain of k_mlp_bwd does) is
    rounded ONCE to bf16 -- that is the dY the kernel stores and feeds on --, then dX = bf16(dY) bf16(W), dW = bf16(dY)^T
    bf16(x), db = sum bf16(dY), each with fp32 accumulation.  hi_lo=True keeps dY as a bf16 hi + lo pair (16 mantissa bits instead of
    8): the student that shows what a higher-precision dY in the kernels would buy (tests/_students.py: nothing measurable)."""

    @staticmethod
    def forward(ctx, x, w, b, hi_lo):
        xq, wq = bf16_round(x), bf16_round(w)
        ctx.save_for_backward(xq, wq)
        ctx.hi_lo = hi_lo
        return F.linear(xq, wq, b)

    @staticmethod
    def backward(ctx, g):
        xq, wq = ctx.saved_tensors
        gq = bf16_round(g)
        if ctx.hi_lo:
            gq = gq + bf16_round(g - gq)
        g2, x2 = gq.reshape(-1, gq.shape[-1]), xq.reshape(-1, xq.shape[-1])
        return gq @ wq, g2.t() @ x2, g2.sum(0), None


def mlp_forward(p, cfg, pts, viewdirs, emulate_bf16=False):
    """pts (S,3), viewdirs (S,3) already normalised (!) -> raw (S, 4+n_sem+n_inst) =
    [rgb(3) sigma(1) semantic logits instance logits].
    emulate_bf16 rounds every Linear's input activations and weights to bf16 (RNE) and
    keeps fp32 accumulation + fp32 bias: the arithmetic of the MFMA bf16 kernel up to
    accumulation order.  emulate_bf16 = "bwd" / "bwd_hilo" also emulates the bf16 BACKWARD of the HIP training path
    (_LinearBf16: every dY rounded to bf16 / to a bf16 hi + lo pair before it is used) -- the student that separates
    "precision of a bf16 backward" from "defect" in tests/test_gpu_convergence.py."""
    q = bf16_round if emulate_bf16 else (lambda t: t)

    if emulate_bf16 in ("bwd", "bwd_hilo"):
        def lin(name, x):
            return _LinearBf16.apply(x, p[name + ".weight"], p[name + ".bias"], emulate_bf16 == "bwd_hilo")
    else:
        def lin(name, x):
            return F.linear(q(x), q(p[name + ".weight"]), p[name + ".bias"])

    ex = embed(pts, cfg.xyz_L)
    ed = embed(viewdirs, cfg.dir_L)
    h = ex
    for i in range(cfg.D):
        h = F.relu(lin(f"pts_linears.{i}", h))
        if i in cfg.skips:
            h = torch.cat([ex, h], -1)           # (!) order: [gamma(x), h]
    sigma = lin("alpha_linear", h)
    feat = lin("feature_linear", h)
    g = F.relu(lin("views_linears.0", torch.cat([feat, ed], -1)))   # (!) order: [feature, gamma(d)]
    rgb = lin("rgb_linear", g)
    outs = [rgb, sigma]
    tap = feat if getattr(cfg, "head_tap", "trunk") == "feature" else h
    deep = getattr(cfg, "head_depth", 2) == 2
    for name, n in (("semantic_linears", cfg.n_sem), ("instance_linears", cfg.n_inst)):
        if n:
            outs.append(lin(name + ".1", F.relu(lin(name + ".0", tap))) if deep else lin(name + ".0", tap))
    return torch.cat(outs, -1)


def run_network(p, cfg, rays, z, emulate_bf16=False, chunk=1 << 16):
    """rays (R,8), z (R,N) -> raw (R,N,ch)."""
    R, N = z.shape
    pts = points(rays, z).reshape(-1, 3)
    d = rays[:, 3:6]
    vd = (d / torch.norm(d, dim=-1, keepdim=True))[:, None, :].expand(R, N, 3).reshape(-1, 3)
    outs = []
    for s in range(0, pts.shape[0], chunk):
        outs.append(mlp_forward(p, cfg, pts[s:s + chunk], vd[s:s + chunk], emulate_bf16))
    return torch.cat(outs, 0).reshape(R, N, -1)


# ------------------------------------------------------------------ a6 raw2outputs
def raw2outputs(raw, z, rays_d, n_sem=0, n_inst=0, noise=None, label_sem=None, label_inst=None,
                sem_mode=0, white_bkgd=False):
    dists = z[..., 1:] - z[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1)    # (!) 1e10
    dists = dists * torch.norm(rays_d[..., None, :], dim=-1)                # (!) * ||d||
    rgb = torch.sigmoid(raw[..., :3])
    sigma = raw[..., 3]
    if noise is not None:
        sigma = sigma + noise
    alpha = 1.0 - torch.exp(-F.relu(sigma) * dists)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[..., :1]), 1.0 - alpha + 1e-10], -1), -1)[..., :-1]
    w = alpha * T
    out = {"weights": w,
           "rgb": torch.sum(w[..., None] * rgb, -2),
           "depth": torch.sum(w * z, -1),
           "acc": torch.sum(w, -1)}
    if white_bkgd:
        out["rgb"] = out["rgb"] + (1.0 - out["acc"][..., None])
    if n_sem:
        s = raw[..., 4:4 + n_sem]
        if sem_mode == 1:
            s = torch.softmax(s, -1)
        out["semantic"] = torch.sum(w[..., None] * s, -2)
        if label_sem is not None:
            oh = F.one_hot(label_sem.clamp(min=0).long(), n_sem).float() * (label_sem >= 0)[..., None]
            out["fix_semantic"] = torch.sum(w[..., None] * oh, -2)
    if n_inst:
        s = raw[..., 4 + n_sem:4 + n_sem + n_inst]
        if sem_mode == 1:
            s = torch.softmax(s, -1)
        out["instance"] = torch.sum(w[..., None] * s, -2)
        if label_inst is not None:
            oh = F.one_hot(label_inst.clamp(min=0).long(), n_inst).float() * (label_inst >= 0)[..., None]
            out["fix_instance"] = torch.sum(w[..., None] * oh, -2)
    return out


# ------------------------------------------------------------------ a7 sample_pdf
def sample_pdf(bins, weights, n_importance, det=True, u=None):
    weights = weights + 1e-5                                                 # (!) 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        assert det
        u = torch.linspace(0.0, 1.0, steps=n_importance).expand(list(cdf.shape[:-1]) + [n_importance])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bins_b, bins_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)         # (!) 1e-5
    t = (u - cdf_b) / denom
    return bins_b + t * (bins_a - bins_b), inds


def importance_z(z, weights, n_importance, u=None):
    """coarse z (R,Nc), coarse weights (R,Nc) -> sorted union z_fine (R,Nc+Nf), z_samples, inds."""
    mid = 0.5 * (z[..., 1:] + z[..., :-1])
    zs, inds = sample_pdf(mid, weights[..., 1:-1], n_importance, det=(u is None), u=u)
    zs = zs.detach()
    z_fine, _ = torch.sort(torch.cat([z, zs], -1), -1)
    return z_fine, zs, inds


# ------------------------------------------------------------------ a8 bbox prior
def bbox_hits(rays, box, max_hits):
    """rays (R,8), box (M,15) = c(3) Rrows(9) e(3).  Dense slab test -> the max_hits NEAREST hits per ray in
    ascending (t_in, box index) order.  Returns hit_t (R,mh,2), hit_box (R,mh) int32, count (R) int32 = the true
    number of intersected boxes (may exceed max_hits)."""
    o, d, near, far = rays[:, None, 0:3], rays[:, None, 3:6], rays[:, None, 6], rays[:, None, 7]
    c, Rm, e = box[None, :, 0:3], box[None, :, 3:12].reshape(1, -1, 3, 3), box[None, :, 12:15]
    p = o - c
    ol = (Rm[..., 0] * p[..., None, 0] + Rm[..., 1] * p[..., None, 1]) + Rm[..., 2] * p[..., None, 2]
    dl = (Rm[..., 0] * d[..., None, 0] + Rm[..., 1] * d[..., None, 1]) + Rm[..., 2] * d[..., None, 2]
    inv = 1.0 / dl
    t1, t2 = (-e - ol) * inv, (e - ol) * inv
    tn, tf = torch.fmin(t1, t2), torch.fmax(t1, t2)
    tmin, tmax = near.expand(-1, box.shape[0]).clone(), far.expand(-1, box.shape[0]).clone()
    for a in range(3):
        tmin = torch.fmax(tmin, tn[..., a])
        tmax = torch.fmin(tmax, tf[..., a])
    hit = tmin <= tmax
    R, M = hit.shape
    hit_t = torch.zeros(R, max_hits, 2)
    hit_box = torch.full((R, max_hits), -1, dtype=torch.int32)
    if M == 0:
        return hit_t, hit_box, torch.zeros(R, dtype=torch.int32)
    key = torch.where(hit, tmin, torch.full_like(tmin, float("inf")))
    order = torch.argsort(key, dim=1, stable=True)[:, :max_hits]          # stable: ties keep ascending box index
    kept = torch.gather(hit, 1, order)
    k = min(max_hits, M)
    hit_t[:, :k, 0] = torch.where(kept, torch.gather(tmin, 1, order), torch.zeros(()))
    hit_t[:, :k, 1] = torch.where(kept, torch.gather(tmax, 1, order), torch.zeros(()))
    hit_box[:, :k] = torch.where(kept, order.int(), torch.full((), -1, dtype=torch.int32))
    return hit_t, hit_box, hit.sum(1).int()


def restrict_rays(rays, hit_t, hit_count):
    """cfg.bbox_sampling = "hull" (a switch): near / far of a ray that hits boxes -> the hull of its
    kept intervals [min t_in, max t_out]; rays without a hit are unchanged."""
    mh = hit_t.shape[1]
    valid = torch.arange(mh)[None, :] < hit_count.clamp(max=mh)[:, None]
    lo = torch.where(valid, hit_t[..., 0], torch.full_like(hit_t[..., 0], float("inf"))).min(-1).values
    hi = torch.where(valid, hit_t[..., 1], torch.full_like(hit_t[..., 1], float("-inf"))).max(-1).values
    has = hit_count > 0
    out = rays.clone()
    out[:, 6] = torch.where(has, lo, rays[:, 6])
    out[:, 7] = torch.where(has, hi, rays[:, 7])
    return out


def sample_labels(z, hit_t, hit_box, hit_count, box_ids):
    R, N = z.shape
    mh = hit_box.shape[1]
    valid = (torch.arange(mh)[None, :] < hit_count.clamp(max=mh)[:, None])[:, None, :]
    ti, to = hit_t[:, None, :, 0], hit_t[:, None, :, 1]
    inside = valid & (ti <= z[..., None]) & (z[..., None] <= to)
    key = torch.where(inside, ti.expand(R, N, mh), torch.full((R, N, mh), float("inf")))
    best = torch.argmin(key, -1)          # first minimum on ties
    any_in = inside.any(-1)
    bidx = torch.gather(hit_box.long(), 1, best).clamp(min=0)
    ids = box_ids.long()
    ls = torch.where(any_in, ids[bidx, 0], torch.full_like(bidx, -1))
    li = torch.where(any_in, ids[bidx, 1], torch.full_like(bidx, -1))
    return ls.int(), li.int()


# ------------------------------------------------------------------ a2 render_rays
def render_rays(params, cfg, rays, n_samples, n_importance=0, lindisp=False, t_rand=None, u=None,
                noise0=None, noise1=None, box=None, box_ids=None, max_hits=8, sem_mode=0,
                white_bkgd=False, emulate_bf16=False, keep_raw=False, bbox_sampling="none"):
    """params: {"coarse": dict, "fine": dict} (fine used when n_importance>0).
    Returns dict with *_0 (coarse) and *_1 (fine) maps."""
    ret = {}
    hits = None
    if box is not None:
        hits = bbox_hits(rays, box, max_hits)
    rays_s = restrict_rays(rays, hits[0], hits[2]) if (hits is not None and bbox_sampling == "hull") else rays
    z = stratified(rays_s, n_samples, lindisp, t_rand)

    def level(tag, prm, zz, noise):
        raw = run_network(prm, cfg, rays, zz, emulate_bf16)
        ls = li = None
        if hits is not None:
            ls, li = sample_labels(zz, hits[0], hits[1], hits[2], box_ids)
        o = raw2outputs(raw, zz, rays[:, 3:6], cfg.n_sem, cfg.n_inst, noise, ls, li, sem_mode, white_bkgd)
        for k, v in o.items():
            ret[f"{k}_{tag}"] = v
        ret[f"z_vals_{tag}"] = zz
        if keep_raw:
            ret[f"raw_{tag}"] = raw
        return o["weights"]

    w0 = level(0, params["coarse"], z, noise0)
    if n_importance > 0:
        z_fine, zs, inds = importance_z(z, w0, n_importance, u)
        ret["z_samples"], ret["inds"] = zs, inds
        level(1, params["fine"], z_fine, noise1)
    return ret


# ---- loss wrapper.  PARITY UNPINNED like everything above: the reference's NetworkWrapper is
# not in the mount; the terms follow SURVEY.md section 2 row 8 (RGB MSE, stereo-depth L1/L2, 2D pseudo-label CE on the
# learned and on the fixed field, 3D bbox CE).
def losses(maps, targets, weights, n_sem=0, n_inst=0, depth_l2=False, fix_eps=1e-5, maps_are_prob=False):
    """maps: rgb (R,3), depth (R), semantic/fix_semantic (R,C), instance/fix_instance (R,K) (any subset);
    targets: rgb, depth, semantic (R) int, instance (R) int.  Returns (dict of the six means, weighted total)."""
    import torch.nn.functional as F
    out = {}
    z = lambda: torch.zeros((), dtype=torch.float32)
    if "rgb" in maps and targets.get("rgb") is not None:
        out["rgb"] = ((maps["rgb"] - targets["rgb"]) ** 2).mean()
    if "depth" in maps and targets.get("depth") is not None:
        v = targets["depth"] > 0
        d = (maps["depth"] - targets["depth"])[v]
        out["depth"] = ((d ** 2) if depth_l2 else d.abs()).sum() / max(int(v.sum()), 1)
    for key, fkey, tkey, n in (("semantic", "fix_semantic", "semantic", n_sem), ("instance", "fix_instance", "instance", n_inst)):
        t = targets.get(tkey)
        if t is None or n == 0:
            continue
        v = (t >= 0) & (t < n)
        cnt = max(int(v.sum()), 1)
        tl = t[v].long()
        if key in maps and maps_are_prob:
            out[key] = (-(maps[key][v].gather(1, tl[:, None])[:, 0] + fix_eps).log()).sum() / cnt if v.any() else z()
        elif key in maps:
            out[key] = F.cross_entropy(maps[key][v], tl, reduction="sum") / cnt if v.any() else z()
        if fkey in maps:
            out[fkey] = (-(maps[fkey][v].gather(1, tl[:, None])[:, 0] + fix_eps).log()).sum() / cnt if v.any() else z()
    total = sum(float(weights.get(k, 0.0)) * v for k, v in out.items())
    return out, total


def ce3d(logits, label):
    """logits (S,n_cls), label (S) int, -1 = unlabelled -> (mean CE over labelled samples, count)."""
    import torch.nn.functional as F
    v = (label >= 0) & (label < logits.shape[1])
    cnt = int(v.sum())
    if cnt == 0:
        return torch.zeros(()), 0
    return F.cross_entropy(logits[v], label[v].long(), reduction="sum") / cnt, cnt


def gen_rays(intr, c2w, width, height, near, far, pix=None):
    """Pinhole ray generation, vectorised: same op order as pnro_gen_rays."""
    intr = torch.as_tensor(intr, dtype=torch.float32).reshape(4)
    c2w = torch.as_tensor(c2w, dtype=torch.float32).reshape(3, 4)
    p = torch.arange(width * height) if pix is None else torch.as_tensor(pix).long()
    j = torch.div(p, width, rounding_mode="floor")
    i = p - j * width
    x = (i.float() - intr[2]) / intr[0]
    y = (j.float() - intr[3]) / intr[1]
    d = torch.stack([(c2w[k, 0] * x + c2w[k, 1] * y) + c2w[k, 2] for k in range(3)], -1)
    o = c2w[:, 3].expand_as(d)
    nf = torch.tensor([near, far], dtype=torch.float32).expand(d.shape[0], 2)
    return torch.cat([o, d, nf], -1).contiguous()
