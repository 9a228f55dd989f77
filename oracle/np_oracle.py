"""Plain-loop numpy float64 restatement of the per-ray stages, written from the equations
of SURVEY.md section 8a (not from torch_oracle.py / pnr_oracle.c) so that the three
restatements check one another (SURVEY.md 8c "how to keep a self-written oracle honest").

TEST INFRASTRUCTURE ONLY; small cases only (pure Python loops).  PARITY UNPINNED: the
reference mount holds no source, see oracle/pnr_oracle.c's header.
"""
import numpy as np


def embed(x, L):
    x = np.asarray(x, np.float64)
    out = np.zeros((x.shape[0], 3 + 6 * L))
    for s in range(x.shape[0]):
        out[s, 0:3] = x[s]
        for k in range(L):
            for a in range(3):
                out[s, 3 + 6 * k + a] = np.sin(x[s, a] * 2.0 ** k)
                out[s, 3 + 6 * k + 3 + a] = np.cos(x[s, a] * 2.0 ** k)
    return out


def stratified(rays, N, lindisp=False, t_rand=None):
    rays = np.asarray(rays, np.float64)
    R = rays.shape[0]
    z = np.zeros((R, N))
    for r in range(R):
        near, far = rays[r, 6], rays[r, 7]
        for i in range(N):
            t = i / (N - 1) if N > 1 else 0.0
            z[r, i] = 1.0 / ((1 - t) / near + t / far) if lindisp else near * (1 - t) + far * t
        if t_rand is not None:
            zz = z[r].copy()
            for i in range(N):
                lo = zz[0] if i == 0 else 0.5 * (zz[i] + zz[i - 1])
                up = zz[N - 1] if i == N - 1 else 0.5 * (zz[i] + zz[i + 1])
                z[r, i] = lo + (up - lo) * float(t_rand[r, i])
    return z


def composite(raw, z, rays, C, K, noise=None, label_sem=None, label_inst=None, sem_mode=0,
              white_bkgd=False):
    """raw (R,N,4+C+K) sample-major."""
    raw = np.asarray(raw, np.float64)
    z = np.asarray(z, np.float64)
    rays = np.asarray(rays, np.float64)
    R, N = z.shape
    out = dict(rgb=np.zeros((R, 3)), depth=np.zeros(R), acc=np.zeros(R), weights=np.zeros((R, N)),
               semantic=np.zeros((R, C)), instance=np.zeros((R, K)),
               fix_semantic=np.zeros((R, C)), fix_instance=np.zeros((R, K)))

    def field(v):
        if sem_mode == 0:
            return v
        e = np.exp(v - v.max())
        return e / e.sum()

    for r in range(R):
        dn = np.sqrt((rays[r, 3:6] ** 2).sum())
        T = 1.0
        for i in range(N):
            dist = (z[r, i + 1] - z[r, i]) if i + 1 < N else 1e10
            sg = raw[r, i, 3] + (0.0 if noise is None else float(noise[r, i]))
            alpha = 1.0 - np.exp(-max(sg, 0.0) * dist * dn)
            w = alpha * T
            T *= (1.0 - alpha + 1e-10)
            out["weights"][r, i] = w
            out["rgb"][r] += w / (1.0 + np.exp(-raw[r, i, 0:3]))
            out["depth"][r] += w * z[r, i]
            out["acc"][r] += w
            if C:
                out["semantic"][r] += w * field(raw[r, i, 4:4 + C])
                if label_sem is not None and 0 <= label_sem[r, i] < C:
                    out["fix_semantic"][r, label_sem[r, i]] += w
            if K:
                out["instance"][r] += w * field(raw[r, i, 4 + C:4 + C + K])
                if label_inst is not None and 0 <= label_inst[r, i] < K:
                    out["fix_instance"][r, label_inst[r, i]] += w
        if white_bkgd:
            out["rgb"][r] += 1.0 - out["acc"][r]
    return out


def composite_by_tiles(raw, z, rays, C, K, tile=32, label_sem=None, label_inst=None, white_bkgd=False):
    """The factorisation the fused inference pass uses (csrc/pnr_mlp_fuse.h + k_composite_combine), restated with plain loops:
    every `tile` consecutive samples of a ray are reduced on their own to a record
        Q = prod (1 - alpha + 1e-10),   S_c = sum lw_i v_ci   with   lw_i = alpha_i prod_{j < i in tile} (1 - alpha_j + 1e-10),
    and the ray is finished from its records:  out_c = sum_k T_k S_c(k),  T_k = prod_{k' < k} Q_k',  weights_i = T_k lw_i.
    Must equal composite() (logits mode) up to rounding: tests/test_oracle.py checks it, so the GPU test that compares the
    fused kernels with k_composite rests on a CPU statement of WHY the two agree.  raw (R,N,4+C+K) sample-major."""
    raw = np.asarray(raw, np.float64)
    z = np.asarray(z, np.float64)
    rays = np.asarray(rays, np.float64)
    R, N = z.shape
    assert N % tile == 0
    out = dict(rgb=np.zeros((R, 3)), depth=np.zeros(R), acc=np.zeros(R), weights=np.zeros((R, N)),
               semantic=np.zeros((R, C)), instance=np.zeros((R, K)),
               fix_semantic=np.zeros((R, C)), fix_instance=np.zeros((R, K)))
    for r in range(R):
        dn = np.sqrt((rays[r, 3:6] ** 2).sum())
        records = []
        for k in range(N // tile):                       # ---- what one wave of the MLP does for its tile
            rec = dict(Q=1.0, acc=0.0, depth=0.0, rgb=np.zeros(3), sem=np.zeros(C), inst=np.zeros(K), fs=np.zeros(C),
                       fi=np.zeros(K), lw=np.zeros(tile))
            for j in range(tile):
                i = k * tile + j
                dist = (z[r, i + 1] - z[r, i]) if i + 1 < N else 1e10      # z of the next sample: also across the tile edge
                alpha = 1.0 - np.exp(-max(raw[r, i, 3], 0.0) * dist * dn)
                lw = alpha * rec["Q"]
                rec["Q"] *= (1.0 - alpha + 1e-10)
                rec["lw"][j] = lw
                rec["acc"] += lw
                rec["depth"] += lw * z[r, i]
                rec["rgb"] += lw / (1.0 + np.exp(-raw[r, i, 0:3]))
                if C:
                    rec["sem"] += lw * raw[r, i, 4:4 + C]
                    if label_sem is not None and 0 <= label_sem[r, i] < C:
                        rec["fs"][label_sem[r, i]] += lw
                if K:
                    rec["inst"] += lw * raw[r, i, 4 + C:4 + C + K]
                    if label_inst is not None and 0 <= label_inst[r, i] < K:
                        rec["fi"][label_inst[r, i]] += lw
            records.append(rec)
        T = 1.0                                            # ---- what k_composite_combine does for the ray
        for k, rec in enumerate(records):
            out["acc"][r] += T * rec["acc"]
            out["depth"][r] += T * rec["depth"]
            out["rgb"][r] += T * rec["rgb"]
            out["semantic"][r] += T * rec["sem"]
            out["instance"][r] += T * rec["inst"]
            out["fix_semantic"][r] += T * rec["fs"]
            out["fix_instance"][r] += T * rec["fi"]
            out["weights"][r, k * tile:(k + 1) * tile] = T * rec["lw"]
            T *= rec["Q"]
        if white_bkgd:
            out["rgb"][r] += 1.0 - out["acc"][r]
    return out


def sample_pdf(z, weights, Nf, u=None):
    """Coarse z (R,Nc), coarse weights (R,Nc) -> z_samples (R,Nf), inds (R,Nf)."""
    z = np.asarray(z, np.float64)
    weights = np.asarray(weights, np.float64)
    R, Nc = z.shape
    zs = np.zeros((R, Nf))
    inds = np.zeros((R, Nf), np.int64)
    for r in range(R):
        bins = [0.5 * (z[r, k + 1] + z[r, k]) for k in range(Nc - 1)]
        w = [weights[r, j + 1] + 1e-5 for j in range(Nc - 2)]
        tot = sum(w)
        cdf = [0.0]
        for j in range(Nc - 2):
            cdf.append(cdf[-1] + w[j] / tot)
        for i in range(Nf):
            uu = float(u[r, i]) if u is not None else (i / (Nf - 1) if Nf > 1 else 0.0)
            ind = sum(1 for c in cdf if c <= uu)
            below, above = max(ind - 1, 0), min(ind, Nc - 2)
            den = cdf[above] - cdf[below]
            if den < 1e-5:
                den = 1.0
            t = (uu - cdf[below]) / den
            zs[r, i] = bins[below] + t * (bins[above] - bins[below])
            inds[r, i] = ind
    return zs, inds


def bbox_hits(rays, box, max_hits):
    """All hits per ray, then the max_hits nearest by (t_in, box index); count = true number of hits."""
    rays = np.asarray(rays, np.float64)
    box = np.asarray(box, np.float64)
    R, M = rays.shape[0], box.shape[0]
    hit_t = np.zeros((R, max_hits, 2))
    hit_box = -np.ones((R, max_hits), np.int64)
    cnt = np.zeros(R, np.int64)
    for r in range(R):
        o, d, near, far = rays[r, 0:3], rays[r, 3:6], rays[r, 6], rays[r, 7]
        found = []
        for m in range(M):
            c, Rm, e = box[m, 0:3], box[m, 3:12].reshape(3, 3), box[m, 12:15]
            ol, dl = Rm @ (o - c), Rm @ d
            tmin, tmax = near, far
            with np.errstate(divide="ignore", invalid="ignore"):
                for a in range(3):
                    t1, t2 = (-e[a] - ol[a]) / dl[a], (e[a] - ol[a]) / dl[a]
                    tmin = np.fmax(tmin, np.fmin(t1, t2))
                    tmax = np.fmin(tmax, np.fmax(t1, t2))
            if tmin <= tmax:
                found.append((tmin, m, tmax))
        cnt[r] = len(found)
        for k, (tmin, m, tmax) in enumerate(sorted(found)[:max_hits]):
            hit_t[r, k] = (tmin, tmax)
            hit_box[r, k] = m
    return hit_t, hit_box, cnt


# ---- SURVEY.md 8f rank 4 (post-processing / evaluator counters), plain numpy.  Parity unpinned: the reference's
# evaluator is not in the mount; the conventions are documented in include/pnr.h.
def panoptic_labels(sem, inst=None, is_thing=None):
    nan_low = lambda a: np.where(np.isnan(a), -np.inf, a)      # a NaN logit never wins (the kernel reads it as -inf)
    sem = nan_low(np.asarray(sem, np.float32))
    sl = np.argmax(sem, 1).astype(np.int32)                    # first maximum = lowest index on ties
    il = np.full(sem.shape[0], -1, np.int32)
    if inst is not None and inst.shape[1] > 0:
        ia = np.argmax(nan_low(np.asarray(inst, np.float32)), 1).astype(np.int32)
        thing = np.ones(sem.shape[0], bool) if is_thing is None else (np.asarray(is_thing)[sl] != 0)
        il = np.where(thing, ia, -1).astype(np.int32)
    pan = np.where(il >= 0, sl * 1000 + il, sl).astype(np.int32)
    return sl, il, pan


def confusion(pred, gt, n_classes):
    pred, gt = np.asarray(pred, np.int64), np.asarray(gt, np.int64)
    v = (gt >= 0) & (gt < n_classes) & (pred >= 0) & (pred < n_classes)
    return np.bincount(gt[v] * n_classes + pred[v], minlength=n_classes * n_classes).reshape(n_classes, n_classes).astype(np.int64)


def panoptic_quality_terms(pred_id, gt_id, n_classes):
    """Per-class (sum of matched IoUs, TP, FP, FN) of one frame; ids = class*1000 + instance on things, class on stuff,
    gt < 0 = ignore.  Plain loops over segments (Kirillov et al., "Panoptic Segmentation": match iff same class and
    IoU > 0.5)."""
    pred_id, gt_id = np.asarray(pred_id, np.int64), np.asarray(gt_id, np.int64)
    valid = gt_id >= 0
    cls = lambda i: int(i // 1000) if i >= 1000 else int(i)
    out = np.zeros((n_classes, 4))
    gts = [g for g in np.unique(gt_id[valid])]
    preds = [p for p in np.unique(pred_id[valid])]
    matched_p = set()
    for g in gts:
        gm = (gt_id == g) & valid
        hit = False
        for p in preds:
            if cls(p) != cls(g):
                continue
            pm = (pred_id == p) & valid
            inter = np.count_nonzero(gm & pm)
            union = np.count_nonzero(gm) + np.count_nonzero(pm) - inter
            if union > 0 and inter / union > 0.5:
                out[cls(g), 0] += inter / union
                out[cls(g), 1] += 1
                matched_p.add(p)
                hit = True
        if not hit:
            out[cls(g), 3] += 1
    for p in preds:
        if p not in matched_p:
            out[cls(p), 2] += 1
    return out
