"""Hypothesis property tests of the path's invariants on the strict-order C oracle -- the checker
the GPU parity tests compare against, so these pin the checker itself: sum(w) <= 1, w >= 0, z_fine sorted and inside
[near, far], sample_pdf indices inside [0, Nc-2], compositing linear in the logits, fixed-field maps bounded by acc,
nearest-hit selection independent of max_hits.  The same properties are asserted of the HIP path at BASELINE's
full-frame size in tests/test_gpu_render.py / test_gpu_configs.py."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import c_oracle as co

SET = dict(max_examples=40, deadline=None)


def _rays(rng, R, near, far):
    o = rng.normal(0, 1.0, (R, 3))
    d = rng.normal(0, 0.4, (R, 3)) + np.array([0, 0, 1.0])
    return np.concatenate([o, d, np.full((R, 1), near), np.full((R, 1), far)], 1).astype(np.float32)


@settings(**SET)
@given(seed=st.integers(0, 2**31 - 1), N=st.sampled_from([4, 8, 32, 64, 192, 256]), C=st.integers(0, 7), K=st.integers(0, 5),
       sig=st.floats(0.01, 5.0), near=st.floats(0.05, 2.0), span=st.floats(0.5, 200.0), sem_mode=st.integers(0, 1))
def test_compositing_invariants(seed, N, C, K, sig, near, span, sem_mode):
    rng = np.random.default_rng(seed)
    R = 5
    rays = _rays(rng, R, near, near + span)
    z = co.stratified(rays, N, t_rand=rng.random((R, N)).astype(np.float32))
    raw = rng.normal(0, 1.0, (R, N, 4 + C + K)).astype(np.float32)
    raw[..., 3] = rng.normal(0, sig, (R, N))
    ls = rng.integers(-1, max(C, 1), (R, N)).astype(np.int32) if C else None
    li = rng.integers(-1, max(K, 1), (R, N)).astype(np.int32) if K else None
    o = co.composite(raw, z, rays, C, K, label_sem=ls, label_inst=li, sem_mode=sem_mode)
    w = o["weights"]
    assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-5).all()
    np.testing.assert_allclose(o["acc"], w.sum(-1), atol=1e-5)
    assert (o["rgb"] >= -1e-6).all() and (o["rgb"] <= 1 + 1e-5).all()                 # convex combination of sigmoids
    assert (o["depth"] <= z.max(-1) * (1 + 1e-5) + 1e-5).all() and (o["depth"] >= -1e-6).all()
    if C:
        assert (o["fix_semantic"] >= 0).all() and (o["fix_semantic"].sum(-1) <= o["acc"] + 1e-5).all()
        if sem_mode == 1:                                                             # softmax compositing: a sub-probability
            assert (o["semantic"] >= 0).all() and np.allclose(o["semantic"].sum(-1), o["acc"], atol=1e-4)
    if K:
        assert (o["fix_instance"].sum(-1) <= o["acc"] + 1e-5).all()
    # linear in the logits (logit compositing): maps(a*s1 + b*s2) = a*maps(s1) + b*maps(s2) for fixed densities
    if C and sem_mode == 0:
        r2 = raw.copy()
        r2[..., 4:4 + C] = rng.normal(0, 1.0, (R, N, C))
        r3 = raw.copy()
        r3[..., 4:4 + C] = 2.0 * raw[..., 4:4 + C] - 0.5 * r2[..., 4:4 + C]
        s1, s2, s3 = (co.composite(r, z, rays, C, K)["semantic"] for r in (raw, r2, r3))
        np.testing.assert_allclose(s3, 2.0 * s1 - 0.5 * s2, atol=2e-4 * (1 + np.abs(s1).max() + np.abs(s2).max()))


@settings(**SET)
@given(seed=st.integers(0, 2**31 - 1), Nc=st.sampled_from([4, 16, 64, 128]), Nf=st.sampled_from([1, 7, 64, 128]),
       det=st.booleans(), peaky=st.booleans(), near=st.floats(0.05, 2.0), span=st.floats(0.5, 200.0))
def test_sample_pdf_invariants(seed, Nc, Nf, det, peaky, near, span):
    rng = np.random.default_rng(seed)
    R = 6
    rays = _rays(rng, R, near, near + span)
    z = co.stratified(rays, Nc, t_rand=rng.random((R, Nc)).astype(np.float32))
    w = rng.random((R, Nc)).astype(np.float32)
    if peaky:                                      # one dominant bin, the rest (almost) empty: the denom < 1e-5 branch
        w *= 1e-7
        w[np.arange(R), rng.integers(0, Nc, R)] = 1.0
    w[0] = 0.0                                     # a ray that hit nothing: uniform pdf from the +1e-5
    u = None if det else rng.random((R, Nf)).astype(np.float32)
    zs, inds = co.sample_pdf(z, w, Nf, u)
    assert np.isfinite(zs).all()
    assert (inds >= 0).all() and (inds <= Nc - 1).all()          # searchsorted(right) over the Nc-1 entry CDF
    mids_lo, mids_hi = 0.5 * (z[:, 0] + z[:, 1]), 0.5 * (z[:, -2] + z[:, -1])
    assert (zs >= mids_lo[:, None] - 1e-4).all() and (zs <= mids_hi[:, None] + 1e-4).all()
    zf = co.merge_sorted(z, zs)
    assert zf.shape == (R, Nc + Nf) and (np.diff(zf, axis=1) >= 0).all()
    assert (zf.min(1) >= near - 1e-5).all() and (zf.max(1) <= near + span + 1e-3).all()
    # the merge is a permutation of the union
    assert np.array_equal(np.sort(np.concatenate([z, zs], 1), 1), zf)
    if det:                                        # deterministic u ascending => samples ascending
        assert (np.diff(zs, axis=1) >= -1e-5).all()


@settings(**SET)
@given(seed=st.integers(0, 2**31 - 1), M=st.integers(0, 24), mh=st.integers(1, 6), scale=st.floats(0.5, 8.0))
def test_bbox_invariants(seed, M, mh, scale):
    rng = np.random.default_rng(seed)
    R, N = 7, 16
    rays = _rays(rng, R, 0.2, 60.0)
    box = np.zeros((M, 15), np.float32)
    box[:, 0:3] = rng.uniform([-6, -3, 1], [6, 3, 50], (M, 3))
    for m in range(M):
        y = rng.uniform(0, np.pi)
        box[m, 3:12] = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]]).reshape(-1)
    box[:, 12:15] = rng.uniform(0.3, 1.0, (M, 3)) * scale
    ids = np.stack([rng.integers(0, 5, M), rng.integers(0, 4, M)], 1).astype(np.int32).reshape(M, 2)
    t, b, cnt = co.bbox_hits(rays, box, mh)
    t_all, b_all, cnt_all = co.bbox_hits(rays, box, max(M, 1))
    assert np.array_equal(cnt, cnt_all)                                           # the true count never depends on max_hits
    k = min(mh, max(M, 1))
    assert np.array_equal(t[:, :k], t_all[:, :k]) and np.array_equal(b[:, :k], b_all[:, :k])   # ... and the kept are the nearest
    for r in range(R):
        n = min(cnt[r], mh)
        assert (np.diff(t[r, :n, 0]) >= 0).all() and (t[r, :n, 0] <= t[r, :n, 1]).all()
        assert (t[r, :n, 0] >= 0.2).all() and (t[r, :n, 1] <= 60.0).all()
        assert (b[r, n:] == -1).all() and len(set(b[r, :n])) == n
    z = co.stratified(rays, N)
    ls, li = co.sample_labels(z, t, b, cnt, ids)
    inside = ((t[:, None, :, 0] <= z[..., None]) & (z[..., None] <= t[:, None, :, 1]) &
              (np.arange(mh)[None, None, :] < np.minimum(cnt, mh)[:, None, None])).any(-1)
    assert np.array_equal(ls >= 0, inside) and np.array_equal(li >= 0, inside)
    if M:
        assert set(np.unique(ls)) <= set(ids[:, 0]) | {-1}


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 6), st.sampled_from([2, 4, 8]), st.integers(0, 2 ** 31 - 1))
def test_compositing_factorises_over_tiles(n_tiles, tile, seed):
    """For ANY split of a ray into equal tiles, per-tile records (transmittance factor + locally weighted sums) recombine to the
    plain front-to-back scan: the identity the fused MLP + compositing pass rests on (np_oracle.composite_by_tiles)."""
    from oracle import np_oracle as no
    rng = np.random.default_rng(seed)
    N, R, C, K = n_tiles * tile, 3, 4, 2
    raw = rng.normal(0, 2.0, (R, N, 4 + C + K))
    raw[:, :, 3] = rng.normal(0.0, 3.0, (R, N))          # about half the samples are empty (sigma <= 0)
    rays = np.concatenate([rng.normal(0, 1, (R, 3)), rng.normal(0, 1, (R, 3)) + 0.1, np.full((R, 1), 0.5), np.full((R, 1), 20.0)], 1)
    z = no.stratified(rays, N, t_rand=rng.random((R, N)))
    ls = np.where(rng.random((R, N)) < 0.5, rng.integers(0, C, (R, N)), -1)
    li = np.where(rng.random((R, N)) < 0.5, rng.integers(0, K, (R, N)), -1)
    a = no.composite(raw, z, rays, C, K, None, ls, li, 0, False)
    b = no.composite_by_tiles(raw, z, rays, C, K, tile, ls, li, False)
    for k in a:
        np.testing.assert_allclose(b[k], a[k], rtol=1e-11, atol=1e-12, err_msg=k)
    assert np.all(b["weights"] >= 0) and np.all(b["weights"].sum(-1) <= 1 + 1e-7)      # (the 1e-10 per factor lets the sum exceed 1 by ~N e-10)
