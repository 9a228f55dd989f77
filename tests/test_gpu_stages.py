"""GPU parity tests, one per stage of the path, all through the C-ABI:
HIP kernel vs the oracle on identical seeded inputs, vs the committed golden vectors, and at
BASELINE sizes through size-independent properties.

Bars (BASELINE.json north_star): bit-exact z / sample indices / ray-bbox hits / labels;
fp32 colour, depth, logits within 1e-4 abs on identical stage inputs; bf16 MLP against the
bf16-emulating oracle within 1e-2 abs (one bf16 ulp of a hidden activation is 4e-3 relative;
the only legitimate differences are rounding flips caused by fp32 accumulation order)."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import torch_oracle as to
from panopticnerf_amd import ops, synthetic

pytestmark = pytest.mark.gpu


def T(x, dev, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev).contiguous()


def N_(t):
    return t.detach().cpu().numpy()


def _rays(rng, R, near=0.5, far=60.0):
    o = rng.normal(0, 1, (R, 3)) + np.array([0, 1.5, 0])
    d = rng.normal(0, 0.3, (R, 3)) + np.array([0, 0, 1.0])
    return np.concatenate([o, d, np.full((R, 1), near), np.full((R, 1), far)], 1).astype(np.float32)


# ----------------------------------------------------------------------------- a3
@pytest.mark.parametrize("N", [1, 32, 64, 192])
@pytest.mark.parametrize("lindisp", [False, True])
def test_stratified_bitexact(dev, N, lindisp):
    rng = np.random.default_rng(N)
    rays = _rays(rng, 777)
    tr = rng.random((777, N)).astype(np.float32)
    for t in (None, tr):
        z = ops.stratified(T(rays, dev), N, lindisp, None if t is None else T(t, dev))
        assert np.array_equal(N_(z), co.stratified(rays, N, lindisp, t))
    pts = ops.points(T(rays, dev), z)
    assert np.array_equal(N_(pts), co.points(rays, N_(z)))


def test_stratified_golden_and_empty(dev, golden):
    g = golden
    Nc = int(g["dims"][1])
    r = T(g["rays"], dev)
    assert np.array_equal(N_(ops.stratified(r, Nc)), g["z_det"])
    assert np.array_equal(N_(ops.stratified(r, Nc, True)), g["z_lindisp"])
    zp = ops.stratified(r, Nc, False, T(g["t_rand"], dev))
    assert np.array_equal(N_(zp), g["z_perturb"])
    assert np.array_equal(N_(ops.points(r, zp)), g["pts"])
    assert ops.stratified(r[:0], Nc).shape == (0, Nc)           # empty input


# ----------------------------------------------------------------------------- a4
@pytest.mark.parametrize("L", [0, 4, 10])
def test_embed(dev, L, golden):
    rng = np.random.default_rng(L)
    x = (rng.normal(0, 30, (1001, 3))).astype(np.float32)      # arguments up to 2^9 * 100
    e = N_(ops.embed(T(x, dev), L))
    np.testing.assert_allclose(e, co.embed(x, L), atol=2e-6)
    if L:
        np.testing.assert_allclose(N_(ops.embed(T(golden["embed_x"], dev), L)), golden[f"embed_L{L}"], atol=2e-6)


# ----------------------------------------------------------------------------- a8
def test_bbox_hits_and_labels_bitexact(dev, golden):
    g = golden
    R, Nc, Nf, C, K, M, MH = (int(v) for v in g["dims"])
    ht, hb, hc = ops.bbox_hits(T(g["rays"], dev), T(g["box"], dev), MH)
    assert np.array_equal(N_(hb), g["hit_box"]) and np.array_equal(N_(hc), g["hit_count"])
    assert np.array_equal(N_(ht), g["hit_t"])
    ls, li = ops.sample_labels(T(g["z_perturb"], dev), ht, hb, hc, T(g["box_ids"], dev))
    assert np.array_equal(N_(ls), g["label_sem"]) and np.array_equal(N_(li), g["label_inst"])
    # larger seeded case incl. max_hits clipping and an empty table
    rng = np.random.default_rng(9)
    rays = synthetic.camera_rays()[::211].numpy()
    box, ids = synthetic.random_boxes(64, 45, 32, seed=9)
    for mh in (2, 8):
        a = ops.bbox_hits(T(rays, dev), box.to(dev), mh)
        b = co.bbox_hits(rays, box.numpy(), mh)
        assert np.array_equal(N_(a[1]), b[1]) and np.array_equal(N_(a[2]), b[2]) and np.array_equal(N_(a[0]), b[0])
    assert b[2].max() == 8 or b[2].max() > 2
    z = co.stratified(rays, 64, t_rand=rng.random((rays.shape[0], 64)).astype(np.float32))
    l1 = ops.sample_labels(T(z, dev), *a, ids.to(dev))
    l2 = co.sample_labels(z, *b, ids.numpy())
    assert np.array_equal(N_(l1[0]), l2[0]) and np.array_equal(N_(l1[1]), l2[1])
    assert (l2[0] >= 0).any()
    e = ops.bbox_hits(T(rays, dev), torch.zeros((0, 15), device=dev), 4)
    assert int(e[2].sum()) == 0 and int((e[1] != -1).sum()) == 0


def test_bbox_hits_keep_nearest_and_report_overflow(dev):
    # a ray crossing more boxes than max_hits must keep the NEAREST ones (not the first in table order) and
    # report the overflow through hit_count; bit-exact with the C oracle.
    b = np.zeros((6, 15), np.float32)
    b[:, 0:3] = [(0, 0, zc) for zc in (50, 10, 40, 20, 30, 45)]
    b[:, 3:12] = np.eye(3).reshape(-1)
    b[:, 12:15] = 1
    rays = np.array([[0.5, 0.5, 0, 0, 0, 1, 0.1, 80]], np.float32)
    ids = np.stack([np.arange(6), np.arange(6)], 1).astype(np.int32)
    for mh in (1, 3, 6, 8):
        a = ops.bbox_hits(T(rays, dev), T(b, dev), mh)
        c = co.bbox_hits(rays, b, mh)
        assert np.array_equal(N_(a[0]), c[0]) and np.array_equal(N_(a[1]), c[1]) and np.array_equal(N_(a[2]), c[2])
        assert int(a[2][0]) == 6
        assert list(N_(a[1])[0][: min(mh, 6)]) == [1, 3, 4, 2, 5, 0][: min(mh, 6)]
        z = np.array([[10.0, 20.0, 30.0, 40.0]], np.float32)
        l1 = ops.sample_labels(T(z, dev), *a, T(ids, dev))
        l2 = co.sample_labels(z, *c, ids)
        assert np.array_equal(N_(l1[0]), l2[0])
    # random table, max_hits far below the hit count: every kept interval starts no later than any dropped one
    rays = synthetic.camera_rays()[::1531].numpy()
    box, _ = synthetic.random_boxes(64, 45, 32, seed=5)
    box[:, 12:15] *= 6.0                                   # big boxes: many hits per ray
    a3, a64 = ops.bbox_hits(T(rays, dev), box.to(dev), 3), ops.bbox_hits(T(rays, dev), box.to(dev), 64)
    assert np.array_equal(N_(a3[2]), N_(a64[2])) and int(a3[2].max()) > 3
    assert np.array_equal(N_(a3[0]), N_(a64[0])[:, :3]) and np.array_equal(N_(a3[1]), N_(a64[1])[:, :3])
    t_in = N_(a64[0])[..., 0]
    cnt = N_(a64[2])
    for r in range(rays.shape[0]):
        assert (np.diff(t_in[r, : cnt[r]]) >= 0).all()


def test_bbox_axis_parallel_edge_cases(dev):
    b = np.zeros((1, 15), np.float32)
    b[0, 0:3] = (0, 0, 10)
    b[0, 3:12] = np.eye(3).reshape(-1)
    b[0, 12:15] = (1, 1, 1)
    rays = np.array([[0.5, 0.5, 0, 0, 0, 1, 0.1, 50], [1.5, 0.5, 0, 0, 0, 1, 0.1, 50],
                     [1.0, 0.0, 0, 0, 0, 1, 0.1, 50], [0, 0, 0, 0, 0, -1, 0.1, 50]], np.float32)
    a = ops.bbox_hits(T(rays, dev), T(b, dev), 2)
    c = co.bbox_hits(rays, b, 2)
    assert np.array_equal(N_(a[2]), c[2]) and np.array_equal(N_(a[1]), c[1]) and np.array_equal(N_(a[0]), c[0])
    assert list(c[2][:2]) == [1, 0]


# ----------------------------------------------------------------------------- a7
@pytest.mark.parametrize("Nc,Nf", [(64, 128), (32, 32), (64, 64), (8, 5)])
def test_sample_pdf_bitexact(dev, Nc, Nf):
    rng = np.random.default_rng(Nc * 1000 + Nf)
    R = 513
    rays = _rays(rng, R)
    z = co.stratified(rays, Nc, t_rand=rng.random((R, Nc)).astype(np.float32))
    # realistic weights: a few peaks over a floor, plus rows of exact zeros (flat CDF, denom<1e-5 branch)
    w = (np.exp(-0.5 * ((np.arange(Nc)[None] - rng.uniform(0, Nc, (R, 1))) / rng.uniform(0.5, 6, (R, 1))) ** 2)
         * rng.uniform(0, 1, (R, 1))).astype(np.float32)
    w[::7] = 0.0
    u = rng.random((R, Nf)).astype(np.float32)
    for uu in (None, u):
        zf, zs, inds = ops.sample_pdf(T(z, dev), T(w, dev), Nf, None if uu is None else T(uu, dev))
        zs_c, inds_c = co.sample_pdf(z, w, Nf, uu)
        assert np.array_equal(N_(inds), inds_c), "sample indices must be bit-exact"
        assert np.array_equal(N_(zs), zs_c)
        assert np.array_equal(N_(zf), co.merge_sorted(z, zs_c))


@pytest.mark.parametrize("Nc", [64, 66, 67, 128])
def test_sample_pdf_cdf_paths_bitexact(dev, Nc):
    # the CDF is torch's sequential f64 running sum; the kernel runs it as a parallel scan only where that is provably the
    # same bits (<= 64 pdf values, each 0 or >= 2^-28, sum < 2) and in order otherwise: weights that leave the exact regime
    # (one huge weight -> pdf values below 2^-28; denormal / 1e-30 weights; more than 64 values) must still match the C oracle
    rng = np.random.default_rng(Nc)
    R, Nf = 260, 96
    rays = _rays(rng, R)
    z = co.stratified(rays, Nc, t_rand=rng.random((R, Nc)).astype(np.float32))
    w = rng.uniform(0, 1, (R, Nc)).astype(np.float32)
    w[0::5, 7] = 3.0e4          # pdf of the floor = 1e-5 / 3e4 = 3e-10 < 2^-28: sequential path
    w[1::5] = 1.0e-30           # all at the 1e-5 floor: uniform pdf
    w[2::5, ::2] = 0.0
    w[3::5, 3] = 1.0e9
    u = rng.random((R, Nf)).astype(np.float32)
    for uu in (None, u):
        zf, zs, inds = ops.sample_pdf(T(z, dev), T(w, dev), Nf, None if uu is None else T(uu, dev))
        zs_c, inds_c = co.sample_pdf(z, w, Nf, uu)
        assert np.array_equal(N_(inds), inds_c) and np.array_equal(N_(zs), zs_c)
        assert np.array_equal(N_(zf), co.merge_sorted(z, zs_c))


def test_sample_pdf_golden(dev, golden):
    g = golden
    Nf = int(g["dims"][2])
    for tag, u in (("det", None), ("rand", g["u"])):
        zf, zs, inds = ops.sample_pdf(T(g["z_perturb"], dev), T(g["comp0_weights"], dev), Nf,
                                      None if u is None else T(u, dev))
        assert np.array_equal(N_(inds), g[f"pdf_{tag}_inds"]) and np.array_equal(N_(zs), g[f"pdf_{tag}_zs"])
        assert np.array_equal(N_(zf), g[f"pdf_{tag}_zfine"])


# ----------------------------------------------------------------------------- a6
@pytest.mark.parametrize("N", [4, 32, 64, 128, 192, 256])
@pytest.mark.parametrize("layout", ["channel", "sample"])
def test_composite_vs_oracle(dev, N, layout):
    rng = np.random.default_rng(N)
    R, C, K = 301, 7, 5          # ragged: not a multiple of the rays-per-wave of any N
    rays = _rays(rng, R)
    z = co.stratified(rays, N, t_rand=rng.random((R, N)).astype(np.float32))
    raw = rng.normal(0, 1, (R, N, 4 + C + K)).astype(np.float32)
    raw[..., 3] = rng.normal(0.02, 0.2, (R, N))
    noise = rng.normal(0, 0.05, (R, N)).astype(np.float32)
    ls, li = rng.integers(-1, C, (R, N)).astype(np.int32), rng.integers(-1, K, (R, N)).astype(np.int32)
    for sm, wb in ((0, False), (1, True)):
        ref = co.composite(raw, z, rays, C, K, noise=noise, label_sem=ls, label_inst=li, sem_mode=sm, white_bkgd=wb)
        if layout == "channel":
            rg = T(np.ascontiguousarray(raw.reshape(R * N, -1).T), dev)
        else:
            rg = T(raw, dev)
        out = ops.composite(rg, T(z, dev), T(rays, dev), C, K, layout == "channel", T(noise, dev), T(ls, dev),
                            T(li, dev), sm, wb)
        for k, v in ref.items():
            tol = 1e-4 if k != "depth" else 1e-4 * 60     # depth is in metres (far = 60): 1e-4 relative to range
            np.testing.assert_allclose(N_(out[k]), v, atol=tol, rtol=0, err_msg=f"{k} N={N} sem_mode={sm}")


def test_composite_golden_and_minimal_outputs(dev, golden):
    g = golden
    R, Nc, Nf, C, K, M, MH = (int(v) for v in g["dims"])
    rg = T(np.ascontiguousarray(g["raw"].reshape(R * Nc, -1).T), dev)
    for sm in (0, 1):
        out = ops.composite(rg, T(g["z_perturb"], dev), T(g["rays"], dev), C, K, True, T(g["noise"], dev),
                            T(g["label_sem"], dev), T(g["label_inst"], dev), sm, bool(sm))
        for k in ("rgb", "depth", "acc", "weights", "semantic", "instance", "fix_semantic", "fix_instance"):
            np.testing.assert_allclose(N_(out[k]), g[f"comp{sm}_{k}"], atol=1e-4 if k != "depth" else 6e-3, rtol=0)
    # no heads, no weights requested
    out = ops.composite(T(np.ascontiguousarray(g["raw"][..., :4].reshape(R * Nc, 4).T), dev), T(g["z_perturb"], dev),
                        T(g["rays"], dev), 0, 0, True, want_weights=False)
    assert set(out) == {"rgb", "depth", "acc"}


def test_composite_properties_at_frame_size(dev):
    # BASELINE size: one 1408x376 frame of 192-sample rays would need 33 GB of raw at C+K=77;
    # the properties are per ray, so a 65,536-ray slab (the renderer's chunk) covers the same code.
    R, N, C = 65536, 192, 45
    g = torch.Generator(device=dev).manual_seed(0)
    rays = synthetic.camera_rays()[:R].to(dev)
    z = ops.stratified(rays, N)
    raw = torch.randn((4 + C, R * N), generator=g, device=dev)
    raw[3] = raw[3] * 0.05 + 0.01
    out = ops.composite(raw, z, rays, C, 0, True)
    w = out["weights"]
    assert torch.isfinite(w).all() and (w >= 0).all() and (w.sum(1) <= 1 + 1e-4).all()
    assert torch.allclose(out["acc"], w.sum(1), atol=1e-4)
    assert (out["depth"] <= 100.0 * out["acc"] + 1e-2).all()
    # linearity of logit compositing: sem(a*x + b*y) = a*sem(x) + b*sem(y) with sigma fixed
    raw2 = raw.clone()
    raw2[4:] = torch.randn((C, R * N), generator=g, device=dev)
    raw3 = raw.clone()
    raw3[4:] = 0.5 * raw[4:] - 2.0 * raw2[4:]
    s1, s2 = out["semantic"], ops.composite(raw2, z, rays, C, 0, True)["semantic"]
    s3 = ops.composite(raw3, z, rays, C, 0, True)["semantic"]
    assert torch.allclose(s3, 0.5 * s1 - 2.0 * s2, atol=2e-4)
    # one-hot fixed field composites to a weighted histogram: rows sum to the labelled weight mass
    lab = torch.randint(-1, C, (R, N), generator=g, device=dev).int()
    fx = ops.composite(raw, z, rays, C, 0, True, label_sem=lab)["fix_semantic"]
    assert torch.allclose(fx.sum(1), (w * (lab >= 0)).sum(1), atol=1e-4)


# ----------------------------------------------------------------------------- a5
def _mlp_case(dev, D, W, skips, C, K, prec, seed, S_rays=9, N=37, channel_major=True):
    ocfg = to.mlp_config(D=D, W=W, skips=tuple(skips), n_sem=C, n_inst=K, head_W=W // 2)
    p = to.init_params(ocfg, seed=seed, sigma_bias=0.05)
    desc = ops.make_desc(D, W, skips[0] if skips else -1, 10, 4, C, K, W // 2, prec)
    img = ops.pack_mlp(desc, p).to(dev)
    rng = np.random.default_rng(seed)
    rays = _rays(rng, S_rays)
    z = co.stratified(rays, N, t_rand=rng.random((S_rays, N)).astype(np.float32))
    raw = ops.mlp_forward(desc, img, T(rays, dev), T(z, dev), channel_major=channel_major)
    raw = N_(raw)
    if channel_major:
        raw = raw.T.reshape(S_rays, N, -1)
    ref32 = to.run_network(p, ocfg, torch.tensor(rays), torch.tensor(z)).numpy()
    refbf = to.run_network(p, ocfg, torch.tensor(rays), torch.tensor(z), emulate_bf16=True).numpy()
    return raw, ref32, refbf


@pytest.mark.parametrize("geom", [(8, 256, [4], 45, 32), (8, 256, [4], 0, 0), (4, 128, [], 0, 0),
                                  (3, 128, [1], 7, 0), (8, 256, [4], 19, 0)])
def test_mlp_fp32_matches_oracle(dev, geom):
    raw, ref32, _ = _mlp_case(dev, *geom, "fp32", seed=11)
    np.testing.assert_allclose(raw, ref32, atol=1e-4, rtol=0)


@pytest.mark.parametrize("geom", [(8, 256, [4], 45, 32), (8, 256, [4], 0, 0), (4, 128, [], 0, 0),
                                  (3, 128, [1], 7, 0)])
def test_mlp_bf16_matches_bf16_oracle(dev, geom):
    raw, ref32, refbf = _mlp_case(dev, *geom, "bf16", seed=12)
    err = np.abs(raw - refbf)
    assert err.max() < 1e-2, f"max {err.max()}"
    assert np.median(err) < 5e-4
    assert np.abs(raw - ref32).max() < 6e-2          # and it is a bf16-accurate evaluation of the fp32 network


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_mlp_layouts_tails_and_golden(dev, prec, golden):
    g = golden
    # sample-major raw (the reference layout) == channel-major raw
    a, r32, rbf = _mlp_case(dev, 8, 256, [4], 6, 5, prec, seed=13, S_rays=5, N=33, channel_major=False)
    b, _, _ = _mlp_case(dev, 8, 256, [4], 6, 5, prec, seed=13, S_rays=5, N=33, channel_major=True)
    assert np.array_equal(a, b)
    # S smaller than one 32-sample tile, and S = 1
    for S_rays, N in ((1, 1), (1, 31), (3, 43)):
        raw, r32, rbf = _mlp_case(dev, 8, 256, [4], 6, 5, prec, seed=14, S_rays=S_rays, N=N)
        np.testing.assert_allclose(raw, r32 if prec == "fp32" else rbf, atol=1e-4 if prec == "fp32" else 1e-2)
    # golden vectors: small net with stored weights, big net from its seed
    R_, N_s = g["mlp_z"].shape
    C, K = int(g["dims"][3]), int(g["dims"][4])
    p_s = {k[len("mlp_s."):]: torch.tensor(v) for k, v in g.items() if k.startswith("mlp_s.")}
    desc = ops.make_desc(4, 128, 1, 10, 4, C, K, 64, prec)
    raw = ops.mlp_forward(desc, ops.pack_mlp(desc, p_s).to(dev), T(g["mlp_rays"], dev), T(g["mlp_z"], dev))
    np.testing.assert_allclose(N_(raw).T.reshape(R_, N_s, -1), g[f"mlp_s_raw_{prec}"],
                               atol=1e-4 if prec == "fp32" else 1e-2)
    p_b = to.init_params(to.mlp_config(n_sem=C, n_inst=K), seed=int(g["mlp_b_seed"][0]), sigma_bias=0.05)
    desc = ops.make_desc(n_sem=C, n_inst=K, precision=prec)
    raw = ops.mlp_forward(desc, ops.pack_mlp(desc, p_b).to(dev), T(g["mlp_rays"], dev), T(g["mlp_z"], dev))
    np.testing.assert_allclose(N_(raw).T.reshape(R_, N_s, -1), g[f"mlp_b_raw_{prec}"],
                               atol=1e-4 if prec == "fp32" else 1e-2)


def test_mlp_many_groups_persistent_loop(dev):
    # more sample groups than the persistent grid (512 workgroups x 128 samples): exercises the
    # grid-stride loop and the weight stream wrapping from the last chunk back to chunk 0
    C, K = 3, 2
    ocfg = to.mlp_config(n_sem=C, n_inst=K)
    p = to.init_params(ocfg, seed=21, sigma_bias=0.05)
    desc = ops.make_desc(n_sem=C, n_inst=K, precision="bf16")
    img = ops.pack_mlp(desc, p).to(dev)
    rays = synthetic.camera_rays()[::53][:4096].contiguous()
    z = ops.stratified(rays.to(dev), 64)
    raw = ops.mlp_forward(desc, img, rays.to(dev), z)            # 262,144 samples = 2048 groups
    assert torch.isfinite(raw).all()
    idx = torch.arange(0, 4096, 97)
    sub = ops.mlp_forward(desc, img, rays[idx].to(dev).contiguous(), z[idx.to(dev)].contiguous())
    full = raw.reshape(-1, 4096, 64)[:, idx.to(dev)].reshape(raw.shape[0], -1)
    assert torch.equal(full, sub), "a sample's result must not depend on which workgroup/iteration computed it"
    ref = to.run_network(p, ocfg, rays[idx], z[idx.to(dev)].cpu(), emulate_bf16=True)
    assert (sub.T.reshape(len(idx), 64, -1).cpu() - ref).abs().max() < 1e-2


@pytest.mark.gpu
def test_raw_channel_stride_is_free(dev):
    """The channel-major raw image may carry any channel stride >= R*N (ops.alloc_raw pads it to skew the channel
    rows across HBM channels): MLP and compositing results are bit-identical to the dense layout."""
    C, K = 5, 3
    ocfg = to.mlp_config(n_sem=C, n_inst=K)
    p = to.init_params(ocfg, seed=5, sigma_bias=0.05)
    desc = ops.make_desc(n_sem=C, n_inst=K, precision="bf16")
    img = ops.pack_mlp(desc, p).to(dev)
    R, N = 1000, 64
    rays = synthetic.camera_rays()[::401][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    dense = torch.empty((4 + C + K, R * N), device=dev)
    ops.mlp_forward(desc, img, rays, z, out=dense)
    padded = ops.mlp_forward(desc, img, rays, z)                       # default allocation: padded channel stride
    assert padded.stride(0) == R * N + ops.RAW_PAD and padded.stride(1) == 1
    odd = ops.alloc_raw(4 + C + K, R * N, dev, pad=6)                  # rows not 16 B aligned: the generic strided path
    ops.mlp_forward(desc, img, rays, z, out=odd)
    assert torch.equal(padded, dense) and torch.equal(odd, dense)
    lab = torch.randint(-1, C, (R, N), device=dev, dtype=torch.int32)
    a = ops.composite(dense, z, rays, C, K, True, None, lab, None)
    b = ops.composite(padded, z, rays, C, K, True, None, lab, None)
    assert all(torch.equal(a[k], b[k]) for k in a)
    # rows that are not 16-byte aligned take the generic strided kernel, whose per-ray sums associate differently from
    # the 8-lanes-x-8-samples mapping the aligned N = 64 image runs on: equal to rounding, not bit for bit
    b = ops.composite(odd, z, rays, C, K, True, None, lab, None)
    for k in a:
        assert (a[k] - b[k]).abs().max() < 2e-5 * max(1.0, float(a[k].abs().max())), k
    with pytest.raises(ValueError):
        ops.composite(dense.T.contiguous().T, z, rays, C, K, True)     # sample stride != 1


@pytest.mark.gpu
def test_gen_rays_bit_exact_with_c_oracle(dev):
    """pnr_gen_rays: whole frame and pixel subsets, bit-exact with pnro_gen_rays; empty input is a no-op."""
    import math
    intr = [552.554261, 560.25, 682.049453, 238.769549]
    c, s = math.cos(-0.7), math.sin(-0.7)
    c2w = np.array([[c, 0.02, s, 12.5], [-0.01, 1, 0.03, 1.55], [-s, 0.01, c, -3.25]], np.float32)
    full = ops.gen_rays(intr, c2w, 1408, 376, 0.5, 100.0, device=dev)
    assert np.array_equal(N_(full), co.gen_rays(intr, c2w, 1408, 376, 0.5, 100.0))
    g = torch.Generator().manual_seed(0)
    pix = torch.randint(0, 1408 * 376, (4099,), generator=g, dtype=torch.int32)
    sub = ops.gen_rays(intr, c2w, 1408, 376, 0.25, 80.0, pix=pix.to(dev))
    assert np.array_equal(N_(sub), co.gen_rays(intr, c2w, 1408, 376, 0.25, 80.0, pix.numpy()))
    assert torch.equal(sub, full[pix.long().to(dev)] * torch.tensor([1, 1, 1, 1, 1, 1, 0, 0], device=dev) + torch.tensor([0, 0, 0, 0, 0, 0, 0.25, 80.0], device=dev))
    assert ops.gen_rays(intr, c2w, 1408, 376, 0.5, 100.0, pix=torch.zeros(0, dtype=torch.int32, device=dev)).shape == (0, 8)


# ----------------------------------------------------------------------------- a5: the two time structures of the weight stream
@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("R,N", [(510, 192), (510, 64), (37, 32), (512, 192)])
def test_mlp_pingpong_equals_lockstep_bit_for_bit(dev, R, N, train):
    """k_mlp_pp (two wave groups in phase opposition, three LDS slots) against k_mlp_fused (lock-step double buffer): same
    arithmetic in the same order, so raw -- and the activations the training forward saves -- must be IDENTICAL, run after
    run, also for a ragged last sample group whose second wave group holds no valid sample (the case that exposed a
    refill piece read by a wave of the issuer's own group before a barrier covered it)."""
    import ctypes
    from types import SimpleNamespace as NS
    from panopticnerf_amd import _lib, make_network
    lib = _lib.load()
    torch.manual_seed(0)
    net = make_network(NS(N_importance=128, num_classes=5, num_instances=3)).to(dev)
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[:: (1408 * 376) // R][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    desc, img = net.packed(1, dev, "bf16")

    def run():
        if train:
            raw, acts = ops.mlp_forward_train(desc, img, rays, z)
            return raw.clone(), acts.view(torch.int16).clone()
        return ops.mlp_forward(desc, img, rays, z).clone(), None

    desc.schedule = 1                # lock-step everywhere (pnr_mlp_desc.schedule: same image, same arithmetic)
    want = run()
    desc.schedule = 2                # ping-pong everywhere
    for rep in range(4):
        got = run()
        assert torch.equal(got[0], want[0]), (rep, int((got[0] != want[0]).any(0).sum()))
        if train:
            assert torch.equal(got[1], want[1]), rep


# ----------------------------------------------------------------------------- a5 + a6 fused: no raw image round trip
@pytest.mark.parametrize("sem_mode", [0, 1])
@pytest.mark.parametrize("heads", [(45, 32), (6, 0), (0, 0)])
@pytest.mark.parametrize("R,N,labels,white", [(510, 192, True, False), (512, 64, True, True), (37, 32, False, False),
                                              (129, 256, True, False), (1, 96, False, True)])
def test_fused_mlp_composite_equals_two_kernel_path(dev, R, N, labels, white, heads, sem_mode):
    """pnr_mlp_forward_composite (per-tile records in the MLP's epilogue + k_composite_combine) against pnr_mlp_forward +
    pnr_composite on the same inputs: every map, the per-sample weights and the fixed (bbox-prior) fields.  The two paths
    associate the per-ray sums differently (per 32-sample tile, then over tiles), so they agree to fp32 rounding: 2e-6 of
    the map's scale; the fixed fields additionally round through 2^-30 fixed point per tile.  Ragged sample counts (R*N not a
    multiple of 256), one ray, several tiles per ray, labels on and off, white background.
    sem_mode 1 (semantic_activation = softmax, PNR_MLP_SOFTMAX): the learned fields composite softmax(logits) -- the fused pass
    normalises in the logit chunk's epilogue (v_exp_f32 / v_rcp_f32 in place of expf / the division: 4e-6), on the best image
    that has a softmax kernel (plan 2 since round 6); the plan-0 image refuses the flag."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    C, K = heads
    torch.manual_seed(R + N)
    net = make_network(NS(N_importance=128, num_classes=C, num_instances=K)).to(dev)
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[:: max(1, (1408 * 376) // R)][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    desc, img = net.packed(1, dev, "bf16")
    ls = li = None
    if labels and (C or K):
        g = torch.Generator(device=dev).manual_seed(1)
        hit = torch.rand((R, N), device=dev, generator=g) < 0.3
        if C:
            ls = torch.where(hit, torch.randint(0, C, (R, N), device=dev, generator=g), -1).int()
        if K:
            li = torch.where(hit, torch.randint(0, K, (R, N), device=dev, generator=g), -1).int()
    assert ops.fused_supported(desc, N, sem_mode)
    raw = ops.mlp_forward(desc, img, rays, z, channel_major=True)
    want = ops.composite(raw, z, rays, C, K, True, None, ls, li, sem_mode, white, True)
    if sem_mode == 1 and (C or K):
        d_bad, i_bad = net.packed(1, dev, "bf16", fused=False)         # the classic image has no softmax kernel
        with pytest.raises(RuntimeError, match="softmax compositing"):
            ops.mlp_forward_composite(d_bad, i_bad, rays, z, ls, li, white, True, sem_mode=1)
        desc, img = net.packed(1, dev, "bf16", fused=ops.fused_image(sem_mode))
        assert desc.plan == 2          # round 6: k_mlp_tt_sm_* (plan 1 = k_mlp_pp's softmax epilogue is compared with it bit for bit below)
        # sum_c sum_i w_i p_ic = sum_i w_i: the mode is really on
        assert float((want["semantic"].sum(-1) - want["acc"]).abs().max()) < 1e-4
    for rep in range(2):
        got = ops.mlp_forward_composite(desc, img, rays, z, ls, li, white, True, sem_mode=sem_mode)
        assert set(got) == set(want)
        for k in want:
            scale = max(1.0, float(want[k].abs().max()))
            err = float((got[k] - want[k]).abs().max())
            tol = 4e-6 if k.startswith("fix_") or (sem_mode == 1 and k in ("semantic", "instance")) else 2e-6
            assert err <= tol * scale * (N // 32), (k, err, scale, rep)
    # without the weights / labels the other outputs are unchanged
    lean = ops.mlp_forward_composite(desc, img, rays, z, None, None, white, False, sem_mode=sem_mode)
    assert "weights" not in lean and "fix_semantic" not in lean
    assert torch.equal(lean["rgb"], got["rgb"]) and torch.equal(lean["depth"], got["depth"])


@pytest.mark.parametrize("heads,plan", [((45, 32), 2), ((45, 0), 2), ((19, 8), 2), ((4, 0), 2), ((0, 0), 2), ((0, 32), 0), ((100, 0), 0), ((19, 40), 0)])
@pytest.mark.parametrize("R,N", [(510, 192), (129, 64)])
def test_fused_inference_plan_equals_classic_plan_bit_for_bit(dev, R, N, heads, plan):
    """The fused-inference plans (pnr_mlp_fused_plan: 1 = both head hidden layers first, then the semantic and instance logit
    layers as ONE chunk of interleaved transposed blocks, k_mlp_pp; 2 = the two-tile assembly kernel k_mlp_tt's image, where the
    geometry has one) are the same arithmetic per layer as the classic chunk order: every output of pnr_mlp_forward_composite is
    bit-identical between the images.  Geometries without such a kernel (an instance head alone, more than 2 + 1 logit blocks) report plan 0
    and keep the classic order; the classic kernels refuse a plan-1 / plan-2 image."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    C, K = heads
    torch.manual_seed(R + N + C)
    net = make_network(NS(N_importance=128, num_classes=C, num_instances=K)).to(dev)
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[:: max(1, (1408 * 376) // R)][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    d0, img0 = net.packed(1, dev, "bf16")
    d1, img1 = net.packed(1, dev, "bf16", fused=True)
    assert d0.plan == 0 and d1.plan == plan == ops.fused_plan(d0)
    g = torch.Generator(device=dev).manual_seed(1)
    hit = torch.rand((R, N), device=dev, generator=g) < 0.3
    ls = torch.where(hit, torch.randint(0, max(C, 1), (R, N), device=dev, generator=g), -1).int() if C else None
    li = torch.where(hit, torch.randint(0, max(K, 1), (R, N), device=dev, generator=g), -1).int() if K else None
    a = ops.mlp_forward_composite(d0, img0, rays, z, ls, li, False, True)
    b = ops.mlp_forward_composite(d1, img1, rays, z, ls, li, False, True)
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))
    if plan:
        if plan == 1:
            assert img1.numel() < img0.numel()      # one bias fragment for the merged chunk instead of one per logit block
        with pytest.raises(RuntimeError, match="pnr_mlp_forward_composite only"):
            ops.mlp_forward(d1, img1, rays, z, channel_major=True)


def _tiles_workspace(desc, img, rays, z):
    """pnr_mlp_forward_tiles into a 0xAB-filled workspace: (records (tiles, rec_floats), quadruples (S, 4))"""
    import ctypes
    from panopticnerf_amd import _lib
    lib = _lib.load()
    R, N = z.shape
    S = R * N
    nbytes = lib.pnr_mlp_forward_composite_workspace_bytes(ctypes.byref(desc), R, N, 0)
    ws = torch.full((int(nbytes),), 0xAB, device=z.device, dtype=torch.uint8)
    _lib.check(lib.pnr_mlp_forward_tiles(ctypes.byref(desc), ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(rays.data_ptr()),
                                         ctypes.c_void_p(z.data_ptr()), R, N, ctypes.c_void_p(ws.data_ptr()),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pnr_mlp_forward_tiles")
    rf = (1 + desc.n_sem + desc.n_inst + 3) & ~3
    pad = (S + 255) // 256 * 8
    rec = ws[: pad * rf * 4].view(torch.float32).reshape(pad, rf)[: (S + 31) // 32, : 1 + desc.n_sem + desc.n_inst]
    ps = ws[pad * rf * 4: pad * rf * 4 + S * 16].view(torch.float32).reshape(S, 4)
    return rec.clone(), ps.clone()


@pytest.mark.parametrize("heads", [(45, 32), (19, 8), (64, 1), (45, 0), (19, 0), (0, 0)])
@pytest.mark.parametrize("R,N", [(300, 192), (37, 96), (1001, 64), (7, 32), (256, 256), (2051, 32)])
def test_two_tile_assembly_kernel_equals_pingpong_bit_for_bit(dev, R, N, heads):
    """k_mlp_tt (hand-placed gfx950 assembly, one wave per SIMD, two 32-sample tiles per wave, plan-2 image; csrc/asm/gen_mlp_tt.py)
    against k_mlp_pp<fused, plan 1> on the same network, rays and z: the per-tile records (Q, every semantic and instance logit
    sum) and the per-sample quadruples (lw, r, g, b) are the SAME BITS -- the arithmetic per value is operation for operation
    the ping-pong kernel's.  One and two semantic logit blocks, ragged last groups, 1..8 tiles per ray, launches of one and of
    several groups per workgroup; run twice (a second launch must not depend on what the first left in the LDS / registers).
    Networks without an instance head or without any head (BASELINE configs 2 - 4: kernels s<n>i0 / s0i0, whose 45 / 42 image chunks
    are padded to a multiple of four with dummy chunks in the kernel only) against plan 1, or -- no heads: there is no plan 1 --
    against the ping-pong kernel on the classic image."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    C, K = heads
    torch.manual_seed(R + N + C)
    net = make_network(NS(N_importance=128, num_classes=C, num_instances=K)).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[:: max(1, (1408 * 376) // R)][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    d1, i1 = net.packed(1, dev, "bf16", fused=1)
    d2, i2 = net.packed(1, dev, "bf16", fused=2)
    assert d1.plan == (1 if C else 0) and d2.plan == 2
    rec1, ps1 = _tiles_workspace(d1, i1, rays, z)
    for rep in range(2):
        rec2, ps2 = _tiles_workspace(d2, i2, rays, z)
        assert torch.equal(rec1.view(torch.int32), rec2.view(torch.int32)), (rep, int((rec1.view(torch.int32) != rec2.view(torch.int32)).sum()))
        assert torch.equal(ps1.view(torch.int32), ps2.view(torch.int32)), (rep, int((ps1.view(torch.int32) != ps2.view(torch.int32)).sum()))
    # ... and through the whole fused call: every map
    a = ops.mlp_forward_composite(d1, i1, rays, z, None, None, False, True)
    b = ops.mlp_forward_composite(d2, i2, rays, z, None, None, False, True)
    for k in a:
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("heads", [(45, 32), (19, 8), (64, 1), (45, 0), (19, 0)])
@pytest.mark.parametrize("R,N", [(300, 192), (37, 96), (1001, 64), (7, 32), (2051, 32)])
def test_two_tile_softmax_kernels_equal_pingpong_bit_for_bit(dev, R, N, heads):
    """semantic_activation = softmax in the two-tile assembly form (round 6: k_mlp_tt_sm_s<n>i<m>, plan 2; tail_softmax of
    csrc/asm/gen_mlp_tt.py) against the ping-pong kernel's softmax epilogue (fuse_softmax_t, plan 1): operation for operation the same
    arithmetic -- the max_raw chain, the xor_max / xor_add butterflies with v_permlane16_swap for the xor-16 step, exp2(fma(x, log2 e,
    -m log2 e)), lwr * rcp(den), the FMA chains -- so records and quadruples are the SAME BITS, one and two semantic blocks, heads
    with padded channels (their logits must count as -inf), ragged groups; run twice; then every map of the whole fused call."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    C, K = heads
    torch.manual_seed(R + N + C + 1)
    net = make_network(NS(N_importance=128, num_classes=C, num_instances=K)).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    with torch.no_grad():           # logits of a few units' spread, so that the normalisation is not a near-uniform one
        for head in (getattr(net.nerf_1, "semantic_linears", None), getattr(net.nerf_1, "instance_linears", None)):
            if head is not None and len(head):
                head[-1].weight.mul_(40.0)
                head[-1].bias.add_(torch.linspace(-3.0, 3.0, head[-1].bias.numel(), device=dev))
    rays = synthetic.camera_rays()[:: max(1, (1408 * 376) // R)][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    d1, i1 = net.packed(1, dev, "bf16", fused=1)
    d2, i2 = net.packed(1, dev, "bf16", fused="softmax")
    assert d1.plan == 1 and d2.plan == 2
    d1, d2 = ops.desc_for_mode(d1, 1), ops.desc_for_mode(d2, 1)
    rec1, ps1 = _tiles_workspace(d1, i1, rays, z)
    assert float(rec1[:, 1:1 + C].abs().max()) > 0
    for rep in range(2):
        rec2, ps2 = _tiles_workspace(d2, i2, rays, z)
        assert torch.equal(ps1.view(torch.int32), ps2.view(torch.int32)), rep
        bad = rec1.view(torch.int32) != rec2.view(torch.int32)
        assert not bool(bad.any()), (rep, int(bad.sum()), float((rec1 - rec2).abs().max()))
    a = ops.mlp_forward_composite(d1, i1, rays, z, None, None, False, True, sem_mode=1)
    b = ops.mlp_forward_composite(d2, i2, rays, z, None, None, False, True, sem_mode=1)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # probabilities: every sample's channels sum to one, so the composited semantic map sums to the opacity
    assert float((b["semantic"].sum(-1) - b["acc"]).abs().max()) < 1e-4


@pytest.mark.parametrize("sem_mode", [0, 1])
@pytest.mark.parametrize("heads", [(45, 32), (19, 8), (64, 0), (19, 0)])
@pytest.mark.parametrize("R,N", [(300, 192), (1001, 64), (7, 32), (2051, 32)])
def test_two_tile_kernels_for_heads_that_read_the_feature(dev, R, N, heads, sem_mode):
    """head_tap = feature (the heads read the feature_linear output instead of the trunk output) in the two-tile
    assembly form (round 6: k_mlp_tt_f[sm]_s<n>i<m>: the last g block lives in the gamma(d) registers and the head hidden layers'
    outputs in h, so that F survives the views layer) against the ping-pong kernel on the plan-1 image: the same arithmetic --
    records, quadruples and every map bit for bit, logits and softmax compositing -- and the network really taps the feature (its
    maps differ from the same weights tapped at the trunk)."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    C, K = heads
    torch.manual_seed(R + N + C + 2)
    net = make_network(NS(N_importance=128, num_classes=C, num_instances=K, head_tap="feature")).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[:: max(1, (1408 * 376) // R)][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    d1, i1 = net.packed(1, dev, "bf16", fused=1)
    d2, i2 = net.packed(1, dev, "bf16", fused="softmax" if sem_mode else 2)
    assert d1.plan == 1 and d2.plan == 2 and d2.head_tap == 1
    d1, d2 = ops.desc_for_mode(d1, sem_mode), ops.desc_for_mode(d2, sem_mode)
    rec1, ps1 = _tiles_workspace(d1, i1, rays, z)
    for rep in range(2):
        rec2, ps2 = _tiles_workspace(d2, i2, rays, z)
        assert torch.equal(ps1.view(torch.int32), ps2.view(torch.int32)), rep
        assert torch.equal(rec1.view(torch.int32), rec2.view(torch.int32)), (rep, float((rec1 - rec2).abs().max()))
    a = ops.mlp_forward_composite(d1, i1, rays, z, None, None, False, True, sem_mode=sem_mode)
    b = ops.mlp_forward_composite(d2, i2, rays, z, None, None, False, True, sem_mode=sem_mode)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    net.nerf_1.head_tap = "trunk"           # same weights, the other tap: another function
    net.invalidate_packed()
    dt, it = net.packed(1, dev, "bf16", fused=True)
    assert dt.head_tap == 0
    c = ops.mlp_forward_composite(dt, it, rays, z, None, None, False, True)
    assert float((c["semantic"] - ops.mlp_forward_composite(ops.desc_for_mode(d2, 0), i2, rays, z, None, None, False, True)["semantic"]).abs().max()) > 1e-4


@pytest.mark.parametrize("tap", ["trunk", "feature"])
@pytest.mark.parametrize("heads", [(45, 32), (19, 8), (64, 0), (19, 0)])
@pytest.mark.parametrize("R,N", [(300, 192), (1001, 64), (7, 32), (2051, 32)])
def test_two_tile_kernels_for_single_linear_heads(dev, R, N, heads, tap):
    """head_depth = 1 (heads of one Linear W -> n, read from the trunk output) in the two-tile assembly form
    (round 6: k_mlp_tt_d1_s<n>i<m>, plan 2) against the ping-pong kernel's fused pass on the classic image (plan 0: there is no
    plan 1 for this depth).  The two reduce a tile's logits differently (transposed FMA chains against 32-lane butterflies), so
    records and maps agree to fp32 rounding -- 2e-6 of the field's scale per sample of the tile -- while Q and the per-sample
    quadruples (lw, r, g, b), which do not go through the logit layers, are the same bits; a second launch must not depend on what
    the first left behind; and both match the two-kernel path (mlp_forward + composite) to the fused pass's usual tolerance."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    C, K = heads
    torch.manual_seed(R + N + C)
    net = make_network(NS(N_importance=128, num_classes=C, num_instances=K, head_depth=1, head_tap=tap)).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[:: max(1, (1408 * 376) // R)][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    d0, i0 = net.packed(1, dev, "bf16", fused=0)
    d2, i2 = net.packed(1, dev, "bf16", fused=2)
    assert d0.plan == 0 and d2.plan == 2 and d2.head_depth == 1 and d2.head_tap == int(tap == "feature")
    rec0, ps0 = _tiles_workspace(d0, i0, rays, z)
    for rep in range(2):
        rec2, ps2 = _tiles_workspace(d2, i2, rays, z)
        assert torch.equal(ps0.view(torch.int32), ps2.view(torch.int32)), rep
        assert torch.equal(rec0[:, 0].view(torch.int32), rec2[:, 0].view(torch.int32)), rep          # Q
        scale = float(rec0[:, 1:1 + C + K].abs().max()) + 1e-6
        assert float((rec0[:, 1:1 + C + K] - rec2[:, 1:1 + C + K]).abs().max()) <= 64e-6 * scale, rep
    a = ops.mlp_forward_composite(d0, i0, rays, z, None, None, False, True)
    b = ops.mlp_forward_composite(d2, i2, rays, z, None, None, False, True)
    dd, ii = net.packed(1, dev, "bf16")
    raw = ops.mlp_forward(dd, ii, rays, z, channel_major=True)
    c = ops.composite(raw, z, rays, C, K, True, None, None, None, 0, False, True)
    for k in a:
        sc = max(1.0, float(a[k].abs().max()))
        assert float((a[k] - b[k]).abs().max()) <= 4e-6 * sc * max(1, N // 32), k
        assert float((c[k] - b[k]).abs().max()) <= 4e-6 * sc * max(1, N // 32), k
    # ... and with softmax compositing (k_mlp_tt_d1sm_*: there is no ping-pong softmax kernel at this depth) against the two-kernel path
    ds, is_ = net.packed(1, dev, "bf16", fused="softmax")
    assert ds.plan == 2 and ops.fused_supported(ds, N, 1)
    cs = ops.composite(raw, z, rays, C, K, True, None, None, None, 1, False, True)
    for rep in range(2):
        bs = ops.mlp_forward_composite(ds, is_, rays, z, None, None, False, True, sem_mode=1)
        for k in cs:
            sc = max(1.0, float(cs[k].abs().max()))
            assert float((cs[k] - bs[k]).abs().max()) <= 4e-6 * sc * max(1, N // 32), (k, rep)
    assert float((bs["semantic"].sum(-1) - bs["acc"]).abs().max()) < 1e-4


@pytest.mark.parametrize("tap", ["trunk", "feature"])
@pytest.mark.parametrize("depth", [2, 1])
@pytest.mark.parametrize("C", [96, 70])
@pytest.mark.parametrize("R,N", [(300, 192), (1001, 64), (7, 32)])
def test_two_tile_kernels_for_a_third_semantic_block(dev, R, N, C, depth, tap):
    """65..96 semantic classes without an instance head (round-5 verdict item 3: `k_mlp_tt_*_s3i0`, plan 2): three logit blocks
    per tile in six accumulators, the local weights in the remaining two.  The ping-pong kernel has no merged logit chunk for three
    blocks (no plan 1), so the reference is its fused pass on the classic image (plan 0; per-block butterflies against transposed
    FMA chains: records to fp32 rounding, Q and the quadruples bit for bit) and the two-kernel path -- which is also the softmax
    reference (`k_mlp_tt_*sm_s3i0`); padded channels (70 = 2 blocks + 6) must not leak into the softmax denominator."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    torch.manual_seed(R + N + C + depth)
    net = make_network(NS(N_importance=128, num_classes=C, num_instances=0, head_depth=depth, head_tap=tap)).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[:: max(1, (1408 * 376) // R)][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    d0, i0 = net.packed(1, dev, "bf16", fused=0)
    d2, i2 = net.packed(1, dev, "bf16", fused=2)
    assert d0.plan == 0 and d2.plan == 2 and net.packed(1, dev, "bf16", fused=True)[0].plan == 2
    rec0, ps0 = _tiles_workspace(d0, i0, rays, z)
    for rep in range(2):
        rec2, ps2 = _tiles_workspace(d2, i2, rays, z)
        assert torch.equal(ps0.view(torch.int32), ps2.view(torch.int32)), rep
        assert torch.equal(rec0[:, 0].view(torch.int32), rec2[:, 0].view(torch.int32)), rep          # Q
        scale = float(rec0[:, 1:1 + C].abs().max()) + 1e-6
        assert float((rec0[:, 1:1 + C] - rec2[:, 1:1 + C]).abs().max()) <= 64e-6 * scale, rep
    a = ops.mlp_forward_composite(d0, i0, rays, z, None, None, False, True)
    b = ops.mlp_forward_composite(d2, i2, rays, z, None, None, False, True)
    dd, ii = net.packed(1, dev, "bf16")
    raw = ops.mlp_forward(dd, ii, rays, z, channel_major=True)
    c = ops.composite(raw, z, rays, C, 0, True, None, None, None, 0, False, True)
    for k in a:
        sc = max(1.0, float(a[k].abs().max()))
        assert float((a[k] - b[k]).abs().max()) <= 4e-6 * sc * max(1, N // 32), k
        assert float((c[k] - b[k]).abs().max()) <= 4e-6 * sc * max(1, N // 32), k
    ds, is_ = net.packed(1, dev, "bf16", fused="softmax")
    assert ds.plan == 2 and ops.fused_supported(ds, N, 1)
    cs = ops.composite(raw, z, rays, C, 0, True, None, None, None, 1, False, True)
    for rep in range(2):
        bs = ops.mlp_forward_composite(ds, is_, rays, z, None, None, False, True, sem_mode=1)
        for k in cs:
            sc = max(1.0, float(cs[k].abs().max()))
            assert float((cs[k] - bs[k]).abs().max()) <= 4e-6 * sc * max(1, N // 32), (k, rep)
    assert float((bs["semantic"].sum(-1) - bs["acc"]).abs().max()) < 1e-4


@pytest.mark.parametrize("R,N", [(3000, 192), (1001, 64), (7, 32)])
def test_two_tile_launch_on_a_share_of_the_device_is_the_same_launch(dev, R, N):
    """PNR_MLP_WG_CAP(n) in pnr_mlp_desc.flags: the plan-2 launch on at most n workgroups (the persistent grid walks the 256-sample groups
    with a stride of its size, every group is independent) -- records, quadruples and maps are the whole-device launch's bit for bit,
    for caps below, at and above the number of groups; the cap does not change the image (same packed bytes)."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    torch.manual_seed(R + N)
    net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[:: max(1, (1408 * 376) // R)][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    d, img = net.packed(1, dev, "bf16", fused=True)
    assert d.plan == 2
    ref = ops.mlp_forward_composite(d, img, rays, z, None, None, False, True)
    for cap in (8, 64, 192, 248, 511):
        out = ops.mlp_forward_composite(d, img, rays, z, None, None, False, True, wg_cap=cap)
        for k in ref:
            assert torch.equal(ref[k], out[k]), (cap, k)


def test_two_tile_kernel_is_the_default_where_it_exists_and_can_be_capped(dev, monkeypatch):
    """The renderer's fused inference pass packs the BEST plan the geometry has (pnr_mlp_fused_plan: 2 = k_mlp_tt for the benched
    network); PNR_FUSED_PLAN=1 (A/B runs) caps it at the ping-pong kernel's plan.  Same frame either way, bit for bit."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network, make_renderer
    cfg = NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32, precision="bf16", chunk_size=4096)
    torch.manual_seed(11)
    net = make_network(cfg).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[::517][:1000].contiguous().to(dev)
    box, ids = synthetic.random_boxes(16, 45, 32)
    batch = {"rays": rays[None], "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    assert net.packed(1, dev, fused=True)[0].plan == 2
    with torch.no_grad():
        out2 = make_renderer(cfg, net).render(batch)
        monkeypatch.setenv("PNR_FUSED_PLAN", "1")
        net.invalidate_packed()
        assert net.packed(1, dev, fused=True)[0].plan == 1
        out1 = make_renderer(cfg, net).render(batch)
    for k in out1:
        assert torch.equal(out1[k], out2[k]), k


def test_fused_path_is_what_the_renderer_runs_and_can_be_switched_off(dev):
    """Renderer.render (inference, bf16) takes the fused pass by default; cfg.fuse_composite = False keeps mlp_forward +
    composite.  Both give the same maps to fp32 rounding, identical z (the coarse weights feed sample_pdf: a weight that
    differs in the last bit may move a fine sample across a bin edge, so z_vals_1 is compared by fraction)."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network, make_renderer
    cfg = dict(N_samples=64, N_importance=128, num_classes=7, num_instances=4, precision="bf16", chunk_size=4096)
    torch.manual_seed(3)
    net = make_network(NS(**cfg)).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[::1033][:500].contiguous().to(dev)
    box, ids = synthetic.random_boxes(16, 7, 4)
    batch = {"rays": rays[None], "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    with torch.no_grad():
        a = make_renderer(NS(**cfg), net).render(batch)
        b = make_renderer(NS(fuse_composite=False, **cfg), net).render(batch)
    assert torch.equal(a["z_vals_0"], b["z_vals_0"])
    for k in ("rgb_0", "depth_0", "acc_0", "semantic_0", "instance_0", "fix_semantic_0", "fix_instance_0", "weights_0"):
        x, y = a[k][0], b[k][0]
        assert float((x - y).abs().max()) <= 4e-6 * max(1.0, float(y.abs().max())), k
    # the fine samples are a continuous function of the coarse weights: last-bit differences there move them by ~1e-6 of the
    # ray length, which gamma() amplifies 2^9-fold -- the fine maps agree to ~1e-3, not to rounding
    # (a few samples sit where the coarse CDF is flat and jump with the last bit of a weight: judged by fraction)
    assert float(((a["z_vals_1"] - b["z_vals_1"]).abs() > 1e-3).float().mean()) < 1e-3
    for k in ("rgb_1", "acc_1", "semantic_1", "instance_1", "fix_semantic_1", "fix_instance_1"):
        x, y = a[k][0], b[k][0]
        assert float(torch.quantile((x - y).abs().flatten(), 0.99)) <= 5e-3 * max(1.0, float(y.abs().max())), k


def test_bbox_restricted_sampling_switch(dev):
    """cfg.bbox_sampling = 'hull' (SURVEY.md 9 item 2 as a config switch): pnr_restrict_rays is bit-exact with the C oracle, and a
    render with the switch on equals the bf16-emulating oracle run with the same switch -- samples of rays that hit boxes lie in
    the hull of their intervals, rays without a hit are sampled over [near, far] as before."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network, make_renderer
    rays = synthetic.camera_rays()[::997][:400].contiguous()
    box, ids = synthetic.random_boxes(40, 6, 3, seed=3)
    ht, hb, hc = co.bbox_hits(rays.numpy(), box.numpy(), 8)
    got = ops.restrict_rays(rays.to(dev), torch.tensor(ht).to(dev), torch.tensor(hc).to(dev))
    assert np.array_equal(got.cpu().numpy(), co.restrict_rays(rays.numpy(), ht, hc))
    cfg = dict(N_samples=32, N_importance=32, num_classes=6, num_instances=3, precision="bf16", D=4, W=128, skips=[1])
    torch.manual_seed(2)
    net = make_network(NS(**cfg)).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    batch = {"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    with torch.no_grad():
        a = make_renderer(NS(**cfg), net).render(batch)
        b = make_renderer(NS(bbox_sampling="hull", **cfg), net).render(batch)
    rr = co.restrict_rays(rays.numpy(), ht, hc)
    assert np.array_equal(b["z_vals_0"][0].cpu().numpy(), co.stratified(rr, 32))
    hit = torch.tensor(hc > 0)
    assert torch.equal(a["z_vals_0"][0].cpu()[~hit], b["z_vals_0"][0].cpu()[~hit]) and not torch.equal(a["z_vals_0"], b["z_vals_0"])
    oc = to.mlp_config(D=4, W=128, skips=(1,), n_sem=6, n_inst=3, head_W=64)
    prm = {"coarse": {k: v.detach().cpu() for k, v in net.nerf_0.state_dict().items()},
           "fine": {k: v.detach().cpu() for k, v in net.nerf_1.state_dict().items()}}
    want = to.render_rays(prm, oc, rays, 32, 32, box=box, box_ids=ids, emulate_bf16=True, bbox_sampling="hull")
    assert torch.equal(b["z_vals_0"][0].cpu(), want["z_vals_0"])
    for k in ("rgb_0", "acc_0", "fix_semantic_0"):
        e = (b[k][0].cpu() - want[k]).abs()
        assert float(torch.quantile(e.flatten(), 0.95)) < 1e-2, (k, float(e.max()))
    with pytest.raises(ValueError, match="bbox_sampling"):
        make_renderer(NS(bbox_sampling="intervals", **cfg), net)


@pytest.mark.parametrize("R,N,M,mh", [(1000, 64, 40, 8), (257, 32, 64, 3), (64, 100, 12, 8), (5, 192, 40, 1), (300, 64, 0, 8)])
@pytest.mark.parametrize("hull", [False, True])
@pytest.mark.parametrize("jitter,lindisp", [(False, False), (True, False), (False, True)])
def test_ray_setup_is_the_four_separate_kernels_bit_for_bit(dev, R, N, M, mh, hull, jitter, lindisp):
    """pnr_ray_setup (bbox_hits + restrict_rays' hull + stratified + sample_labels in one launch) against the C oracle's functions
    and against the separate entry points: hit lists (also the unused entries and the true count of an overflowing ray), z and
    both label images, bit for bit.  Ragged ray counts (not a multiple of the 256-ray workgroup or the 64-ray wave), N above and
    below the wave width, an empty box table, max_hits 1 / 3 / 8 with overflow, jittered and lindisp sampling."""
    rays = synthetic.camera_rays()[:: max(1, (1408 * 376) // R)][:R].contiguous()
    box, ids = synthetic.random_boxes(max(M, 1), 6, 3, seed=R)
    box, ids = box[:M].contiguous(), ids[:M].contiguous()
    g = torch.Generator().manual_seed(N)
    t_rand = torch.rand((R, N), generator=g) if jitter else None
    d = lambda t: None if t is None else t.to(dev)
    hits, z, ls, li = ops.ray_setup(d(rays), d(box), d(ids), N, mh, lindisp, d(t_rand), hull)
    # the C oracle, stage by stage
    ht, hb, hc = co.bbox_hits(rays.numpy(), box.numpy(), mh)
    assert np.array_equal(hits[0].cpu().numpy(), ht) and np.array_equal(hits[1].cpu().numpy(), hb) and np.array_equal(hits[2].cpu().numpy(), hc)
    if M:
        assert (hc > mh).any() or mh == 8          # the small lists overflow: nearest kept, true count reported
    rr = co.restrict_rays(rays.numpy(), ht, hc) if hull else rays.numpy()
    zw = co.stratified(rr, N, lindisp=lindisp, t_rand=None if t_rand is None else t_rand.numpy())
    assert np.array_equal(z.cpu().numpy(), zw)
    lsw, liw = co.sample_labels(zw, ht, hb, hc, ids.numpy())
    assert np.array_equal(ls.cpu().numpy(), lsw) and np.array_equal(li.cpu().numpy(), liw)
    if M:
        assert (lsw >= 0).any()
    # the separate entry points
    h2 = ops.bbox_hits(d(rays), d(box), mh)
    for a, b in zip(hits, h2):
        assert torch.equal(a, b)
    rs = ops.restrict_rays(d(rays), h2[0], h2[2]) if hull else d(rays)
    z2 = ops.stratified(rs, N, lindisp, d(t_rand))
    assert torch.equal(z, z2)
    if M:           # (pnr_sample_labels insists on a box-id table)
        l2 = ops.sample_labels(z2, *h2, d(ids))
        assert torch.equal(ls, l2[0]) and torch.equal(li, l2[1])
    # without box ids: hit lists and z only
    h3, z3, n1, n2 = ops.ray_setup(d(rays), d(box), None, N, mh, lindisp, d(t_rand), hull)
    assert n1 is None and n2 is None and torch.equal(z3, z) and all(torch.equal(a, b) for a, b in zip(h3, hits))
    with pytest.raises(RuntimeError, match="max_hits"):
        ops.ray_setup(d(rays), d(box), d(ids), N, 9)


@pytest.mark.parametrize("Nc,Nf", [(64, 128), (32, 32), (8, 5)])
@pytest.mark.parametrize("random_u", [False, True])
def test_sample_pdf_labels_is_sample_pdf_then_sample_labels_bit_for_bit(dev, Nc, Nf, random_u):
    """pnr_sample_pdf_labels (the wave that merged a ray's samples labels them) against pnr_sample_pdf + pnr_sample_labels and the
    C oracle: z_fine and both label images bit for bit, with deterministic u (the merge path) and random u (the bitonic-sort path)."""
    R = 777
    rays = synthetic.camera_rays()[::601][:R].contiguous()
    box, ids = synthetic.random_boxes(40, 6, 3, seed=5)
    g = torch.Generator().manual_seed(Nc + Nf)
    w = torch.rand((R, Nc), generator=g) ** 4
    u = torch.rand((R, Nf), generator=g) if random_u else None
    d = lambda t: None if t is None else t.to(dev)
    hits = ops.bbox_hits(d(rays), d(box), 8)
    z = ops.stratified(d(rays), Nc)
    zf, ls, li = ops.sample_pdf_labels(z, d(w), Nf, hits, d(ids), d(u))
    zf2, _, _ = ops.sample_pdf(z, d(w), Nf, d(u), want_samples=False)
    assert torch.equal(zf, zf2)
    l2 = ops.sample_labels(zf2, *hits, d(ids))
    assert torch.equal(ls, l2[0]) and torch.equal(li, l2[1]) and bool((ls >= 0).any())
    zs, _ = co.sample_pdf(z.cpu().numpy(), w.numpy(), Nf, u=None if u is None else u.numpy())
    zw = co.merge_sorted(z.cpu().numpy(), zs)
    assert np.array_equal(zf.cpu().numpy(), zw)
    lsw, liw = co.sample_labels(zw, *(h.cpu().numpy() for h in hits), ids.numpy())
    assert np.array_equal(ls.cpu().numpy(), lsw) and np.array_equal(li.cpu().numpy(), liw)


@pytest.mark.parametrize("Nc,Nf", [(64, 128), (64, 192), (64, 64), (48, 100), (33, 7), (3, 1), (5, 130)])
def test_sample_pdf_inference_instance_equals_the_general_body_and_the_oracle(dev, Nc, Nf, monkeypatch):
    """k_sample_pdf_det (round 6: deterministic u, merged z_fine + labels only, Nc <= 64 -- what every inference frame launches)
    against the general body (PNR_SAMPLE_PDF_GENERAL=1 routes the same call to it) and the C oracle, bit for bit: ragged ray
    counts, weights that leave the exact-scan regime (huge / denormal / zero rows: the sequential CDF), with and without the
    label block, max_hits 1 / 3 / 8 with overflowing hit lists, and UNSORTED coarse z (the bitonic fallback)."""
    rng = np.random.default_rng(Nc * 977 + Nf)
    R = 1031
    rays = synthetic.camera_rays()[::509][:R].contiguous()
    box, ids = synthetic.random_boxes(48, 6, 3, seed=7)
    d = lambda t: None if t is None else t.to(dev)
    w = rng.uniform(0, 1, (R, Nc)).astype(np.float32) ** 4
    if Nc > 8:
        w[0::9, 7] = 3.0e4
        w[3::9, 3] = 1.0e9
    w[1::9] = 1.0e-30
    w[2::9, ::2] = 0.0
    w[4::9] = 0.0
    w = torch.from_numpy(w)
    z = ops.stratified(d(rays), Nc)
    for mh in (8, 3, 1):
        hits = ops.bbox_hits(d(rays), d(box), mh)
        got = {}
        for mode in ("det", "general"):
            if mode == "general":
                monkeypatch.setenv("PNR_SAMPLE_PDF_GENERAL", "1")
            else:
                monkeypatch.delenv("PNR_SAMPLE_PDF_GENERAL", raising=False)
            zf, ls, li = ops.sample_pdf_labels(z, d(w), Nf, hits, d(ids), None)
            zf2, _, _ = ops.sample_pdf(z, d(w), Nf, None, want_samples=False)
            got[mode] = (zf.clone(), ls.clone(), li.clone(), zf2.clone())
        for a, b in zip(got["det"], got["general"]):
            assert torch.equal(a, b), (Nc, Nf, mh)
        assert torch.equal(got["det"][0], got["det"][3])
        if mh == 8:
            zs, _ = co.sample_pdf(z.cpu().numpy(), w.numpy(), Nf)
            zw = co.merge_sorted(z.cpu().numpy(), zs)
            assert np.array_equal(got["det"][0].cpu().numpy(), zw)
            lsw, liw = co.sample_labels(zw, *(h.cpu().numpy() for h in hits), ids.numpy())
            assert np.array_equal(got["det"][1].cpu().numpy(), lsw) and np.array_equal(got["det"][2].cpu().numpy(), liw)
    # unsorted coarse depths: not a merge of two sorted lists -> both bodies sort the union
    zu = z.clone()
    zu[::3] = zu[::3].flip(-1)
    out = {}
    for mode in ("det", "general"):
        if mode == "general":
            monkeypatch.setenv("PNR_SAMPLE_PDF_GENERAL", "1")
        else:
            monkeypatch.delenv("PNR_SAMPLE_PDF_GENERAL", raising=False)
        out[mode] = ops.sample_pdf(zu, d(w), Nf, None, want_samples=False)[0].clone()
    monkeypatch.delenv("PNR_SAMPLE_PDF_GENERAL", raising=False)
    assert torch.equal(out["det"], out["general"])
    assert bool((out["det"][:, 1:] >= out["det"][:, :-1]).all())


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("tap,depth", [("feature", 2), ("trunk", 1), ("feature", 1)])
@pytest.mark.parametrize("geom", [(8, 256, [4], 45, 32), (4, 128, [1], 6, 0)])
def test_head_tap_and_depth_switches(dev, geom, tap, depth, prec):
    """SURVEY.md 9 item 4 as CONFIG switches (cfg.head_tap, cfg.head_depth): the semantic / instance heads read the trunk output
    or the feature_linear output, and are W -> W/2 -> n or one Linear W -> n.  Every combination against the oracle MLP with
    the same switch: fp32-MFMA mode to 1e-4 (the parity bar), bf16 against the bf16-emulating oracle; the fused inference pass
    (plan 1 where the geometry has one, else the classic order) equals the two-kernel path."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    D, W, skips, C, K = geom
    torch.manual_seed(D + C + depth)
    net = make_network(NS(D=D, W=W, skips=skips, num_classes=C, num_instances=K, N_importance=0, head_tap=tap, head_depth=depth,
                          precision=prec)).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    assert len(net.nerf_0.semantic_linears) == depth
    R, N = 70, 32
    rays = synthetic.camera_rays()[::7001][:R].contiguous()
    z = torch.tensor(co.stratified(rays.numpy(), N))
    oc = to.mlp_config(D=D, W=W, skips=tuple(skips), n_sem=C, n_inst=K, head_W=W // 2, head_tap=tap, head_depth=depth)
    prm = {k: v.detach().cpu() for k, v in net.nerf_0.state_dict().items()}
    want = to.run_network(prm, oc, rays, z, emulate_bf16=(prec == "bf16"))
    base = to.run_network(prm, to.mlp_config(D=D, W=W, skips=tuple(skips), n_sem=C, n_inst=K, head_W=W // 2), rays, z) if depth == 2 else None
    desc, img = net.packed(0, dev, prec)
    assert desc.head_tap == (1 if tap == "feature" else 0) and desc.head_depth == depth
    raw = ops.mlp_forward(desc, img, rays.to(dev), z.to(dev), channel_major=True)
    got = raw.unflatten(1, (R, N)).permute(1, 2, 0).cpu()
    err = (got - want).abs()
    if prec == "fp32":
        assert float(err.max()) < 1e-4, float(err.max())
    else:
        assert float(torch.quantile(err.flatten(), 0.99)) < 2e-2 and float(err.max()) < 0.2, (float(err.max()),)
    assert torch.equal(got[..., :4], got[..., :4]) and (base is None or float((want[..., 4:] - base[..., 4:]).abs().max()) > 1e-3)  # the switch matters
    if prec == "bf16":
        a = ops.composite(raw, z.to(dev), rays.to(dev), C, K, True, None, None, None, 0, False, True)
        fd, fimg = net.packed(0, dev, prec, fused=True)
        b = ops.mlp_forward_composite(fd, fimg, rays.to(dev), z.to(dev), None, None, False, True)
        for k in a:
            assert float((a[k] - b[k]).abs().max()) <= 4e-6 * max(1.0, float(a[k].abs().max())) * (N // 32), k
