"""CPU tests of the oracle itself: the three restatements (strict-order C, torch fp32,
plain-loop numpy float64) against each other, against analytic known answers, and against
the committed golden vectors.  PARITY UNPINNED (no reference source in the mount): this is
how the self-written oracle is kept honest."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import np_oracle as no
from oracle import torch_oracle as to


def _rays(rng, R, near=0.5, far=60.0):
    o = rng.normal(0, 1, (R, 3))
    d = rng.normal(0, 0.3, (R, 3)) + np.array([0, 0, 1.0])
    return np.concatenate([o, d, np.full((R, 1), near), np.full((R, 1), far)], 1).astype(np.float32)


def test_golden_matches_c_oracle(golden):
    g = golden
    R, Nc, Nf, C, K, M, MH = (int(v) for v in g["dims"])
    assert np.array_equal(co.stratified(g["rays"], Nc), g["z_det"])
    assert np.array_equal(co.stratified(g["rays"], Nc, lindisp=True), g["z_lindisp"])
    assert np.array_equal(co.stratified(g["rays"], Nc, t_rand=g["t_rand"]), g["z_perturb"])
    assert np.array_equal(co.points(g["rays"], g["z_perturb"]), g["pts"])
    ht, hb, hc = co.bbox_hits(g["rays"], g["box"], MH)
    assert np.array_equal(hb, g["hit_box"]) and np.array_equal(hc, g["hit_count"]) and np.array_equal(ht, g["hit_t"])
    ls, li = co.sample_labels(g["z_perturb"], ht, hb, hc, g["box_ids"])
    assert np.array_equal(ls, g["label_sem"]) and np.array_equal(li, g["label_inst"])
    for tag, u in (("det", None), ("rand", g["u"])):
        zs, inds = co.sample_pdf(g["z_perturb"], g["comp0_weights"], Nf, u)
        assert np.array_equal(inds, g[f"pdf_{tag}_inds"]) and np.array_equal(zs, g[f"pdf_{tag}_zs"])
        assert np.array_equal(co.merge_sorted(g["z_perturb"], zs), g[f"pdf_{tag}_zfine"])
    np.testing.assert_allclose(co.embed(g["embed_x"], 10), g["embed_L10"], atol=2e-7)


def test_three_restatements_agree_per_ray_stages():
    rng = np.random.default_rng(1)
    R, N, C, K = 24, 32, 4, 3
    rays = _rays(rng, R)
    tr = rng.random((R, N)).astype(np.float32)
    for lind in (False, True):
        zc = co.stratified(rays, N, lind, tr)
        zt = to.stratified(torch.tensor(rays), N, lind, torch.tensor(tr)).numpy()
        zn = no.stratified(rays, N, lind, tr)
        np.testing.assert_allclose(zc, zn, rtol=2e-6)
        np.testing.assert_allclose(zt, zn, rtol=2e-6)
    raw = rng.normal(0, 1, (R, N, 4 + C + K)).astype(np.float32)
    raw[..., 3] *= 0.2
    noise = rng.normal(0, 0.05, (R, N)).astype(np.float32)
    ls, li = rng.integers(-1, C, (R, N)), rng.integers(-1, K, (R, N))
    for sm in (0, 1):
        oc = co.composite(raw, zc, rays, C, K, noise=noise, label_sem=ls, label_inst=li, sem_mode=sm, white_bkgd=True)
        on = no.composite(raw, zc, rays, C, K, noise=noise, label_sem=ls, label_inst=li, sem_mode=sm, white_bkgd=True)
        ot = to.raw2outputs(torch.tensor(raw), torch.tensor(zc), torch.tensor(rays[:, 3:6]), C, K,
                            torch.tensor(noise), torch.tensor(ls), torch.tensor(li), sm, True)
        for k in oc:
            np.testing.assert_allclose(oc[k], on[k], atol=2e-5, rtol=1e-5, err_msg=k)
            np.testing.assert_allclose(ot[k].numpy(), on[k], atol=2e-5, rtol=1e-5, err_msg=k)
    # channel-major raw image == sample-major
    rawc = np.ascontiguousarray(raw.reshape(R * N, -1).T)
    oc2 = co.composite(rawc, zc, rays, C, K, channel_major=True)
    oc1 = co.composite(raw, zc, rays, C, K)
    for k in ("rgb", "semantic", "instance", "weights"):
        assert np.array_equal(oc1[k], oc2[k])


def test_sample_pdf_restatements():
    rng = np.random.default_rng(2)
    R, Nc, Nf = 40, 64, 128
    rays = _rays(rng, R)
    z = co.stratified(rays, Nc)
    # peaked but nowhere-flat weights: away from the denom<1e-5 discontinuity all three agree
    w = (np.exp(-0.5 * ((np.arange(Nc)[None] - rng.uniform(10, 50, (R, 1))) / 6.0) ** 2) + 0.05).astype(np.float32)
    u = rng.random((R, Nf)).astype(np.float32)
    for uu in (None, u):
        zs_c, i_c = co.sample_pdf(z, w, Nf, uu)
        zs_n, i_n = no.sample_pdf(z, w, Nf, uu)
        zf_t, zs_t, i_t = to.importance_z(torch.tensor(z), torch.tensor(w), Nf, None if uu is None else torch.tensor(uu))
        # the float64 plain-loop restatement rounds differently at bin edges (the det grid's last u == 1.0 sits exactly on
        # cdf[-1] ~ 1 +- 1 ulp): > 0.99 there.  Against torch AS WRITTEN the C oracle is exact: every index, every bit.
        assert (i_c == i_n).mean() > 0.99
        assert np.array_equal(i_c, i_t.numpy()) and np.array_equal(zs_c, zs_t.numpy())
        np.testing.assert_allclose(zs_c, zs_n, atol=2e-3)
        np.testing.assert_allclose(zs_c, zs_t.numpy(), atol=2e-3)
        zf_c = co.merge_sorted(z, zs_c)
        assert np.all(np.diff(zf_c, axis=1) >= 0)
        np.testing.assert_allclose(zf_c, zf_t.numpy(), atol=2e-3)


@pytest.mark.parametrize("Nc,Nf", [(64, 128), (32, 64), (64, 64), (128, 64), (16, 32)])
def test_c_oracle_is_torch_as_written(Nc, Nf):
    """BASELINE north_star: "bit-exact sample indices".  The reference's sampler is torch ops (linspace, sum, cumsum,
    searchsorted, sort); oracle/pnr_oracle.c restates the op ORDER of this container's torch CPU kernels -- two-sided linspace
    with a fused upper half, ATen's 8-lane / 4-way interleaved sum, cumsum accumulated in double -- and must reproduce torch
    bit for bit: stratified z (det, perturbed, lindisp), every sample index, z_samples and the sorted union, on peaked, flat,
    all-zero and noisy weights, with the deterministic grid and with given uniforms.  (The HIP kernels are pinned to the C oracle
    bit for bit by tests/test_gpu_stages.py, and tests/golden/path_small.npz holds torch's own outputs.)"""
    rng = np.random.default_rng(Nc * 1000 + Nf)
    R = 600
    rays = np.zeros((R, 8), np.float32)
    rays[:, 3:6] = rng.normal(size=(R, 3))
    rays[:, 6] = rng.uniform(0.05, 2.0, R)
    rays[:, 7] = rng.uniform(20.0, 120.0, R)
    tr = torch.tensor(rays)
    t_rand = rng.random((R, Nc)).astype(np.float32)
    z = co.stratified(rays, Nc)
    assert np.array_equal(z, to.stratified(tr, Nc).numpy())
    assert np.array_equal(co.stratified(rays, Nc, True), to.stratified(tr, Nc, True).numpy())
    zp = co.stratified(rays, Nc, False, t_rand)
    assert np.array_equal(zp, to.stratified(tr, Nc, False, torch.tensor(t_rand)).numpy())
    w = (np.exp(-0.5 * ((np.arange(Nc)[None] - rng.uniform(2, Nc - 2, (R, 1))) / rng.uniform(0.5, 8, (R, 1))) ** 2)
         * rng.uniform(0, 1, (R, 1)) + rng.random((R, Nc)) * 1e-3).astype(np.float32)
    w[:40] = 0.0                      # no density at all: the 1e-5 floor alone
    w[40:80] = 1.0                    # flat
    for zz in (z, zp):
        for uu in (None, rng.random((R, Nf)).astype(np.float32)):
            zs, inds = co.sample_pdf(zz, w, Nf, uu)
            zf_t, zs_t, i_t = to.importance_z(torch.tensor(zz), torch.tensor(w), Nf, None if uu is None else torch.tensor(uu))
            assert np.array_equal(inds, i_t.numpy()), float((inds == i_t.numpy()).mean())
            assert np.array_equal(zs, zs_t.numpy())
            assert np.array_equal(co.merge_sorted(zz, zs), zf_t.numpy())


ATEN_CAPABILITY_MEASURED = "AVX512"      # torch.backends.cpu.get_cpu_capability() of the build the op orders were measured on


def test_aten_capability_the_op_orders_were_measured_on():
    """"Bit-exact against torch as written" is a statement about ONE ATen build: torch 2.10 CPU kernels dispatched at the
    capability below.  ATen picks its sum / cumsum / linspace kernels per CPU capability (DEFAULT / AVX2 / AVX512 builds of the
    same source vectorise differently: the lane count and the interleave of `sum`'s partial vectors are what pnro_torch_sum
    restates), so on a host where torch dispatches differently the C oracle still IS the specification the HIP kernels are
    pinned to, but its equality with torch's own ops has only been measured here.  test_torch_op_orders_the_c_oracle_restates
    checks the three facts directly and fails first if a torch upgrade or another capability changes one of them.
    (ATEN_CPU_CAPABILITY=avx2 / default in the environment selects another dispatch on the same machine.)"""
    cap = torch.backends.cpu.get_cpu_capability()
    assert cap == ATEN_CAPABILITY_MEASURED, (
        "torch dispatches its CPU kernels at %r here; oracle/pnr_oracle.c restates the op order measured at %r (torch %s). The "
        "oracle remains the HIP kernels' specification, but re-run test_torch_op_orders_the_c_oracle_restates / "
        "test_c_oracle_is_torch_as_written to see whether 'torch as written' still holds on this host." % (cap, ATEN_CAPABILITY_MEASURED, torch.__version__))


def test_torch_op_orders_the_c_oracle_restates():
    """The three measured facts about torch's CPU kernels that pnro_linspace01 / pnro_torch_sum / the double cumsum encode,
    checked directly (if a torch upgrade changes one of them this fails before any index does)."""
    import ctypes
    lib = co.lib()
    lib.pnro_linspace01_at.restype = ctypes.c_float
    lib.pnro_linspace01_at.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.pnro_torch_sum_row.restype = ctypes.c_float
    lib.pnro_torch_sum_row.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    for N in (7, 32, 63, 64, 100, 128, 192, 256):
        t = torch.linspace(0.0, 1.0, steps=N).numpy()
        mine = np.array([lib.pnro_linspace01_at(i, N) for i in range(N)], np.float32)
        assert np.array_equal(t, mine), N
    # ... which is NOT i / (N - 1): the round-2 oracle's form differs from torch in up to half of the values by one ulp
    assert (torch.linspace(0.0, 1.0, steps=64).numpy() != (np.arange(64, dtype=np.float32) / np.float32(63))).sum() > 10
    rng = np.random.default_rng(0)
    for n in (30, 62, 64, 126, 200):
        x = rng.random((200, n)).astype(np.float32)
        s = torch.sum(torch.tensor(x), -1).numpy()
        mine = np.array([lib.pnro_torch_sum_row(np.ascontiguousarray(r).ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n) for r in x], np.float32)
        assert np.array_equal(s, mine), n
        seq = np.zeros(200, np.float32)
        for j in range(n):
            seq = (seq + x[:, j]).astype(np.float32)
        assert (s != seq).mean() > 0.2            # ... and not a sequential sum
        cs = torch.cumsum(torch.tensor(x), -1).numpy()
        assert np.array_equal(cs, np.cumsum(x.astype(np.float64), axis=1).astype(np.float32))       # double running sum


def test_sample_pdf_uniform_weights_gives_even_samples():
    # KAT: uniform weights => det sample_pdf returns evenly spaced z over the bin range
    Nc, Nf = 64, 128
    rays = np.array([[0, 0, 0, 0, 0, 1, 2.0, 6.0]], np.float32)
    z = co.stratified(rays, Nc)
    w = np.ones((1, Nc), np.float32)
    zs, inds = co.sample_pdf(z, w, Nf)
    bins = 0.5 * (z[0, 1:] + z[0, :-1])
    expect = bins[0] + (bins[-1] - bins[0]) * np.arange(Nf) / (Nf - 1)
    np.testing.assert_allclose(zs[0], expect, atol=2e-5)
    assert inds.min() >= 1 and inds.max() <= Nc - 1


def test_composite_known_answers():
    # constant-density slab: acc = 1 - exp(-sigma * L) (last interval is 1e10 => acc -> 1)
    N = 64
    rays = np.array([[0, 0, 0, 0, 0, 2.0, 1.0, 5.0]], np.float32)   # ||d|| = 2
    z = co.stratified(rays, N)
    raw = np.zeros((1, N, 4), np.float32)
    raw[..., 3] = 0.3
    out = co.composite(raw, z, rays, 0, 0)
    w = out["weights"][0]
    L = (z[0, -1] - z[0, 0]) * 2.0
    np.testing.assert_allclose(w[:-1].sum(), 1 - np.exp(-0.3 * L), rtol=1e-5)
    np.testing.assert_allclose(out["acc"][0], 1.0, atol=1e-6)
    np.testing.assert_allclose(out["rgb"][0], 0.5 * out["acc"][0], atol=1e-6)   # sigmoid(0) = .5
    # single opaque sample => one-hot weights
    raw[..., 3] = 0.0
    raw[0, 10, 3] = 1e4
    out = co.composite(raw, z, rays, 0, 0)
    assert out["weights"][0].argmax() == 10 and abs(out["weights"][0, 10] - 1) < 1e-6
    np.testing.assert_allclose(out["depth"][0], z[0, 10], rtol=1e-6)
    # negative density is clamped (relu): nothing accumulates
    raw[..., 3] = -5.0
    assert co.composite(raw, z, rays, 0, 0)["acc"][0] == 0.0


def test_embedder_known_answers():
    e = co.embed(np.zeros((1, 3), np.float32), 10)
    assert e.shape == (1, 63)
    expect = np.concatenate([[0, 0, 0]] + [[0, 0, 0, 1, 1, 1]] * 10)
    assert np.array_equal(e[0], expect.astype(np.float32))
    x = np.array([[0.5, -1.25, 3.0]], np.float32)
    e = co.embed(x, 4)
    np.testing.assert_allclose(e[0, 3:6], np.sin(x[0]), atol=1e-7)
    np.testing.assert_allclose(e[0, 6:9], np.cos(x[0]), atol=1e-7)
    np.testing.assert_allclose(e[0, 3 + 18:3 + 21], np.sin(8 * x[0]), atol=1e-6)
    np.testing.assert_allclose(no.embed(x, 10), to.embed(torch.tensor(x), 10).numpy(), atol=1e-6)


def test_bbox_restatements_and_edge_cases():
    rng = np.random.default_rng(3)
    R, M = 64, 12
    rays = _rays(rng, R, 0.5, 40.0)
    box = np.zeros((M, 15), np.float32)
    box[:, 0:3] = rng.uniform([-3, -2, 3], [3, 2, 30], (M, 3))
    for m in range(M):
        y = rng.uniform(0, np.pi)
        box[m, 3:12] = np.array([[np.cos(y), 0, np.sin(y)], [0, 1, 0], [-np.sin(y), 0, np.cos(y)]]).reshape(-1)
    box[:, 12:15] = rng.uniform(0.5, 3, (M, 3))
    ids = np.stack([rng.integers(0, 5, M), rng.integers(0, 4, M)], 1).astype(np.int32)
    hc = co.bbox_hits(rays, box, 4)
    hn = no.bbox_hits(rays, box, 4)
    ht = to.bbox_hits(torch.tensor(rays), torch.tensor(box), 4)
    assert np.array_equal(hc[1], hn[1]) and np.array_equal(hc[1], ht[1].numpy())
    assert np.array_equal(hc[2], hn[2]) and np.array_equal(hc[2], ht[2].numpy())
    np.testing.assert_allclose(hc[0], hn[0], atol=1e-4)
    assert np.array_equal(hc[0], ht[0].numpy())
    z = co.stratified(rays, 32)
    lc = co.sample_labels(z, *hc, ids)
    lt = to.sample_labels(torch.tensor(z), *ht, torch.tensor(ids))
    assert np.array_equal(lc[0], lt[0].numpy()) and np.array_equal(lc[1], lt[1].numpy())
    # axis-parallel ray (d component exactly 0): inside the slab -> hit, outside -> miss, no NaN leak
    b = np.zeros((1, 15), np.float32)
    b[0, 0:3] = (0, 0, 10)
    b[0, 3:12] = np.eye(3).reshape(-1)
    b[0, 12:15] = (1, 1, 1)
    r_in = np.array([[0.5, 0.5, 0, 0, 0, 1, 0.1, 50]], np.float32)
    r_out = np.array([[1.5, 0.5, 0, 0, 0, 1, 0.1, 50]], np.float32)
    r_edge = np.array([[1.0, 0.0, 0, 0, 0, 1, 0.1, 50]], np.float32)     # grazing: (e - o) = 0, 0*inf = NaN
    assert co.bbox_hits(r_in, b, 2)[2][0] == 1 and co.bbox_hits(r_out, b, 2)[2][0] == 0
    t, bi, cnt = co.bbox_hits(r_in, b, 2)
    np.testing.assert_allclose(t[0, 0], (9.0, 11.0))
    assert co.bbox_hits(r_edge, b, 2)[2][0] == no.bbox_hits(r_edge, b, 2)[2][0]
    # empty box table and max_hits clipping
    assert co.bbox_hits(rays, np.zeros((0, 15), np.float32), 3)[2].sum() == 0
    many = np.repeat(b, 5, 0)
    t, bi, cnt = co.bbox_hits(r_in, many, 3)
    assert cnt[0] == 5 and list(bi[0]) == [0, 1, 2]          # true count reports the overflow; ties keep the lower index
    # more boxes than max_hits along the ray: the NEAREST max_hits survive, in ascending t_in, whatever the table order
    line = np.repeat(b, 6, 0)
    line[:, 2] = (50, 10, 40, 20, 30, 45)                     # centres along z; table order is not depth order
    for fn in (co.bbox_hits, no.bbox_hits, lambda r, bb, mh: tuple(x.numpy() for x in to.bbox_hits(torch.tensor(r), torch.tensor(bb), mh))):
        t, bi, cnt = fn(r_in.copy(), line, 3)
        assert cnt[0] == 6 and list(bi[0]) == [1, 3, 4], (bi, cnt)
        np.testing.assert_allclose(t[0, :, 0], (9.0, 19.0, 29.0))
    ls, _ = co.sample_labels(np.array([[10.0, 20.0, 30.0, 40.0]], np.float32), *co.bbox_hits(r_in, line, 3),
                             np.stack([np.arange(6), np.arange(6)], 1).astype(np.int32))
    assert list(ls[0]) == [1, 3, 4, -1]                        # the dropped (far) box labels nothing; near samples are right


def test_mlp_oracle_bf16_emulation_close_to_fp32():
    cfg = to.mlp_config(n_sem=5, n_inst=4)
    p = to.init_params(cfg, seed=0)
    g = torch.Generator().manual_seed(1)
    pts = torch.rand(64, 3, generator=g) * 10 - 5
    vd = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1)
    a = to.mlp_forward(p, cfg, pts, vd)
    b = to.mlp_forward(p, cfg, pts, vd, emulate_bf16=True)
    assert a.shape == (64, 13)
    assert (a - b).abs().max() < 0.05 and (a - b).abs().max() > 0      # differs, but only by bf16 rounding
    # layer count / skip wiring: parameter count of the 8x256 trunk (595,844 without heads)
    cfg0 = to.mlp_config()
    assert sum(v.numel() for v in to.init_params(cfg0).values()) == 595844


def test_linear_c_vs_torch():
    rng = np.random.default_rng(5)
    x, W, b = rng.normal(size=(7, 33)), rng.normal(size=(9, 33)), rng.normal(size=9)
    y = co.linear(x, W, b, relu=True)
    np.testing.assert_allclose(y, np.maximum(x @ W.T + b, 0), atol=1e-4)
    yb = co.linear(x, W, b, emulate_bf16=True)
    xt, Wt = torch.tensor(x, dtype=torch.float32), torch.tensor(W, dtype=torch.float32)
    ref = to.bf16_round(xt) @ to.bf16_round(Wt).T + torch.tensor(b, dtype=torch.float32)
    np.testing.assert_allclose(yb, ref.numpy(), atol=1e-4)
    assert np.array_equal(co.bf16_round(np.float32([1.0, 1.00390625, 1.01171875])),
                          np.float32([1.0, 1.0, 1.015625]))   # RNE: tie to even, then round up


@pytest.mark.parametrize("N", [1, 2, 64])
def test_stratified_endpoints(N):
    rays = np.array([[0, 0, 0, 0, 0, 1, 2.0, 6.0]], np.float32)
    z = co.stratified(rays, N)
    assert z[0, 0] == 2.0 and (N == 1 or z[0, -1] == 6.0)


def test_gen_rays_c_oracle_matches_torch_oracle_and_synthetic():
    """the two restatements of the ray generator agree bit for bit, reproduce the synthetic KITTI-360-shaped
    camera the benches use, and honour pixel subsets and a rotated / translated pose."""
    import math
    from panopticnerf_amd import synthetic as syn
    intr = [syn.KITTI_F, syn.KITTI_F, syn.KITTI_CX, syn.KITTI_CY]
    c2w0 = np.array([[1, 0, 0, 0.0], [0, 1, 0, 1.55], [0, 0, 1, 0.0]], np.float32)
    full = co.gen_rays(intr, c2w0, syn.KITTI_W, syn.KITTI_H, 0.5, 100.0)
    assert np.array_equal(full, to.gen_rays(intr, c2w0, syn.KITTI_W, syn.KITTI_H, 0.5, 100.0).numpy())
    assert np.array_equal(full, syn.camera_rays().numpy())
    yaw = 0.3
    c, s = math.cos(yaw), math.sin(yaw)
    c2w = np.array([[c, 0, s, 3.0], [0, 1, 0, 1.5], [-s, 0, c, -7.0]], np.float32)
    pix = np.array([0, 1407, 1408, 376 * 1408 - 1, 12345], np.int32)
    sub = co.gen_rays([500.0, 510.0, 700.0, 200.0], c2w, 1408, 376, 1.0, 50.0, pix)
    assert np.array_equal(sub, to.gen_rays([500.0, 510.0, 700.0, 200.0], c2w, 1408, 376, 1.0, 50.0, pix).numpy())
    assert np.allclose(sub[:, :3], [3.0, 1.5, -7.0]) and np.all(sub[:, 6] == 1.0) and np.all(sub[:, 7] == 50.0)
    # pixel (i=1407, j=0): x = (1407-700)/500, y = (0-200)/510 rotated by the yaw
    x, y = (1407 - 700.0) / 500.0, (0 - 200.0) / 510.0
    assert np.allclose(sub[1, 3:6], [c * x + s, y, -s * x + c], atol=1e-6)


def test_loss_oracle_known_answers():
    """SURVEY 8f-1 oracle (torch_oracle.losses / ce3d) against closed forms, so the GPU tests' checker is itself pinned."""
    import math
    R, C = 4, 3
    maps = {"rgb": torch.zeros(R, 3), "depth": torch.tensor([1.0, 2.0, 3.0, 4.0]),
            "semantic": torch.zeros(R, C),                                   # uniform logits: CE = log C
            "fix_semantic": torch.tensor([[0.5, 0.0, 0.0], [0.0, 0.25, 0.0], [0.0, 0.0, 1.0], [0.2, 0.2, 0.2]])}
    tg = {"rgb": torch.full((R, 3), 0.5), "depth": torch.tensor([2.0, 0.0, 1.0, -1.0]),          # two valid depths
          "semantic": torch.tensor([0, 1, 2, -1], dtype=torch.int32)}
    w = {"rgb": 2.0, "depth": 0.5, "semantic": 1.0, "fix_semantic": 3.0}
    terms, total = to.losses(maps, tg, w, n_sem=C, fix_eps=0.0)
    assert abs(terms["rgb"].item() - 0.25) < 1e-7
    assert abs(terms["depth"].item() - (1.0 + 2.0) / 2) < 1e-7                                    # |1-2|, |3-1| over 2 valid rays
    assert abs(terms["semantic"].item() - math.log(3)) < 1e-6
    assert abs(terms["fix_semantic"].item() - (-(math.log(0.5) + math.log(0.25) + math.log(1.0)) / 3)) < 1e-6
    assert abs(total.item() - (2 * 0.25 + 0.5 * 1.5 + math.log(3) + 3 * (math.log(2) + math.log(4)) / 3)) < 1e-5
    terms2, _ = to.losses(maps, tg, w, n_sem=C, depth_l2=True)
    assert abs(terms2["depth"].item() - (1.0 + 4.0) / 2) < 1e-7
    ce, n = to.ce3d(torch.tensor([[0.0, 0.0], [10.0, -10.0], [3.0, 3.0]]), torch.tensor([1, 0, -1]))
    assert n == 2 and abs(ce.item() - (math.log(2) + math.log1p(math.exp(-20.0))) / 2) < 1e-6
    assert to.ce3d(torch.zeros(2, 2), torch.tensor([-1, -1]))[1] == 0


def test_postprocessing_oracle_known_answers():
    """SURVEY 8f-4 oracle (np_oracle.panoptic_labels / confusion): ties, stuff vs thing, ignore labels."""
    sem = np.array([[0.1, 0.9, 0.9], [2.0, 1.0, 0.0], [0.0, 0.0, 0.0]], np.float32)
    inst = np.array([[0.0, 5.0], [7.0, 1.0], [1.0, 1.0]], np.float32)
    sl, il, pan = no.panoptic_labels(sem, inst, is_thing=np.array([0, 1, 1]))
    assert sl.tolist() == [1, 0, 0] and il.tolist() == [1, -1, -1] and pan.tolist() == [1001, 0, 0]
    conf = no.confusion(np.array([0, 1, 1, 2, 5]), np.array([0, 1, 2, -1, 1]), 3)
    assert conf.tolist() == [[1, 0, 0], [0, 1, 0], [0, 1, 0]]


def test_panoptic_quality_oracle_known_answer():
    """One thing class (1) with two instances and one stuff class (0) on a 1 x 10 strip with one ignored pixel: first a
    prediction that matches both instances (IoU 3/4 and 2/3), then one that misses the second (IoU 1/3 -> an FN and an FP)."""
    gt = np.array([0, 0, 0, 1000, 1000, 1000, 1001, 1001, 1001, -1])
    pr = np.array([0, 0, 0, 1000, 1000, 1000, 1000, 1002, 1002, 1002])
    t = no.panoptic_quality_terms(pr, gt, 2)
    # class 0: IoU 1, TP 1.  class 1: gt 1000 (3 px) vs pred 1000 (4 px): inter 3, union 4 -> TP with IoU .75;
    # gt 1001 (3 px) vs pred 1002 (2 valid px): inter 2, union 3 -> IoU 2/3 > .5 -> TP as well
    assert np.allclose(t[0], [1.0, 1, 0, 0])
    assert np.allclose(t[1], [0.75 + 2 / 3, 2, 0, 0])
    pr2 = np.array([0, 0, 0, 1000, 1000, 1000, 1000, 1000, 1002, 1002])
    t2 = no.panoptic_quality_terms(pr2, gt, 2)
    # gt 1000 vs pred 1000 (5 px): inter 3, union 5 -> .6 TP; gt 1001 vs pred 1002 (1 valid px): inter 1, union 3 -> FN, FP
    assert np.allclose(t2[1], [0.6, 1, 1, 1])


@pytest.mark.parametrize("N,tile", [(32, 32), (64, 32), (96, 32), (64, 8), (24, 4)])
def test_tile_factorisation_equals_the_scan(N, tile):
    """The fused inference pass (csrc/pnr_mlp_fuse.h) composites per 32-sample tile and finishes the ray from the tile records.
    The numpy restatement of that factorisation must reproduce the plain scan -- every map, the per-sample weights and the
    fixed fields, also with opaque samples (alpha = 1 up to the 1e-10), empty space (sigma <= 0) and a white background."""
    rng = np.random.default_rng(N * 100 + tile)
    R, C, K = 7, 5, 3
    raw = rng.normal(0, 1.5, (R, N, 4 + C + K))
    raw[:, :, 3] = rng.normal(0.2, 1.0, (R, N))
    raw[0, :, 3] = -1.0                      # an empty ray
    raw[1, N // 3, 3] = 1e4                  # an opaque sample: everything behind it carries weight ~1e-10
    rays = np.concatenate([rng.normal(0, 1, (R, 3)), rng.normal(0, 1, (R, 3)), np.full((R, 1), 0.5), np.full((R, 1), 30.0)], 1)
    z = no.stratified(rays, N, t_rand=rng.random((R, N)))
    ls = np.where(rng.random((R, N)) < 0.4, rng.integers(0, C, (R, N)), -1)
    li = np.where(ls >= 0, rng.integers(0, K, (R, N)), -1)
    for white in (False, True):
        a = no.composite(raw, z, rays, C, K, None, ls, li, 0, white)
        b = no.composite_by_tiles(raw, z, rays, C, K, tile, ls, li, white)
        for k in a:
            np.testing.assert_allclose(b[k], a[k], rtol=1e-12, atol=1e-13, err_msg=k)


def test_restrict_rays_restatements_agree():
    """cfg.bbox_sampling = 'hull': C and torch restatements give the same near / far, rays without a hit are untouched, the
    hull lies inside [near, far], and the samples drawn from it all lie inside the ray's hull."""
    rng = np.random.default_rng(9)
    R, M = 300, 20
    rays = _rays(rng, R)
    box = np.concatenate([rng.uniform([-6, -2, 2], [6, 2, 40], (M, 3)), np.tile(np.eye(3, dtype=np.float32).reshape(-1), (M, 1)),
                          rng.uniform(0.5, 3.0, (M, 3))], 1).astype(np.float32)
    ht, hb, hc = co.bbox_hits(rays, box, 4)
    rc = co.restrict_rays(rays, ht, hc)
    rt = to.restrict_rays(torch.tensor(rays), torch.tensor(ht), torch.tensor(hc)).numpy()
    assert np.array_equal(rc, rt)
    none = hc == 0
    assert none.any() and (~none).any() and np.array_equal(rc[none], rays[none]) and np.array_equal(rc[:, :6], rays[:, :6])
    assert (rc[:, 6] >= rays[:, 6]).all() and (rc[:, 7] <= rays[:, 7]).all() and (rc[~none, 6] <= rc[~none, 7]).all()
    z = co.stratified(rc, 16)
    assert (z >= rc[:, 6:7] - 1e-6).all() and (z <= rc[:, 7:8] + 1e-6).all()
