"""GPU end-to-end tests of Renderer.render against the oracle's
render_rays, the golden vectors, and -- at BASELINE's full-frame size -- size-independent
properties (chunk independence, sortedness, weight mass, PSNR parity)."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import torch_oracle as to
from panopticnerf_amd import make_network, make_renderer, ops, synthetic

pytestmark = pytest.mark.gpu


def _setup(dev, C, K, prec, seeds=(4, 5), **kw):
    cfg = NS(N_samples=64, N_importance=128, num_classes=C, num_instances=K, precision=prec, **kw)
    net = make_network(cfg).eval()
    oc = to.mlp_config(n_sem=C, n_inst=K)
    params = {"coarse": to.init_params(oc, seeds[0], sigma_bias=0.05), "fine": to.init_params(oc, seeds[1], sigma_bias=0.05)}
    net.nerf_0.load_state_dict(params["coarse"])
    net.nerf_1.load_state_dict(params["fine"])
    return cfg, net.to(dev), oc, params


def psnr(a, b):
    return -10.0 * torch.log10(torch.mean((a - b) ** 2)).item()


def _oracle_maps_on_hip_z(params, oc, rays, out, lv, box=None, ids=None, max_hits=8, sem_mode=0, white=False):
    """Oracle MLP + raw2outputs evaluated on the HIP path's OWN z_vals of level lv (identical stage
    inputs).  Why not compare two independent end-to-end runs at 1e-4: the L=10 positional encoding
    multiplies a 1-ulp difference in z (strict i/(N-1) vs torch.linspace, sequential vs vectorised CDF)
    by 2^9 * |d|, i.e. ~1e-3 in the highest band, so two correct fp32 pipelines differ by ~1e-3 in
    colour unless they are fed the same z ('hard parts')."""
    z = out[f"z_vals_{lv}"][0].cpu()
    prm = params["coarse" if lv == 0 else "fine"]
    raw = to.run_network(prm, oc, rays, z)
    ls = li = None
    if box is not None:
        hits = co.bbox_hits(rays.numpy(), box.numpy(), max_hits)
        ls, li = co.sample_labels(z.numpy(), *hits, ids.numpy())
        ls, li = torch.tensor(ls), torch.tensor(li)
    return to.raw2outputs(raw, z, rays[:, 3:6], oc.n_sem, oc.n_inst, None, ls, li, sem_mode, white)


def test_render_fp32_matches_golden_and_oracle(dev, golden):
    g = golden
    R, Nc, Nf, C, K, M, MH = (int(v) for v in g["dims"])
    cfg, net, oc, params = _setup(dev, C, K, "fp32", max_hits=MH)
    rend = make_renderer(cfg, net)
    rays = torch.tensor(g["rays"][:32])
    box, ids = torch.tensor(g["box"]), torch.tensor(g["box_ids"])
    with torch.no_grad():
        out = rend.render({"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev)})
    # (1) golden maps: independent end-to-end run of the torch oracle -> PE-amplified tolerance (see above)
    tol = {"depth_0": 2e-2, "depth_1": 5e-2, "z_vals_1": 5e-2}    # z in metres (far = 60); PE-amplified weights shift z_fine
    for k in ("rgb_0", "depth_0", "acc_0", "rgb_1", "depth_1", "acc_1", "semantic_1", "instance_1",
              "fix_semantic_1", "fix_instance_1", "z_vals_1"):
        assert out[k].shape[:2] == (1, 32)
        np.testing.assert_allclose(out[k][0].cpu().numpy(), g["e2e_" + k], atol=tol.get(k, 1e-2), rtol=0, err_msg=k)
    # (2) identical stage inputs: 1e-4 on every map of both levels
    for lv in (0, 1):
        ref = _oracle_maps_on_hip_z(params, oc, rays, out, lv, box, ids, MH)
        for k in ("rgb", "acc", "semantic", "instance", "fix_semantic", "fix_instance", "weights"):
            err = (out[f"{k}_{lv}"][0].cpu() - ref[k]).abs().max().item()
            assert err < 1e-4, (k, lv, err)
        assert (out[f"depth_{lv}"][0].cpu() - ref["depth"]).abs().max() < 6e-3        # metres, far = 60
    # (3) the z the HIP path sampled is the strict-order oracle's z, bit for bit, given the HIP coarse weights
    z0 = out["z_vals_0"][0].cpu().numpy()
    assert np.array_equal(z0, co.stratified(rays.numpy(), Nc))
    zs, _ = co.sample_pdf(z0, out["weights_0"][0].cpu().numpy(), Nf)
    assert np.array_equal(out["z_vals_1"][0].cpu().numpy(), co.merge_sorted(z0, zs))


@pytest.mark.parametrize("perturb", [False, True])
def test_render_fp32_vs_oracle_with_explicit_uniforms(dev, perturb):
    C, K = 5, 3
    cfg, net, oc, params = _setup(dev, C, K, "fp32", seeds=(7, 8), chunk_size=100, white_bkgd=True,
                                  semantic_activation="softmax")
    rays = synthetic.camera_rays()[:: (1408 * 376) // 257][:257].contiguous()      # 3 ragged chunks
    box, ids = synthetic.random_boxes(40, C, K, seed=2)
    gen = torch.Generator().manual_seed(3)
    t_rand = torch.rand(257, 64, generator=gen) if perturb else None
    u = torch.rand(257, 128, generator=gen) if perturb else None
    batch = {"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    if perturb:
        batch["t_rand"], batch["u"] = t_rand[None].to(dev), u[None].to(dev)
    with torch.no_grad():
        out = make_renderer(cfg, net).render(batch)
    # independent end-to-end oracle run: PE-amplified tolerance
    ref = to.render_rays(params, oc, rays, 64, 128, t_rand=t_rand, u=u, box=box, box_ids=ids, sem_mode=1, white_bkgd=True)
    for k in ("rgb_0", "acc_0", "rgb_1", "acc_1", "semantic_1", "instance_1", "fix_semantic_1", "fix_instance_1"):
        err = (out[k][0].cpu() - ref[k]).abs().max().item()
        assert err < 2e-2, (k, err)
    # identical stage inputs: 1e-4
    for lv in (0, 1):
        hyb = _oracle_maps_on_hip_z(params, oc, rays, out, lv, box, ids, 8, sem_mode=1, white=True)
        for k in ("rgb", "acc", "semantic", "instance", "fix_semantic", "fix_instance"):
            err = (out[f"{k}_{lv}"][0].cpu() - hyb[k]).abs().max().item()
            assert err < 1e-4, (k, lv, err)
    z0 = out["z_vals_0"][0].cpu().numpy()
    assert np.array_equal(z0, co.stratified(rays.numpy(), 64, t_rand=None if t_rand is None else t_rand.numpy()))
    zs, _ = co.sample_pdf(z0, out["weights_0"][0].cpu().numpy(), 128, None if u is None else u.numpy())
    assert np.array_equal(out["z_vals_1"][0].cpu().numpy(), co.merge_sorted(z0, zs))


def test_render_bf16_psnr_parity(dev):
    # north_star: PSNR within 0.05 dB of the reference.  No ground-truth images exist here, so the
    # "ground truth" is the fp32 oracle render of a DIFFERENT fine network; the criterion is then
    # |PSNR(hip bf16, gt) - PSNR(oracle fp32, gt)| < 0.05 dB, plus a direct bound on the bf16 error.
    C, K = 6, 4
    cfg, net, oc, params = _setup(dev, C, K, "bf16")
    rays = synthetic.camera_rays()[:: (1408 * 376) // 2048][:2048].contiguous()
    ref = to.render_rays(params, oc, rays, 64, 128)
    gt = to.render_rays({"coarse": params["coarse"], "fine": to.init_params(oc, 99, sigma_bias=0.05)}, oc, rays, 64, 128)
    with torch.no_grad():
        out = make_renderer(cfg, net).render({"rays": rays[None].to(dev)})
    rgb = out["rgb_1"][0].cpu()
    assert abs(psnr(rgb, gt["rgb_1"]) - psnr(ref["rgb_1"], gt["rgb_1"])) < 0.05
    assert psnr(rgb, ref["rgb_1"]) > 45.0
    assert (out["semantic_1"][0].cpu() - ref["semantic_1"]).abs().max() < 3e-2


def test_render_full_frame_properties(dev):
    # BASELINE size: one 1408x376 frame, 64+128 samples, semantic + instance heads, bf16.
    C, K = 19, 8
    cfg, net, oc, params = _setup(dev, C, K, "bf16", chunk_size=65536)
    rays = synthetic.camera_rays()
    box, ids = synthetic.random_boxes(64, C, K)
    rend = make_renderer(cfg, net)
    batch = {"rays": rays.reshape(376, 1408, 8).to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    with torch.no_grad():
        out = rend.render(batch)
    assert out["rgb_1"].shape == (376, 1408, 3) and out["semantic_1"].shape == (376, 1408, C)
    for k, v in out.items():
        assert torch.isfinite(v).all(), k
    z1 = out["z_vals_1"]
    assert (z1[..., 1:] >= z1[..., :-1]).all() and z1.min() >= 0.5 and z1.max() <= 100.0
    for lv in (0, 1):
        w = out[f"weights_{lv}"]
        assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-4).all()
        assert torch.allclose(out[f"acc_{lv}"], w.sum(-1), atol=1e-4)
        assert (out[f"fix_semantic_{lv}"].sum(-1) <= out[f"acc_{lv}"] + 1e-4).all()
    # chunk independence / idempotence: any subset of rays rendered alone gives the same maps
    idx = torch.arange(0, 376 * 1408, 1409)
    sb = {"rays": rays[idx][None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    with torch.no_grad():
        sub = rend.render(sb)
    for k in ("rgb_1", "depth_1", "semantic_1", "instance_1", "fix_semantic_1", "z_vals_1"):
        a = out[k].reshape(-1, *out[k].shape[2:])[idx.to(dev)]
        assert torch.equal(a, sub[k][0]), k
    # the differentiable path (grad enabled: training forward + k_composite, the raw image in HBM) against the fused inference
    # pass: same coarse maps to fp32 rounding (the per-ray sums are associated differently), fine maps to ~1e-3 (the fine
    # samples follow the coarse weights' last bits and gamma() amplifies them)
    trn = rend.render(sb)
    assert torch.equal(trn["z_vals_0"], sub["z_vals_0"])
    for k in ("rgb_0", "depth_0", "semantic_0", "fix_semantic_0", "weights_0"):
        assert float((trn[k] - sub[k]).abs().max()) <= 4e-6 * max(1.0, float(sub[k].abs().max())), k
    assert float(torch.quantile((trn["rgb_1"] - sub["rgb_1"]).abs().flatten(), 0.99)) < 5e-3
    # and the subset agrees with the oracle (bf16 emulation) where it can afford to run
    ref = to.render_rays(params, oc, rays[idx], 64, 128, box=box, box_ids=ids, emulate_bf16=True)
    assert (sub["rgb_1"][0].cpu() - ref["rgb_1"]).abs().max() < 2e-2


def test_render_chunk_is_graph_capturable(dev):
    """include/pnr.h promises every entry point is graph-capture safe (no allocation, no synchronisation, all work on the
    given stream).  Capture one whole render_rays chunk (stratified, bbox hits + labels, coarse MLP, compositing,
    sample_pdf, fine MLP, compositing) into a HIP graph and replay it on NEW ray data written into the captured input
    buffer: the replayed outputs must equal an eager render of the same rays bit for bit."""
    cfg, net, _, _ = _setup(dev, 5, 3, "bf16")
    rend = make_renderer(cfg, net)
    all_rays = synthetic.camera_rays()
    box, ids = synthetic.random_boxes(16, 5, 3)
    box, ids = box.to(dev), ids.to(dev)
    R = 2048
    rays_a = all_rays[::251][:R].contiguous().to(dev)
    rays_b = all_rays[7::233][:R].contiguous().to(dev)
    net.packed(0, dev), net.packed(1, dev)              # the packer copies descriptors from the host: outside the capture
    static_in = rays_a.clone()
    with torch.no_grad():
        rend.render_rays(static_in, box, ids)            # warm-up (lazy module loading is not capturable either)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            static_out = rend.render_rays(static_in, box, ids)
        static_in.copy_(rays_b)
        g.replay()
        torch.cuda.synchronize()
        got = {k: v.clone() for k, v in static_out.items()}
        ref = rend.render_rays(rays_b, box, ids)
    assert set(got) == set(ref)
    for k in ref:
        assert torch.equal(got[k], ref[k]), k
    assert not torch.equal(got["rgb_1"], rend.render_rays(rays_a, box, ids)["rgb_1"])      # it really re-ran on the new rays


def test_reference_checkpoint_loads_packs_and_renders(dev, tmp_path):
    """SURVEY 8f-3 on the GPU: a reference-style checkpoint file ({'net': state_dict with wrapper
    prefixes / alternative names, 'epoch': ...}) -> load_reference_state_dict on a GPU-resident network -> the packed MFMA
    images are rebuilt -> the render equals the source network's render bit for bit (and differs from the render before
    the load: the stale packed image was dropped)."""
    C, K = 6, 4
    cfg, src, _, _ = _setup(dev, C, K, "bf16", seeds=(11, 12))
    ren = lambda k: "module.net." + k.replace("nerf_0.", "coarse.").replace("nerf_1.", "fine.").replace("alpha_linear", "sigma_linear")
    path = tmp_path / "latest.pth"
    torch.save({"net": {ren(k): v.detach().cpu() for k, v in src.state_dict().items()}, "epoch": 7}, path)
    rays = synthetic.camera_rays()[::2311][:200].contiguous().to(dev)
    box, ids = synthetic.random_boxes(16, C, K)
    batch = {"rays": rays[None], "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    dst = make_network(cfg).to(dev).eval()
    rend = make_renderer(cfg, dst)
    with torch.no_grad():
        before = rend.render(batch)                                  # packs dst's random initialisation
        rep = dst.load_reference_state_dict(torch.load(path, map_location="cpu")["net"])
        assert not rep["missing"] and not rep["unexpected"]
        after = rend.render(batch)
        want = make_renderer(cfg, src).render(batch)
    assert not torch.equal(before["rgb_1"], after["rgb_1"])
    for k in want:
        assert torch.equal(after[k], want[k]), k
    # a .data write (no version bump) is picked up after invalidate_packed()
    with torch.no_grad():
        dst.nerf_1.rgb_linear.bias.data.add_(0.5)
        stale = rend.render(batch)
        dst.invalidate_packed()
        fresh = rend.render(batch)
    assert torch.equal(stale["rgb_1"], after["rgb_1"]) and not torch.equal(fresh["rgb_1"], after["rgb_1"])


def test_strict_hits_reports_overflow_once_per_render(dev):
    """cfg.strict_hits: rays that cross more than max_hits boxes raise -- accumulated on the device over the chunks and checked
    with ONE sync at the end of render(), naming the count and the worst case; without the flag the nearest
    max_hits intervals are kept silently."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network, make_renderer
    cfg = dict(N_samples=32, N_importance=0, num_classes=5, num_instances=0, precision="bf16", chunk_size=128, D=2, W=128, skips=[], max_hits=1)
    net = make_network(NS(**cfg)).to(dev).eval()
    rays = synthetic.camera_rays()[::1500][:300].contiguous().to(dev)
    box, ids = synthetic.random_boxes(64, 5, 1)
    batch = {"rays": rays[None], "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    with torch.no_grad():
        out = make_renderer(NS(**cfg), net).render(batch)
        assert out["fix_semantic_0"].shape == (1, 300, 5)
        with pytest.raises(RuntimeError, match=r"cross more than max_hits = 1 boxes \(up to \d+\)"):
            make_renderer(NS(strict_hits=True, **cfg), net).render(batch)
        big = dict(cfg, max_hits=64)
        make_renderer(NS(strict_hits=True, **big), net).render(batch)        # room for every box: no complaint


def test_chunked_frame_is_written_in_place_and_equals_one_chunk(dev):
    # a frame of several chunks (ragged last one) is rendered into frame-sized maps, chunk by chunk, without concatenation:
    # every key equals the one-chunk render bit for bit; ops refuse a caller-owned output of the wrong shape
    C, K = 5, 3
    cfg, net, oc, params = _setup(dev, C, K, "bf16", chunk_size=4096)
    rays = synthetic.camera_rays()[::53][:1000].contiguous()
    box, ids = synthetic.random_boxes(16, C, K)
    b = {"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    with torch.no_grad():
        one = make_renderer(cfg, net).render(b)
        cfg.chunk_size = 384
        many = make_renderer(cfg, net).render(b)
    assert set(one) == set(many)
    for k in one:
        assert one[k].shape == many[k].shape and torch.equal(one[k], many[k]), k
    with pytest.raises(ValueError):
        ops.stratified(rays.to(dev), 64, out=torch.empty((999, 64), device=dev))



@pytest.mark.parametrize("act", ["none", "softmax"])
@pytest.mark.parametrize("heads,with_box", [((45, 32), True), ((45, 0), True), ((0, 0), False)])
def test_overlapped_levels_equal_the_serial_frame(dev, heads, with_box, act):
    """Inference frames of several chunks run the fine level of chunk c beside the coarse level of chunk c + 1 (two streams, the
    two-tile MLP launches on 3/4 and 1/4 of the compute units: Renderer._render_overlapped, PNR_MLP_WG_CAP).  Same kernels on the
    same inputs: every map is the serial frame's (cfg.overlap_levels = False), bit for bit -- five chunks with a ragged last one,
    with and without heads / boxes, logits and softmax compositing, and twice in a row (the side streams are reused)."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    C, K = heads
    if act == "softmax" and not C:
        pytest.skip("no learned field to activate")
    cfg = NS(N_samples=64, N_importance=128, num_classes=C, num_instances=K, precision="bf16", chunk_size=4096, keep_weights=True,
             semantic_activation=act, overlap_levels=True)          # (explicit: PNR_OVERLAP=0 in the environment must not switch the test off)
    torch.manual_seed(5)
    net = make_network(cfg).to(dev).eval()
    synthetic.trained_like_(net, 0.05)
    rays = synthetic.camera_rays()[::29][:18000].contiguous().to(dev)
    b = {"rays": rays[None]}
    if with_box:
        box, ids = synthetic.random_boxes(16, max(C, 1), max(K, 1))
        b.update(bbox=box.to(dev), bbox_ids=ids.to(dev))
    with torch.no_grad():
        r_over = make_renderer(cfg, net)
        assert r_over._overlap_caps(dev, [(0, 1), (1, 2)], False, False, with_box, None, None) == (64, 192)
        over = r_over.render(b)
        over2 = r_over.render(b)
        cfg.overlap_levels = False
        r_ser = make_renderer(cfg, net)
        assert r_ser._overlap_caps(dev, [(0, 1), (1, 2)], False, False, with_box, None, None) is None
        ser = r_ser.render(b)
    torch.cuda.synchronize()
    assert set(over) == set(ser)
    for k in ser:
        assert over[k].shape == ser[k].shape and torch.equal(over[k], ser[k]) and torch.equal(over2[k], ser[k]), k


def test_a_ranks_share_of_the_frame_is_one_chunk(dev):
    # strong scaling over 8 ranks hands every rank 66,176 rays: with the balanced plan (renderer.chunk_plan) that is ONE chunk per
    # level -- not a 65,536-ray chunk plus a 640-ray chunk with its own launches -- and it equals the two-chunk render bit for bit
    from panopticnerf_amd import renderer as R_
    C, K = 5, 3
    cfg, net, oc, params = _setup(dev, C, K, "bf16", chunk_size=65536)
    rays = synthetic.camera_rays()[::8][:66176].contiguous()
    assert rays.shape[0] == 66176
    box, ids = synthetic.random_boxes(16, C, K)
    b = {"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    calls = []
    rend = make_renderer(cfg, net)
    inner = rend.render_rays
    rend.render_rays = lambda r, *a, **k: (calls.append(r.shape[0]), inner(r, *a, **k))[1]
    with torch.no_grad():
        one = rend.render(b)
        assert calls == [66176]
        cfg.chunk_size = 40000
        assert [e - s for s, e in R_.chunk_plan(66176, 40000)] == [33792, 32384]
        two = make_renderer(cfg, net).render(b)
    for k in one:
        assert torch.equal(one[k], two[k]), k


@pytest.mark.parametrize("with_box", [False, True])
def test_render_of_zero_rays_returns_every_key_with_zero_rows(dev, with_box):
    # empty input (a mask that selects nothing, a shard of a frame with fewer rays than ranks): same keys, dtypes and trailing
    # shapes as a non-empty render, zero rows, no kernel launched -- and a one-ray render works (ragged: far below one tile)
    C, K = 5, 3
    cfg, net, oc, params = _setup(dev, C, K, "bf16", chunk_size=4096)
    rays = synthetic.camera_rays()[::53][:7].contiguous().to(dev)
    b = {"rays": rays[None]}
    if with_box:
        box, ids = synthetic.random_boxes(16, C, K)
        b.update(bbox=box.to(dev), bbox_ids=ids.to(dev))
    rend = make_renderer(cfg, net)
    with torch.no_grad():
        some = rend.render(b)
        none = rend.render(dict(b, rays=rays[None, :0]))
        one = rend.render(dict(b, rays=rays[None, :1]))
    assert set(none) == set(some) == set(one)
    for k, v in some.items():
        assert none[k].shape == (1, 0) + tuple(v.shape[2:]) and none[k].dtype == v.dtype and none[k].device == v.device, k
        assert one[k].shape == (1, 1) + tuple(v.shape[2:]), k
        assert torch.equal(one[k][0, 0], v[0, 0]), k                 # rays are independent: ray 0 alone == ray 0 in a batch
    with torch.enable_grad():
        net.train()
        empty = make_renderer(cfg, net).render(dict(b, rays=rays[None, :0]))
        net.eval()
    assert all(v.shape[1] == 0 for v in empty.values())
