"""BASELINE.json configs 1..5 rendered END TO END through make_network / make_renderer:
coarse-only 32-sample 4x128, coarse-only 64-sample 8x256, coarse+fine without heads, + semantic head and bbox prior,
full panoptic -- each against (a) the committed per-config fixture tests/golden/configs.npz (an independent fp32
oracle run), (b) the oracle evaluated on the HIP path's own z (identical stage inputs: 1e-4 in fp32-MFMA mode, 1e-2 in
bf16 mode against the bf16-emulating oracle), (c) the strict-order C oracle for z, bit for bit; then at BASELINE's
full-frame size through size-independent properties."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import torch_oracle as to
from panopticnerf_amd import make_network, make_renderer, synthetic

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import config_case  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg_golden():
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "configs.npz")))


def _build(n, prec, dev, **extra):
    c, oc, params, rays, box, ids = config_case(n)
    cfg = synthetic.baseline_cfg(n, precision=prec, **extra)
    net = make_network(cfg).eval()
    net.nerf_0.load_state_dict(params["coarse"])
    if c["N_importance"]:
        net.nerf_1.load_state_dict(params["fine"])
    else:
        assert net.nerf_1 is None
    return c, oc, params, rays, box, ids, cfg, make_renderer(cfg, net.to(dev))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5])
def test_render_baseline_configs(dev, cfg_golden, n, prec):
    c, oc, params, rays, box, ids, cfg, rend = _build(n, prec, dev, chunk_size=16)       # 24 rays: two chunks, one ragged
    batch = {"rays": rays[None].to(dev)}
    if c["bbox"]:
        batch.update(bbox=box.to(dev), bbox_ids=ids.to(dev))
    with torch.no_grad():
        out = rend.render(batch)
    top = 1 if c["N_importance"] else 0
    C, K = c["num_classes"], c["num_instances"]
    # the key set says which parts of the path ran
    assert ("rgb_1" in out) == bool(c["N_importance"]) and ("semantic_0" in out) == bool(C) and ("instance_0" in out) == bool(K)
    assert ("fix_semantic_0" in out) == bool(C and c["bbox"]) and ("fix_instance_0" in out) == bool(K and c["bbox"])
    assert out[f"z_vals_{top}"].shape == (1, 24, c["N_samples"] + c["N_importance"])
    # (c) z: strict-order oracle, bit for bit (z_1 given the HIP coarse weights)
    z0 = out["z_vals_0"][0].cpu().numpy()
    assert np.array_equal(z0, co.stratified(rays.numpy(), c["N_samples"]))
    if top:
        zs, _ = co.sample_pdf(z0, out["weights_0"][0].cpu().numpy(), c["N_importance"])
        assert np.array_equal(out["z_vals_1"][0].cpu().numpy(), co.merge_sorted(z0, zs))
    # (a) committed fixture (independent end-to-end fp32 run of the torch oracle: PE-amplified tolerance, see test_gpu_render)
    if prec == "fp32":
        for k in cfg_golden[f"c{n}_keys"]:
            got, want = out[str(k)][0].cpu().numpy(), cfg_golden[f"c{n}_{k}"]
            if k.startswith("z_vals"):
                # a PE-amplified weight difference can move an importance sample that sits on a CDF bin edge into the
                # neighbouring bin (a whole coarse interval away): all but a handful of the samples must agree
                assert np.mean(np.abs(got - want) > 5e-2) < 2e-3, (n, k, np.abs(got - want).max())
                continue
            tol = 5e-2 if k.startswith("depth") else 1e-2
            np.testing.assert_allclose(got, want, atol=tol, rtol=0, err_msg=f"config {n} {k}")
    # (b) identical stage inputs
    hits = co.bbox_hits(rays.numpy(), box.numpy(), 8) if c["bbox"] else None
    for lv in range(top + 1):
        z = out[f"z_vals_{lv}"][0].cpu()
        raw = to.run_network(params["coarse" if lv == 0 else "fine"], oc, rays, z, emulate_bf16=(prec == "bf16"))
        ls = li = None
        if hits is not None:
            ls, li = (torch.tensor(a) for a in co.sample_labels(z.numpy(), *hits, ids.numpy()))
        want = to.raw2outputs(raw, z, rays[:, 3:6], C, K, None, ls, li)
        # bf16: the last sample's 1e10 interval makes alpha_last a step function of sign(sigma_last); a ray whose
        # sigma_last sits within bf16 noise of zero may flip -- such rays are judged in fp32 mode only
        ok = torch.ones(24, dtype=torch.bool) if prec == "fp32" else raw[:, -1, 3].abs() > 2e-2
        print("config %d %s level %d: %d of 24 rays excluded (|sigma_last| <= 2e-2)" % (n, prec, lv, 24 - int(ok.sum())))
        assert ok.sum() >= 20          # at most 4 of 24: the exclusion must stay the exception it is described as
        tol = 1e-4 if prec == "fp32" else 1e-2
        for k in ("rgb", "acc", "weights", "semantic", "instance", "fix_semantic", "fix_instance"):
            if f"{k}_{lv}" in out:
                err = (out[f"{k}_{lv}"][0].cpu() - want[k])[ok].abs().max().item()
                assert err < tol, (n, prec, k, lv, err)
        derr = (out[f"depth_{lv}"][0].cpu() - want["depth"])[ok].abs().max().item()
        assert derr < (1e-4 if prec == "fp32" else 1e-2) * 100.0, (n, prec, "depth", lv, derr)     # metres, far = 100


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5])
def test_baseline_config_full_frame_properties(dev, n):
    """Every BASELINE config at the size it is quoted on (one 1408x376 frame), bf16, at its own head widths (config 5: the
    benched 45 semantic + 32 instance logits, through the fused pass with the fused-inference chunk order)."""
    c, oc, params, _, box, ids, cfg, rend = _build(n, "bf16", dev)
    rays = synthetic.camera_rays()
    batch = {"rays": rays.reshape(376, 1408, 8).to(dev)}
    if c["bbox"]:
        batch.update(bbox=box.to(dev), bbox_ids=ids.to(dev))
    with torch.no_grad():
        out = rend.render(batch)
    top = 1 if c["N_importance"] else 0
    assert out[f"rgb_{top}"].shape == (376, 1408, 3)
    for k, v in out.items():
        assert torch.isfinite(v).all(), k
    for lv in range(top + 1):
        w = out[f"weights_{lv}"]
        assert (w >= 0).all() and (w.sum(-1) <= 1 + 1e-4).all() and torch.allclose(out[f"acc_{lv}"], w.sum(-1), atol=1e-4)
        z = out[f"z_vals_{lv}"]
        assert (z[..., 1:] >= z[..., :-1]).all() and z.min() >= 0.5 and z.max() <= 100.0
        assert (out[f"rgb_{lv}"] >= 0).all() and (out[f"rgb_{lv}"] <= 1 + 1e-5).all()
    # chunk independence: a ray subset rendered alone gives the same maps, bit for bit
    idx = torch.arange(5, 376 * 1408, 2311)
    sb = {"rays": rays[idx][None].to(dev)}
    if c["bbox"]:
        sb.update(bbox=box.to(dev), bbox_ids=ids.to(dev))
    with torch.no_grad():
        sub = rend.render(sb)
    for k in out:
        a = out[k].reshape(-1, *out[k].shape[2:])[idx.to(dev)]
        assert torch.equal(a, sub[k][0]), (n, k)
    # the subset agrees with the bf16-emulating oracle where it can afford to run
    want = to.render_rays(params, oc, rays[idx], c["N_samples"], c["N_importance"], box=box, box_ids=ids, emulate_bf16=True)
    e = (sub[f"rgb_{top}"][0].cpu() - want[f"rgb_{top}"]).abs()
    assert torch.quantile(e.flatten(), 0.95) < 1e-2, (n, e.max())


FRAME_TOL = {
    # (max, 99.9th percentile) of |hip - oracle| per map over the non-excluded rays of the 4,136-ray subset; depth in metres (far = 100).
    # From the error histograms of the MI355X run committed as profiles/r06/r06c_frame_scale_oracle_errors.txt: ~3-5x the worst figure of
    # the four cases (config 4 / 5 x logits / softmax; measured worst: rgb 1.7e-4 / 1.1e-4, acc 3.2e-4 / 2.1e-4, weights 3.2e-4 / 5.4e-5,
    # depth 1.8e-2 / 5.8e-3 m, learned fields 5.6e-5 / 1.7e-5, fixed fields 2.1e-4 / 1.0e-5).  north_star's 1e-4 is the fp32 bar; this is
    # the bf16 kernel that is benched, against the oracle in ITS arithmetic -- what is left is accumulation order and exp / sigmoid ulps.
    "rgb": (1e-3, 5e-4), "acc": (1.5e-3, 1e-3), "weights": (1.5e-3, 3e-4), "depth": (0.1, 0.03),
    "semantic": (3e-4, 1e-4), "instance": (3e-4, 1e-4), "fix_semantic": (1e-3, 1e-4), "fix_instance": (1e-3, 1e-4),
}


FRAME_MAX_EXCLUDED = 0.10     # share of rays the sigma_last rule may exclude (these are untrained random networks: sigma ~ 0 is common)


@pytest.mark.parametrize("act", ["logits", "softmax"])
@pytest.mark.parametrize("n", [4, 5])
def test_benched_kernel_against_the_oracle_at_frame_scale(dev, n, act):
    """the kernel that is BENCHED (bf16 k_mlp_tt with the fused compositing epilogue; k_mlp_pp plan 1 for softmax
    frames) against the bf16-emulating oracle on a whole 1408x376 frame -- 4,136 rays strided over the frame, EVERY output map
    of both levels (rgb, depth, acc, weights, semantic, instance, fix_*), the oracle evaluated on the HIP path's own z of each
    level (identical stage inputs; z itself is bit-exact against the C oracle in the tests above).  Asserted: max and 99.9th
    percentile per map (FRAME_TOL), argmax agreement of the learned semantic / instance maps >= 99.5 %, and the number of rays
    excluded by the sigma_last rule (bf16 noise flips the sign of a near-zero last density, whose 1e10 interval makes alpha a
    step function) printed and bounded (FRAME_MAX_EXCLUDED).  The error histogram goes to gpurun_out/ for profiles/."""
    extra = {"semantic_activation": "softmax"} if act == "softmax" else {}
    c, oc, params, _, box, ids, cfg, rend = _build(n, "bf16", dev, **extra)
    C, K = c["num_classes"], c["num_instances"]
    rays = synthetic.camera_rays()
    batch = {"rays": rays.reshape(376, 1408, 8).to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    with torch.no_grad():
        out = rend.render(batch)
    idx = torch.arange(3, 376 * 1408, 128)                      # 4,136 rays, every image row and column phase
    assert idx.numel() >= 4096
    sub = rays[idx]
    hits = co.bbox_hits(sub.numpy(), box.numpy(), 8)
    lines = ["config %d, %s compositing, %d rays of the 1408x376 frame (|hip - oracle|; depth in metres)" % (n, act, idx.numel())]
    bad = []
    for lv in (0, 1):
        got = {k[:-2]: v.reshape(-1, *v.shape[2:])[idx.to(dev)].cpu() for k, v in out.items() if k.endswith("_%d" % lv)}
        z = got["z_vals"]
        raw = to.run_network(params["coarse" if lv == 0 else "fine"], oc, sub, z, emulate_bf16=True)
        ls, li = (torch.tensor(a) for a in co.sample_labels(z.numpy(), *hits, ids.numpy()))
        want = to.raw2outputs(raw, z, sub[:, 3:6], C, K, None, ls, li if K else None, sem_mode=1 if act == "softmax" else 0)
        ok = raw[:, -1, 3].abs() > 2e-2
        n_ex = int((~ok).sum())
        lines.append("level %d: %d of %d rays excluded (|sigma_last| <= 2e-2)" % (lv, n_ex, idx.numel()))
        if n_ex > FRAME_MAX_EXCLUDED * idx.numel():
            bad.append(("excluded", lv, n_ex))
        for k, (tmax, t999) in FRAME_TOL.items():
            if k not in got:
                assert want.get(k) is None, (n, act, k, lv)
                continue
            e = (got[k] - want[k])[ok].abs().flatten().double()
            qs = [float(torch.quantile(e, q)) for q in (0.5, 0.95, 0.99, 0.999)]
            lines.append("  %-13s lv %d  p50 %.2e  p95 %.2e  p99 %.2e  p99.9 %.2e  max %.2e" % (k, lv, *qs, float(e.max())))
            if not (float(e.max()) < tmax and qs[3] < t999):
                bad.append((k, lv, float(e.max()), qs[3]))
        for k in ("semantic", "instance"):
            if k in got:
                agree = float((got[k][ok].argmax(-1) == want[k][ok].argmax(-1)).float().mean())
                # rays whose two best classes are within the map tolerance of each other may legitimately swap
                top2 = want[k][ok].topk(2, -1).values
                clear = (top2[:, 0] - top2[:, 1]) > 2 * FRAME_TOL[k][1]
                agree_clear = float((got[k][ok][clear].argmax(-1) == want[k][ok][clear].argmax(-1)).float().mean()) if bool(clear.any()) else 1.0
                lines.append("  %-13s lv %d  argmax agreement %.4f (all rays), %.4f (%d rays with a top-2 margin > %.0e)"
                             % (k, lv, agree, agree_clear, int(clear.sum()), 2 * FRAME_TOL[k][1]))
                if not (agree >= 0.995 and agree_clear == 1.0):
                    bad.append((k + " argmax", lv, agree, agree_clear))
    print("\n".join(lines))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "frame_scale_oracle_errors_c%d_%s.txt" % (n, act)), "w") as f:
        f.write("\n".join(lines) + "\n")
    assert not bad, (n, act, bad)


ODD = {
    # name: (cfg fields, rays, bbox)
    "36 samples (no fused pass), 4x128, semantic only, one ragged tile": (dict(N_samples=36, N_importance=0, D=4, W=128, skips=[], num_classes=7, num_instances=0), 37, True),
    "20 + 12 samples, 130 + 3 logits (above the fused pass's 128)": (dict(N_samples=20, N_importance=12, D=8, W=256, skips=[4], num_classes=130, num_instances=3), 5, True),
    "one ray at the benched geometry": (dict(N_samples=64, N_importance=128, D=8, W=256, skips=[4], num_classes=45, num_instances=32), 1, True),
    "short encodings (6 / 2 bands), skip at layer 1, 3 layers": (dict(N_samples=32, N_importance=32, D=3, W=256, skips=[1], xyz_res=6, view_res=2, num_classes=4, num_instances=2), 33, False),
    "4 samples per ray": (dict(N_samples=4, N_importance=0, D=2, W=128, skips=[], num_classes=0, num_instances=0), 70, False),
}


def test_full_frame_in_one_launch_equals_the_chunked_frame(dev):
    """BASELINE config 5's frame in 8 chunks (the default), in 2, and in ONE (529,408 rays x 192 samples = 101.6 M samples in a single
    fused launch -- the largest launch a frame can ask for: 32-bit sample indices, 397 K groups over 256 workgroups, the two-tile
    kernel's 67 MB per-ray table): rays are independent, so every output map must be the same bits whatever the chunking."""
    ref = None
    for chunk in (65536, 300000, 600000):
        cfg = synthetic.baseline_cfg(5, precision="bf16", chunk_size=chunk)
        torch.manual_seed(0)
        net = make_network(cfg).eval()
        synthetic.trained_like_(net)
        net = net.to(dev)
        rays = synthetic.camera_rays().to(dev)
        box, ids = (t.to(dev) for t in synthetic.random_boxes(64, cfg.num_classes, cfg.num_instances))
        with torch.no_grad():
            out = make_renderer(cfg, net).render({"rays": rays[None], "bbox": box, "bbox_ids": ids})
        torch.cuda.synchronize()
        if ref is None:
            ref = {k: v.cpu() for k, v in out.items()}
            continue
        for k in ref:
            assert torch.equal(ref[k], out[k].cpu()), (chunk, k)
        del out
        torch.cuda.empty_cache()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", list(ODD))
def test_render_odd_geometries_against_the_oracle(dev, name, prec):
    """Geometries off BASELINE's beaten track -- sample counts that are no multiple of 32 (two-kernel path), more logits than the
    fused pass takes, a single ray, ragged tiles, other depths / widths / encoding lengths / skip positions -- through
    make_network / make_renderer, against the oracle on identical stage inputs (1e-4 fp32-MFMA, 1e-2 bf16 vs the bf16-emulating
    oracle) and the C oracle for z, bit for bit."""
    from types import SimpleNamespace as NS
    fields, R, with_box = ODD[name]
    cfg = NS(precision=prec, chunk_size=4096, **fields)
    C, K = fields["num_classes"], fields["num_instances"]
    oc = to.mlp_config(D=fields["D"], W=fields["W"], skips=tuple(fields["skips"]), xyz_L=fields.get("xyz_res", 10), dir_L=fields.get("view_res", 4),
                       n_sem=C, n_inst=K, head_W=fields["W"] // 2)
    params = {"coarse": to.init_params(oc, 11, sigma_bias=0.05), "fine": to.init_params(oc, 12, sigma_bias=0.05)}
    net = make_network(cfg).eval()
    net.nerf_0.load_state_dict(params["coarse"])
    if fields["N_importance"]:
        net.nerf_1.load_state_dict(params["fine"])
    rend = make_renderer(cfg, net.to(dev))
    rays = synthetic.camera_rays()[:: (1408 * 376) // R][:R].contiguous()
    box, ids = synthetic.random_boxes(32, max(C, 1), max(K, 1))
    batch = {"rays": rays[None].to(dev)}
    if with_box:
        batch.update(bbox=box.to(dev), bbox_ids=ids.to(dev))
    with torch.no_grad():
        out = rend.render(batch)
    top = 1 if fields["N_importance"] else 0
    z0 = out["z_vals_0"][0].cpu().numpy()
    assert np.array_equal(z0, co.stratified(rays.numpy(), fields["N_samples"]))
    if top:
        zs, _ = co.sample_pdf(z0, out["weights_0"][0].cpu().numpy(), fields["N_importance"])
        assert np.array_equal(out["z_vals_1"][0].cpu().numpy(), co.merge_sorted(z0, zs))
    hits = co.bbox_hits(rays.numpy(), box.numpy(), 8) if with_box else None
    for lv in range(top + 1):
        z = out[f"z_vals_{lv}"][0].cpu()
        raw = to.run_network(params["coarse" if lv == 0 else "fine"], oc, rays, z, emulate_bf16=(prec == "bf16"))
        ls = li = None
        if hits is not None:
            ls, li = (torch.tensor(a) for a in co.sample_labels(z.numpy(), *hits, ids.numpy()))
        want = to.raw2outputs(raw, z, rays[:, 3:6], C, K, None, ls, li)
        ok = torch.ones(R, dtype=torch.bool) if prec == "fp32" else raw[:, -1, 3].abs() > 2e-2      # see test_render_baseline_configs
        assert ok.sum() >= R // 2
        tol = 1e-4 if prec == "fp32" else 1e-2
        if not ok.any():          # the single ray of the one-ray case may be one of those: it is judged in fp32 mode
            continue
        for k in ("rgb", "acc", "weights", "semantic", "instance", "fix_semantic", "fix_instance"):
            assert (f"{k}_{lv}" in out) == (want.get(k) is not None), (name, k, lv)      # the key set says which parts ran
            if f"{k}_{lv}" in out:
                err = (out[f"{k}_{lv}"][0].cpu() - want[k])[ok].abs().max().item()
                assert err < tol, (name, prec, k, lv, err)
        derr = (out[f"depth_{lv}"][0].cpu() - want["depth"])[ok].abs().max().item()
        assert derr < tol * 100.0, (name, prec, "depth", lv, derr)


def test_unsupported_geometries_are_refused_loudly(dev):
    """What the kernels do not take is an error that names the field -- never a silent other path (the table of
    accepted geometries)."""
    from types import SimpleNamespace as NS
    rays = synthetic.camera_rays()[::9001][:8].contiguous().to(dev)
    for fields, pat in ((dict(N_samples=30), r"n_samples|% 4|multiple"),                 # samples per ray not a multiple of 4
                        (dict(N_samples=64, N_importance=200), r"256|n_samples"),           # 264 samples at the fine level
                        (dict(N_samples=32, W=192), r"W=192"),                             # trunk width
                        (dict(N_samples=32, xyz_res=12), r"xyz_L"),                        # more bands than the encoder stages hold
                        (dict(N_samples=32, D=1), r"D=1")):
        base = dict(N_importance=0, D=8, W=256, skips=[4], num_classes=3, num_instances=0, precision="bf16")
        base.update(fields)
        cfg = NS(**base)
        with pytest.raises(Exception, match=pat):
            with torch.no_grad():
                make_renderer(cfg, make_network(cfg).eval().to(dev)).render({"rays": rays[None]})
