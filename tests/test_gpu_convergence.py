"""Convergence / PSNR-parity of TRAINING through the HIP path (north_star: "PSNR within 0.05 dB of
reference").  A teacher network renders a synthetic scene (the oracle, fp32, CPU); two students start from the same
initialisation and see the same ray batches:
  * HIP student    -- Renderer.render under autograd (bf16 MFMA forward / dgrad / wgrad kernels, HIP compositing backward);
  * oracle student -- torch autograd through oracle/torch_oracle.render_rays (fp32, CPU).
The HIP loss must fall, and the two students' held-out PSNR against the teacher must agree within 0.05 dB."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import torch_oracle as to
from panopticnerf_amd import make_network, make_renderer

pytestmark = pytest.mark.gpu

C = 4
GEOM = dict(D=4, W=128, skips=(1,))
NC, NF = 32, 32
STEPS, BATCH, LR = 200, 256, 1e-3


def _scene_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = rng.normal(0, 0.05, (n, 3)) + np.array([0.0, 0.0, -3.0])
    d = rng.normal(0, 0.25, (n, 3)) + np.array([0.0, 0.0, 1.0])
    return torch.tensor(np.concatenate([o, d, np.full((n, 1), 1.0), np.full((n, 1), 5.0)], 1).astype(np.float32))


def _loss(out, tgt):
    l = 0
    for lv in (0, 1):
        l = l + ((out[f"rgb_{lv}"] - tgt["rgb"]) ** 2).mean() + 0.1 * ((out[f"depth_{lv}"] - tgt["depth"]) ** 2).mean() \
            + 0.1 * ((out[f"semantic_{lv}"] - tgt["semantic"]) ** 2).mean()
    return l


def _psnr(a, b):
    return -10.0 * torch.log10(torch.mean((a - b) ** 2)).item()


def test_training_converges_like_the_oracle(dev):
    oc = to.mlp_config(n_sem=C, head_W=GEOM["W"] // 2, **GEOM)
    teacher = {"coarse": to.init_params(oc, 31, sigma_bias=0.4), "fine": to.init_params(oc, 32, sigma_bias=0.4)}
    for p in teacher.values():                       # a teacher with structure: high-frequency first layer (the gamma(x)
        p["pts_linears.0.weight"] *= 25.0            # bands dominate), larger colour / density / logit weights -- the
        p["rgb_linear.weight"] *= 12.0               # students end near 30 dB, where the model error (not the bf16
        p["alpha_linear.weight"] *= 4.0              # rounding of the evaluation render, ~50 dB) decides the PSNR
        p["semantic_linears.1.weight"] *= 4.0
    pool, held = _scene_rays(2048, 1), _scene_rays(1024, 2)
    with torch.no_grad():
        t_pool = to.render_rays(teacher, oc, pool, NC, NF)
        t_held = to.render_rays(teacher, oc, held, NC, NF)
    tgt_pool = {"rgb": t_pool["rgb_1"], "depth": t_pool["depth_1"], "semantic": t_pool["semantic_1"]}
    init = {"coarse": to.init_params(oc, 41, sigma_bias=0.2), "fine": to.init_params(oc, 42, sigma_bias=0.2)}
    g = torch.Generator().manual_seed(7)
    batches = [torch.randint(0, pool.shape[0], (BATCH,), generator=g) for _ in range(STEPS)]

    # ---- HIP student
    cfg = NS(N_samples=NC, N_importance=NF, num_classes=C, num_instances=0, precision="bf16", D=GEOM["D"], W=GEOM["W"],
             skips=list(GEOM["skips"]))
    net = make_network(cfg)
    net.nerf_0.load_state_dict(init["coarse"])
    net.nerf_1.load_state_dict(init["fine"])
    net = net.to(dev).train()
    rend = make_renderer(cfg, net)
    opt = torch.optim.Adam(net.parameters(), lr=LR)
    hip_losses = []
    for idx in batches:
        out = rend.render({"rays": pool[idx][None].to(dev)})
        loss = _loss({k: v[0] for k, v in out.items()}, {k: v[idx].to(dev) for k, v in tgt_pool.items()})
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        hip_losses.append(loss.item())
    with torch.no_grad():
        hip_eval = make_renderer(cfg, net.eval()).render({"rays": held[None].to(dev)})
    psnr_hip = _psnr(hip_eval["rgb_1"][0].cpu(), t_held["rgb_1"])

    # ---- oracle student: same init, same batches, fp32 torch autograd on the CPU (a few threads: the GEMMs are small, and
    # an OpenMP team over all 256 logical CPUs of the GPU host is several times slower than 16 threads)
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(16, n_thr))
    prm = {lv: {k: v.clone().requires_grad_(True) for k, v in init[lv].items()} for lv in ("coarse", "fine")}
    opt_o = torch.optim.Adam([p for d in prm.values() for p in d.values()], lr=LR)
    ora_losses = []
    for idx in batches:
        out = to.render_rays(prm, oc, pool[idx], NC, NF)
        loss = _loss(out, {k: v[idx] for k, v in tgt_pool.items()})
        opt_o.zero_grad(set_to_none=True)
        loss.backward()
        opt_o.step()
        ora_losses.append(loss.item())
    with torch.no_grad():
        ora_eval = to.render_rays({lv: {k: v.detach() for k, v in d.items()} for lv, d in prm.items()}, oc, held, NC, NF)
    psnr_ora = _psnr(ora_eval["rgb_1"], t_held["rgb_1"])
    torch.set_num_threads(n_thr)

    first, last = np.mean(hip_losses[:10]), np.mean(hip_losses[-10:])
    print(f"convergence: HIP loss {first:.5f} -> {last:.5f}; oracle loss {np.mean(ora_losses[:10]):.5f} -> {np.mean(ora_losses[-10:]):.5f}; "
          f"held-out PSNR HIP {psnr_hip:.3f} dB vs oracle-trained {psnr_ora:.3f} dB")
    assert last < 0.5 * first, (first, last)                                  # it learns
    assert abs(last - np.mean(ora_losses[-10:])) < 0.1 * np.mean(ora_losses[-10:])    # ... the same thing at the same rate
    assert abs(psnr_hip - psnr_ora) < 0.05, (psnr_hip, psnr_ora)              # north_star: PSNR within 0.05 dB


def _family(weights):
    """The CPU students of tests/golden/students_cpu.json: fp32 (the HIP student's batches), fp32:jitter (initialisation perturbed by
    1e-6 relative), fp32:order1.. (other batch orders), bf16_bwd (the HIP arithmetic restated on the CPU).  Made by
    `python tools/train_fidelity.py --weights W --students ... --out ...` (tests/golden/make_students.sh) in the build container:
    CPU students need no GPU, and on the GPU box's shared host eight of them took 11 minutes of the GPU budget (round 4, pass A)."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "students_cpu.json")) as f:
        d = json.load(f)[weights]
    return d["steps"], {r["name"]: r for r in d["rows"]}


@pytest.mark.parametrize("weights", ["unit", "image"])
def test_training_at_the_benched_geometry_against_the_fp32_family(dev, weights):
    """Training at the geometry bench.py runs (8x256 NeRFs with skip, semantic 45 + instance 32 heads, 64 + 128 samples, the 3D
    bbox prior, the trainer's loss wrapper with every term): the HIP students against the fp32 oracle students AND against the
    fp32 student's OWN spread.  Two weightings: "unit" (the wrapper's defaults: the six cross-entropy terms
    dominate; 100 Adam steps) and "image" (the colour term in charge; 150 steps).  Students:
      fp32, fp32:jitter, fp32:order1..   torch autograd through the oracle (CPU; committed: _family)
      bf16_bwd      the HIP path's arithmetic restated on the CPU (bf16 forward, every dY rounded to bf16)
      hip           NetworkWrapper on the MI355X, bf16 training kernels            } trained here, same initialisation and
      hip fp32      the same through the fp32 PARITY MODE of the training kernels  } batches as "fp32"
    Asserted, for both HIP students: |PSNR - PSNR_fp32| <= max(0.3 dB, spread of the fp32 family) -- the bound the round-3 verdict
    set in place of the fixed 1.5 dB / 25 % --, the total loss within max(1 %, the family's spread), the colour term inside the
    family's range (10 % margin); for the bf16 student: the two renderers give its checkpoint the same PSNR to 0.05 dB
    (north_star) and its semantic argmax map is the fp32 student's.
    What the numbers SHOW (profiles/README.md, round 4): at unit weights the fp32 student is chaotic -- 1e-6 on the initialisation
    moves its held-out PSNR by dB after 100 steps (25.3 vs 26.4 dB on the GPU host, 20.0 vs 27.8 dB in the build container; other
    batch orders 16.8 ... 28.8 dB), and the HIP fp32-mode student, which differs from "fp32" in summation order only, lands 5.8 dB
    away from it.  Round 3's 1.58 dB between the bf16 HIP student and ONE fp32 student was inside that spread, as is every bf16
    variant.  The per-step statement -- the HIP gradient IS the emulated bf16 gradient, to 1e-4 at the coarse level -- is pinned by
    test_gpu_backward.py::test_hip_gradient_is_the_emulated_bf16_gradient_at_the_benched_geometry."""
    import _students as S
    steps, f = _family(weights)
    W, w3d = S.WEIGHTS[weights]
    n_thr = torch.get_num_threads()
    torch.set_num_threads(min(16, n_thr))
    try:
        sc = S.scene(steps=steps)
        h = S.hip_student(sc, dev, W, w3d)
        h32 = S.hip_student(sc, dev, W, w3d, precision="fp32")
        with torch.no_grad():       # north_star "PSNR within 0.05 dB of reference": one checkpoint (the HIP-trained one), the two renderers
            same_w = to.render_rays(h["params"], sc.oc, sc.held, sc.Nc, sc.Nf, box=sc.box, box_ids=sc.ids)
    finally:
        torch.set_num_threads(n_thr)
    f32 = [r for n, r in f.items() if n.startswith("fp32")]
    assert len(f32) >= 3
    spread = max(r["psnr"] for r in f32) - min(r["psnr"] for r in f32)
    loss_spread = max(r["loss_last5"] for r in f32) - min(r["loss_last5"] for r in f32)
    rgb_lo, rgb_hi = min(r["rgb_last10"] for r in f32), max(r["rgb_last10"] for r in f32)
    hs, h32s = S.summary(h), S.summary(h32)
    print(f"[{weights}] held-out PSNR: hip bf16 {hs['psnr']:.3f}  hip fp32 mode {h32s['psnr']:.3f} | " +
          "  ".join("%s %.3f" % (n, r["psnr"]) for n, r in sorted(f.items())) + f" | fp32 family spread {spread:.3f} dB")
    print(f"[{weights}] total loss: hip bf16 {hs['loss_last5']:.4f}  hip fp32 mode {h32s['loss_last5']:.4f}  fp32 {f['fp32']['loss_last5']:.4f} "
          f"(family spread {loss_spread:.4f}); colour term: hip bf16 {hs['rgb_last10']:.6f}  hip fp32 mode {h32s['rgb_last10']:.6f}  "
          f"fp32 family [{rgb_lo:.6f}, {rgb_hi:.6f}]  bf16_bwd {f['bf16_bwd']['rgb_last10']:.6f}")
    for tag, st, run in (("bf16", hs, h), ("fp32 mode", h32s, h32)):
        first, last = np.mean(run["losses"][:5]), np.mean(run["losses"][-5:])
        assert last < (0.8 if weights == "unit" else 0.6) * first, (tag, first, last)                    # it learns
        # PSNR gate: the fp32 family's own spread, CAPPED at 1 dB.  At unit weights the family spans 12 dB (chaotic regime): a bound of
        # that size asserts nothing, so there the PSNR is printed only and the steps that CAN be compared are pinned by
        # test_training_trajectory_at_the_benched_geometry (parameters after 5 / 10 / 20 steps) and by the one-step gradient test.
        if spread <= 1.0:
            assert abs(st["psnr"] - f["fp32"]["psnr"]) <= max(0.3, spread), (tag, st["psnr"], f["fp32"]["psnr"], spread)
        else:
            # chaotic regime: the family's own RANGE (+- 1 dB) is still a statement -- a student that fell out of it (a collapsed
            # density field, a dead head) fails; inside it no order between students means anything
            lo_f, hi_f = min(r["psnr"] for r in f.values()), max(r["psnr"] for r in f.values())
            print(f"[{weights}] {tag}: PSNR {st['psnr']:.3f} vs fp32 {f['fp32']['psnr']:.3f}; the fp32 family's spread is {spread:.1f} dB, "
                  f"every committed student lies in [{lo_f:.2f}, {hi_f:.2f}] dB")
            assert lo_f - 1.0 <= st["psnr"] <= hi_f + 1.0, (tag, st["psnr"], lo_f, hi_f)
        assert abs(st["loss_last5"] - f["fp32"]["loss_last5"]) <= max(0.01 * abs(f["fp32"]["loss_last5"]), loss_spread), (tag, st["loss_last5"])
        assert rgb_lo / 1.1 <= st["rgb_last10"] <= rgb_hi * 1.1, (tag, st["rgb_last10"], rgb_lo, rgb_hi)
    psnr_same = S.psnr(same_w["rgb_1"], sc.t_held["rgb_1"])
    print(f"[{weights}] HIP-trained (bf16) weights rendered by HIP (bf16) {hs['psnr']:.3f} dB, by the oracle (fp32) {psnr_same:.3f} dB")
    assert abs(hs["psnr"] - psnr_same) < 0.05, (hs["psnr"], psnr_same)
    agree = float((h["eval"]["semantic_1"].argmax(-1) == torch.tensor(f["fp32"]["sem_argmax"])).float().mean())
    assert agree >= 0.99, agree


def _traj():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory_cpu.npz"))


def _traj_subset(name, numel, n_sub=512):
    """The fixed pseudo-random entries tests/golden/make_trajectory.py committed of parameter tensor `name`."""
    h = 1469598103934665603
    for c in name.encode():
        h = ((h ^ c) * 1099511628211) % (2 ** 64)
    g = torch.Generator().manual_seed(h % (2 ** 31))
    return torch.randperm(numel, generator=g)[:n_sub].sort().values


def _traj_scale(g, a_name, b_sub, k, names):
    """Pooled projection coefficient <x_b, x_a> / <x_a, x_a> of the updates theta_k - theta_0 (a: committed CPU student, b: measured):
    1 for the same trajectory, whatever zero-mean distance the two have; a systematic scale error of the step (Adam's bias
    correction, a dropped loss weight, a learning rate) shows here undiluted."""
    num = den = 0.0
    for pn in names:
        x = g[f"{a_name}/{k}/{pn}/sub"].astype(np.float64)
        y = b_sub[pn].astype(np.float64)
        num += float(np.sum(x * y))
        den += float(np.sum(x * x))
    return num / den


def _traj_distance(g, a_name, b_sub, k, names):
    """(worst tensor, worst relative L2, pooled relative L2) of the updates theta_k - theta_0 on the committed entries."""
    worst, wn, num, den = 0.0, None, 0.0, 0.0
    for pn in names:
        x = g[f"{a_name}/{k}/{pn}/sub"].astype(np.float64)
        y = b_sub[pn].astype(np.float64) if isinstance(b_sub, dict) else g[f"{b_sub}/{k}/{pn}/sub"].astype(np.float64)
        nx = np.linalg.norm(x)
        if nx == 0.0:
            assert np.linalg.norm(y) == 0.0, pn
            continue
        rel = np.linalg.norm(x - y) / nx
        if rel > worst:
            worst, wn = rel, pn
        num += np.sum((x - y) ** 2)
        den += np.sum(x ** 2)
    return wn, worst, float(np.sqrt(num / den))


def test_training_trajectory_at_the_benched_geometry(dev):
    """a multi-step training check that CAN fail.  Held-out PSNR after 100 unit-weight steps is chaotic (above);
    the PARAMETERS after k = 5, 10, 20 Adam steps are not.  Same scene, initialisation and batches as the students of
    tests/_students.py; compared per parameter tensor: the update theta_k - theta_0 (on the 512 committed entries per tensor,
    tests/golden/trajectory_cpu.npz made by tests/golden/make_trajectory.py on the CPU) of
        the HIP bf16 student        vs  "bf16_bwd" (the training kernels' arithmetic restated on the CPU)
        the HIP fp32-mode student   vs  "fp32"     (torch autograd through the oracle)
        the HIP bf16 student run through train.GraphedStep  ==  the eager HIP bf16 student (same fused Adam), bit for bit.
    Bounds: relative to what two CPU students of the SAME arithmetic differ by when the initialisation is jittered by 1e-6
    ("fp32" vs "fp32_jitter", "bf16_bwd" vs "bf16_bwd_jitter" at the same k) -- the trajectory's own sensitivity -- times a
    margin, with a floor; and always far below the distance a WRONG step produces: Adam's step counter off by one changes every
    update by 14 % at k = 5 (bias correction), a missing gradient term or an un-rounded dY tensor moves whole tensors by tens of
    per cent.  The fp32-vs-bf16 distance (a different arithmetic, printed) is the scale of "another trajectory"."""
    import _students as S
    g = _traj()
    W, w3d = S.WEIGHTS["unit"]
    KS = (5, 10, 20)
    sc = S.scene(steps=max(KS))
    names = [f"{lv}.{k}" for lv in ("coarse", "fine") for k in sc.init[lv]]
    init = {f"{lv}.{k}": v for lv, d in sc.init.items() for k, v in d.items()}

    def subs(student, k):
        out = {}
        for pn in names:
            d = (student["snaps"][k][pn] - init[pn]).reshape(-1)
            out[pn] = d[_traj_subset(pn, d.numel())].numpy()
        return out

    h = S.hip_student(sc, dev, W, w3d, snap=KS, evaluate=False)
    h32 = S.hip_student(sc, dev, W, w3d, precision="fp32", snap=KS, evaluate=False)
    hf = S.hip_student(sc, dev, W, w3d, snap=KS, evaluate=False, fused_adam=True)      # eager, the optimiser GraphedStep needs
    hg = S.hip_student(sc, dev, W, w3d, snap=KS, evaluate=False, graphed=True)
    for k in KS:
        for pn in names:
            assert torch.equal(hf["snaps"][k][pn], hg["snaps"][k][pn]), ("GraphedStep != eager", k, pn)
    for k in KS:
        _, jit32_w, jit32 = _traj_distance(g, "fp32", "fp32_jitter", k, names)
        _, jit16_w, jit16 = _traj_distance(g, "bf16_bwd", "bf16_bwd_jitter", k, names)
        _, x_w, x_p = _traj_distance(g, "fp32", "bf16_bwd", k, names)
        n16, w16, p16 = _traj_distance(g, "bf16_bwd", subs(h, k), k, names)
        n32, w32, p32 = _traj_distance(g, "fp32", subs(h32, k), k, names)
        print(f"[trajectory k={k}] HIP bf16 vs bf16_bwd: pooled {p16:.3e} worst {w16:.3e} ({n16}) | HIP fp32 mode vs fp32: pooled {p32:.3e} "
              f"worst {w32:.3e} ({n32}) | CPU jitter 1e-6: fp32 {jit32:.3e} (worst {jit32_w:.3e}), bf16_bwd {jit16:.3e} (worst {jit16_w:.3e}) | "
              f"fp32 vs bf16_bwd {x_p:.3e} (worst {x_w:.3e})")
        a16, a32 = _traj_scale(g, "bf16_bwd", subs(h, k), k, names), _traj_scale(g, "fp32", subs(h32, k), k, names)
        print(f"[trajectory k={k}] pooled projection of the HIP update on the CPU student's: bf16 {a16:.4f}, fp32 mode {a32:.4f}")
        assert abs(a16 - 1.0) <= TRAJ_SCALE and abs(a32 - 1.0) <= TRAJ_SCALE, (k, a16, a32)
        assert p32 <= TRAJ_POOLED["fp32"][k], (k, p32)
        assert p16 <= TRAJ_POOLED["bf16"][k], (k, p16)
        assert w32 <= TRAJ_WORST["fp32"][k] and w16 <= TRAJ_WORST["bf16"][k], (k, w32, w16)
    # teeth: a step that is wrong the way a mis-counted Adam step is (bias correction of step k + 1 instead of k: every update
    # x (1 - 0.9^5) / (1 - 0.9^6) = 0.874 at k = 5) must FAIL the fp32-mode bound -- applied to the measured updates, no second run
    broken = {pn: v * 0.874 for pn, v in subs(h32, 5).items()}
    _, _, p_broken = _traj_distance(g, "fp32", broken, 5, names)
    assert p_broken > TRAJ_POOLED["fp32"][5], p_broken
    # ... and for the bf16 student, whose pooled DISTANCE bound (0.18 at k = 5: the bf16 trajectory's own sensitivity is 0.12) would
    # let that defect through: the projection coefficient of its update on bf16_bwd's is 0.874 instead of 1
    broken16 = {pn: v * 0.874 for pn, v in subs(h, 5).items()}
    assert abs(_traj_scale(g, "bf16_bwd", broken16, 5, names) - 1.0) > TRAJ_SCALE
    # the loss curves of the first 20 steps agree as well (same batches, same arithmetic)
    l16, l32 = np.asarray(h["losses"]), np.asarray(h32["losses"])
    assert np.max(np.abs(l16 - g["bf16_bwd/losses"]) / g["bf16_bwd/losses"]) < 2e-2
    assert np.max(np.abs(l32 - g["fp32/losses"]) / g["fp32/losses"]) < 2e-3


# Bounds of test_training_trajectory_at_the_benched_geometry: pooled / worst-tensor relative L2 of the parameter UPDATE after k steps.
# Set from the first measurement on the MI355X (profiles/README.md, round 5) at ~2x the measured pooled distance (worst tensor:
# 1.3 - 2x); a deliberately broken step (Adam's step counter off by one: every update x 0.874 at k = 5) is checked to land above
# them inside the test.
# fp32 mode: 2.5 x the distance two CPU fp32 students have when the initialisation is jittered by 1e-6 (pooled 1.38e-2 / 9.9e-3 /
# 8.3e-3, worst tensor 6.3e-2 / 4.2e-2 / 2.7e-2 at k = 5 / 10 / 20); measured on the MI355X (r05e): pooled 1.58e-2 / 1.05e-2 / 8.9e-3,
# worst 8.6e-2 / 5.5e-2 / 4.0e-2 -- the HIP fp32 mode IS a jittered fp32 student.  bf16: 1.5 x the bf16_bwd students' own jitter
# distance (pooled 1.19e-1 / 9.2e-2 / 7.1e-2, worst 0.42 / 0.31 / 0.21); measured 9.0e-2 / 6.9e-2 / 5.1e-2, worst 0.48 / 0.37 / 0.21.
TRAJ_SCALE = 0.03      # |pooled projection coefficient - 1| of a HIP student's update on its CPU twin's: measured 0.9958 / 0.9976 / 0.9989 (bf16) and 0.9999 / 1.0000 / 1.0000 (fp32 mode) at k = 5 / 10 / 20 (profiles/r06/r06g); an off-by-one Adam step gives 0.874
TRAJ_POOLED = {"fp32": {5: 0.035, 10: 0.025, 20: 0.021}, "bf16": {5: 0.18, 10: 0.14, 20: 0.107}}
TRAJ_WORST = {"fp32": {5: 0.16, 10: 0.105, 20: 0.07}, "bf16": {5: 0.63, 10: 0.46, 20: 0.31}}
