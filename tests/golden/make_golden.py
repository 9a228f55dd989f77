"""Generates tests/golden/*.npz -- small seeded input/expected-output vectors for every stage
of the path.

PARITY UNPINNED: the reference's code branches are not in /root/reference,
so these vectors come from this repo's own oracle, accepted only where its independent
restatements agree (torch-order C, vectorised torch fp32, plain-loop numpy float64; see
tests/test_oracle.py).  The sampler vectors (z, sample indices, z_samples, z_fine) are produced by
torch ops AS WRITTEN (torch_oracle) -- the reference's arithmetic substrate -- and the C oracle /
HIP kernels must reproduce them bit for bit.  When the `panopticnerf` branch is mounted, regenerate them by importing
the real renderer here and keep this script as the record of how.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle as co  # noqa: E402
from oracle import torch_oracle as to  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rot_y(yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)


def main():
    rng = np.random.default_rng(20260926)
    R, Nc, Nf, C, K, M, MH = 96, 64, 128, 6, 5, 24, 8
    o = rng.normal(0, 1.0, (R, 3)) + np.array([0, 1.5, 0])
    d = rng.normal(0, 0.3, (R, 3)) + np.array([0, 0, 1.0])
    rays = np.concatenate([o, d, np.full((R, 1), 0.5), np.full((R, 1), 60.0)], 1).astype(np.float32)
    t_rand = rng.random((R, Nc)).astype(np.float32)
    u = rng.random((R, Nf)).astype(np.float32)
    box = np.concatenate([rng.uniform([-8, -1, 4], [8, 3, 50], (M, 3)),
                          np.stack([rot_y(y).reshape(-1) for y in rng.uniform(0, np.pi, M)]),
                          rng.uniform(0.5, 4.0, (M, 3))], 1).astype(np.float32)
    box_ids = np.stack([rng.integers(0, C, M), rng.integers(0, K, M)], 1).astype(np.int32)

    g = {"rays": rays, "t_rand": t_rand, "u": u, "box": box, "box_ids": box_ids,
         "dims": np.array([R, Nc, Nf, C, K, M, MH], np.int32)}
    # a3 -- FROM TORCH AS WRITTEN (oracle/torch_oracle.py: torch.linspace and elementwise ops on this container's torch CPU);
    # the C oracle restates torch's op order and must reproduce every bit
    tr = torch.tensor(rays)
    g["z_det"] = to.stratified(tr, Nc).numpy()
    g["z_lindisp"] = to.stratified(tr, Nc, lindisp=True).numpy()
    g["z_perturb"] = to.stratified(tr, Nc, t_rand=torch.tensor(t_rand)).numpy()
    assert np.array_equal(g["z_det"], co.stratified(rays, Nc)) and np.array_equal(g["z_lindisp"], co.stratified(rays, Nc, lindisp=True))
    assert np.array_equal(g["z_perturb"], co.stratified(rays, Nc, t_rand=t_rand))
    g["pts"] = co.points(rays, g["z_perturb"])
    # a4
    x = g["pts"].reshape(-1, 3)[:257]
    g["embed_x"] = x
    g["embed_L10"] = co.embed(x, 10)
    g["embed_L4"] = co.embed(x, 4)
    # a8
    hit_t, hit_box, hit_count = co.bbox_hits(rays, box, MH)
    g["hit_t"], g["hit_box"], g["hit_count"] = hit_t, hit_box, hit_count
    ls, li = co.sample_labels(g["z_perturb"], hit_t, hit_box, hit_count, box_ids)
    g["label_sem"], g["label_inst"] = ls, li
    # a6 on synthetic raw (densities chosen so that weights are neither all 0 nor saturated)
    raw = rng.normal(0, 1.0, (R, Nc, 4 + C + K)).astype(np.float32)
    raw[..., 3] = rng.normal(0.0, 0.15, (R, Nc)).astype(np.float32)
    noise = rng.normal(0, 0.05, (R, Nc)).astype(np.float32)
    g["raw"], g["noise"] = raw, noise
    for sm in (0, 1):
        out = co.composite(raw, g["z_perturb"], rays, C, K, noise=noise, label_sem=ls, label_inst=li,
                           sem_mode=sm, white_bkgd=bool(sm))
        for k, v in out.items():
            g[f"comp{sm}_{k}"] = v
    # a7
    w = g["comp0_weights"]
    for tag, uu in (("det", None), ("rand", u)):
        # sample indices, z_samples and the sorted union FROM TORCH AS WRITTEN (sum, cumsum, linspace, searchsorted, sort)
        zf_t, zs_t, i_t = to.importance_z(torch.tensor(g["z_perturb"]), torch.tensor(w), Nf, None if uu is None else torch.tensor(uu))
        g[f"pdf_{tag}_zs"], g[f"pdf_{tag}_inds"] = zs_t.numpy(), i_t.numpy().astype(np.int32)
        g[f"pdf_{tag}_zfine"] = zf_t.numpy()
        zs, inds = co.sample_pdf(g["z_perturb"], w, Nf, uu)
        assert np.array_equal(inds, g[f"pdf_{tag}_inds"]) and np.array_equal(zs, g[f"pdf_{tag}_zs"])
        assert np.array_equal(co.merge_sorted(g["z_perturb"], zs), g[f"pdf_{tag}_zfine"])
    # a5: small MLP with stored weights; big MLP from seed
    cfg_s = to.mlp_config(D=4, W=128, skips=(1,), n_sem=C, n_inst=K, head_W=64)
    p_s = to.init_params(cfg_s, seed=3)
    for k, v in p_s.items():
        g["mlp_s." + k] = v.numpy()
    S = 160
    rr, zz = torch.tensor(rays[:5]), torch.tensor(g["z_perturb"][:5, :32].copy())
    g["mlp_rays"], g["mlp_z"] = rr.numpy(), zz.numpy()
    g["mlp_s_raw_fp32"] = to.run_network(p_s, cfg_s, rr, zz).numpy()
    g["mlp_s_raw_bf16"] = to.run_network(p_s, cfg_s, rr, zz, emulate_bf16=True).numpy()
    cfg_b = to.mlp_config(n_sem=C, n_inst=K)
    p_b = to.init_params(cfg_b, seed=4, sigma_bias=0.05)
    g["mlp_b_seed"] = np.array([4], np.int32)
    g["mlp_b_raw_fp32"] = to.run_network(p_b, cfg_b, rr, zz).numpy()
    g["mlp_b_raw_bf16"] = to.run_network(p_b, cfg_b, rr, zz, emulate_bf16=True).numpy()
    assert S == rr.shape[0] * zz.shape[1]
    # a2 end to end (fp32), 32 rays, big nets from seeds 4 (coarse) / 5 (fine)
    p_f = to.init_params(cfg_b, seed=5, sigma_bias=0.05)
    e2e = to.render_rays({"coarse": p_b, "fine": p_f}, cfg_b, torch.tensor(rays[:32]), Nc, Nf,
                         box=torch.tensor(box), box_ids=torch.tensor(box_ids), max_hits=MH)
    for k in ("rgb_0", "depth_0", "acc_0", "rgb_1", "depth_1", "acc_1", "semantic_1", "instance_1",
              "fix_semantic_1", "fix_instance_1", "z_vals_1"):
        g["e2e_" + k] = e2e[k].numpy()
    np.savez_compressed(os.path.join(HERE, "path_small.npz"), **g)
    print("wrote", os.path.join(HERE, "path_small.npz"), sum(v.nbytes for v in g.values()) // 1024, "KiB raw")
    make_configs()


def config_case(n, R=24):
    """Inputs of the per-config end-to-end fixture: (oracle MLP config, params, rays, box, ids).  Shared with
    tests/test_gpu_configs.py so that the test renders exactly what the fixture holds."""
    from panopticnerf_amd import synthetic
    c = synthetic.BASELINE_CONFIGS[n]
    oc = to.mlp_config(D=c["D"], W=c["W"], skips=tuple(c["skips"]), n_sem=c["num_classes"], n_inst=c["num_instances"],
                       head_W=c["W"] // 2)
    params = {"coarse": to.init_params(oc, seed=10 + n, sigma_bias=0.05), "fine": to.init_params(oc, seed=20 + n, sigma_bias=0.05)}
    rays = synthetic.camera_rays()[(7 * n) :: (1408 * 376) // R][:R].contiguous()
    box = ids = None
    if c["bbox"]:
        box, ids = synthetic.random_boxes(48, max(c["num_classes"], 1), max(c["num_instances"], 1), seed=30 + n)
    return c, oc, params, rays, box, ids


def make_configs():
    """tests/golden/configs.npz: one small end-to-end render (torch fp32 oracle) per BASELINE.json config 1..5."""
    g = {}
    for n in range(1, 6):
        c, oc, params, rays, box, ids = config_case(n)
        out = to.render_rays(params, oc, rays, c["N_samples"], c["N_importance"], box=box, box_ids=ids)
        lv = 1 if c["N_importance"] else 0
        keys = [f"{k}_{l}" for l in range(lv + 1) for k in ("rgb", "depth", "acc")]
        if c["num_classes"]:
            keys += [f"semantic_{lv}"] + ([f"fix_semantic_{lv}"] if c["bbox"] else [])
        if c["num_instances"]:
            keys += [f"instance_{lv}"] + ([f"fix_instance_{lv}"] if c["bbox"] else [])
        keys.append(f"z_vals_{lv}")
        for k in keys:
            g[f"c{n}_{k}"] = out[k].numpy()
        g[f"c{n}_keys"] = np.array(keys)
    np.savez_compressed(os.path.join(HERE, "configs.npz"), **g)
    print("wrote", os.path.join(HERE, "configs.npz"), sum(v.nbytes for v in g.values()) // 1024, "KiB raw")


if __name__ == "__main__":
    main()
