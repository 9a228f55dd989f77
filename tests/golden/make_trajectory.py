"""tests/golden/trajectory_cpu.npz: the first 20 Adam steps of the CPU students at the benched geometry, for
tests/test_gpu_convergence.py::test_training_trajectory_at_the_benched_geometry.

Same scene, initialisation and batches as tests/_students.py (unit loss weights -- the regime whose PSNR after 100 steps is
chaotic; over 5 / 10 / 20 steps the PARAMETERS are not).  Students: fp32, fp32:jitter (initialisation x (1 + 1e-6 u)), bf16_bwd
(the HIP training path's arithmetic restated on the CPU), bf16_bwd:jitter.  Committed per student, step count k in (5, 10, 20)
and parameter tensor: the update theta_k - theta_0 on a FIXED pseudo-random subset of <= 512 entries (the whole tensors would be
32 MB) and the full update's L2 norm.  CPU only, ~6 min on 8 cores:  python tests/golden/make_trajectory.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _students as S  # noqa: E402

KS = (5, 10, 20)
N_SUB = 512


def subset(name, numel):
    g = torch.Generator().manual_seed(abs(hash_name(name)) % (2 ** 31))
    return torch.randperm(numel, generator=g)[:N_SUB].sort().values


def hash_name(name):
    h = 1469598103934665603
    for c in name.encode():
        h = ((h ^ c) * 1099511628211) % (2 ** 64)
    return h


def main():
    torch.set_num_threads(int(os.environ.get("PNR_THREADS", "8")))
    sc = S.scene(steps=max(KS))
    W, w3d = S.WEIGHTS["unit"]
    init = {f"{lv}.{k}": v for lv, d in sc.init.items() for k, v in d.items()}
    out = {}
    for name, mode, jit in (("fp32", "fp32", 0.0), ("fp32_jitter", "fp32", 1e-6), ("bf16_bwd", "bf16_bwd", 0.0), ("bf16_bwd_jitter", "bf16_bwd", 1e-6)):
        r = S.oracle_student(sc, mode, W, w3d, init_jitter=jit, snap=KS, log=print)
        out[f"{name}/losses"] = np.asarray(r["losses"], np.float64)
        for k in KS:
            for pn, v in r["snaps"][k].items():
                d = (v - init[pn]).reshape(-1)
                out[f"{name}/{k}/{pn}/sub"] = d[subset(pn, d.numel())].numpy().astype(np.float32)
                out[f"{name}/{k}/{pn}/norm"] = np.float64(d.double().norm().item())
        print(name, "done", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "trajectory_cpu.npz"), **out)
    # the distances the test's bounds are derived from
    for a, b in (("fp32", "fp32_jitter"), ("bf16_bwd", "bf16_bwd_jitter"), ("fp32", "bf16_bwd")):
        for k in KS:
            worst, pooled_n, pooled_d = 0.0, 0.0, 0.0
            for pn in init:
                x, y = out[f"{a}/{k}/{pn}/sub"].astype(np.float64), out[f"{b}/{k}/{pn}/sub"].astype(np.float64)
                rel = np.linalg.norm(x - y) / max(np.linalg.norm(x), 1e-30)
                worst = max(worst, rel)
                pooled_n += np.sum((x - y) ** 2)
                pooled_d += np.sum(x ** 2)
            print(f"{a} vs {b} @ k={k}: worst tensor rel L2 {worst:.3e}, pooled {np.sqrt(pooled_n / pooled_d):.3e}")


if __name__ == "__main__":
    main()
