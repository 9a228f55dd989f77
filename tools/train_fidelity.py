"""Training-fidelity experiment at the benched geometry: which part of the HIP student's PSNR gap against
the fp32 oracle student is the PRECISION of a bf16 backward, and how large is an fp32 student's own spread?

    python tools/train_fidelity.py [--steps 150] [--procs 4] [--threads 8] [--weights unit|image] [--hip] \
        [--students fp32,bf16_fwd,bf16_bwd,bf16_hilo,fp32:order1,fp32:order2,fp32:order3,fp32:jitter] [--out FILE.json]

Every CPU student (tests/_students.py) runs in its own process with its own OpenMP team; `--hip` adds the HIP student(s)
(the bf16 training path and its fp32 parity mode) on cuda:0.  Prints one line per student and writes the table as JSON.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(job):
    from tests import _students as S
    return S.run_job(job)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--weights", default="unit")
    ap.add_argument("--students", default="fp32,bf16_fwd,bf16_bwd,bf16_hilo,fp32:order1,fp32:order2,fp32:order3,fp32:jitter")
    ap.add_argument("--hip", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    jobs = [(n, a.steps, a.threads, a.weights) for n in a.students.split(",") if n]
    rows = []
    if a.hip:
        import torch
        from tests import _students as S
        sc = S.scene(steps=a.steps, threads=32)
        W, w3d = S.WEIGHTS[a.weights]
        for gp in ("bf16", "fp32"):
            t0 = time.time()
            s = S.summary(S.hip_student(sc, torch.device("cuda:0"), W, w3d, precision=gp))
            s["name"], s["seconds"] = "hip:" + gp, round(time.time() - t0, 1)
            rows.append(s)
            print(json.dumps(s), flush=True)
    if jobs:
        with mp.get_context("spawn").Pool(a.procs) as pool:
            for s in pool.imap_unordered(_run, jobs):
                rows.append(s)
                print(json.dumps({k: v for k, v in s.items() if k not in ("rgb_curve", "sem_argmax")}), flush=True)
    rows.sort(key=lambda r: r["name"])
    print("\n%-14s %9s %11s %11s" % ("student", "PSNR dB", "loss", "rgb term"))
    for r in rows:
        print("%-14s %9.3f %11.4f %11.6f" % (r["name"], r["psnr"], r["loss_last5"], r["rgb_last10"]))
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"steps": a.steps, "weights": a.weights, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
