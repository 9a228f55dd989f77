"""Round-5 experiment (verdict r4 item 1): TIMING-ONLY prototypes of the fused MLP's steady hidden-layer loop with one wave per
SIMD and TWO 32-sample tiles per wave, next to HEAD's fused fine-level launch on the SAME box in the SAME process.

  python tools/proto_two_tile.py [--json out.json] [--iters 5]

Prints per variant: ms per launch of 65,536 x 192 samples, shader MHz, cycles per MFMA (one wave = one SIMD's matrix pipe),
and `ms_at_real_work` = ms x 1336 / 1280 (the prototype runs 1280 MFMAs per 32-sample tile, the real plan 1336).  HEAD's row is
the real kernel (encoders, skip, heads, fused epilogue); the prototypes leave all of that out, so a prototype must beat HEAD's
STEADY hidden chunks, not HEAD's launch: the decision rule is in profiles/README.md (round 5)."""
import argparse
import json
import os
import sys
from types import SimpleNamespace as NS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from panopticnerf_amd import benchlib, make_network, ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--skip-head", action="store_true")
    ap.add_argument("--debug-asm", action="store_true", help="run the staged debug builds of the assembly kernel (flags 101..104)")
    ap.add_argument("--one", default=None, help="(internal) run ONE prototype variant in this process: asm:7, hipcc:6, ...")
    a = ap.parse_args()
    if a.one:
        kind, fl = a.one.split(":")
        image = benchlib.proto_two_tile_image(torch.device("cuda:0"))
        benchlib.proto_two_tile(image, 65536 * 192, int(fl), 1, kind == "asm")
        best = min((benchlib.proto_two_tile(image, 65536 * 192, int(fl), a.iters, kind == "asm") for _ in range(3)), key=lambda t: t[0])
        print("RESULT " + json.dumps(best), flush=True)
        return
    if a.debug_asm:
        import subprocess
        for fl in (108, 107, 102, 104, 0, 7):
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", "asm:%d" % fl, "--iters", "1"], capture_output=True, text=True, timeout=120)
            print("asm debug stage %d: rc %d %s" % (fl, p.returncode, (p.stdout + p.stderr)[-400:].replace("\n", " | ")), flush=True)
        return
    dev = torch.device("cuda:0")
    R, N = 65536, 192
    S = R * N
    rows = []
    groups_wg0 = (S // 256 + 255) // 256
    if not a.skip_head:
        torch.manual_seed(0)
        net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
        synthetic.trained_like_(net)
        rays = synthetic.camera_rays()[:R].to(dev)
        z = ops.stratified(rays, N)
        desc, img = net.packed(1, dev, fused=True)
        benchlib.time_mlp_forward_tiles(desc, img, rays, z, 2)
        best = min((benchlib.time_mlp_forward_tiles(desc, img, rays, z, a.iters) for _ in range(3)), key=lambda t: t[0])
        ms, mhz = best
        cyc = ms * 1e-3 * mhz * 1e6 / (groups_wg0 * 1336 * 2)        # two waves share a SIMD's pipe: cycles per MFMA of the PIPE
        rows.append({"variant": "HEAD k_mlp_pp<fused, plan 1> (8 waves x 1 tile, the real kernel)", "ms": ms, "mhz": mhz,
                     "cycles_per_mfma_per_simd": cyc, "ms_at_real_work": ms})
        tf, pm = benchlib.probe_mfma_peak(True)
        rows.append({"variant": "register-only random-operand MFMA loop (8 waves per CU)", "tflops": tf, "mhz": pm,
                     "cycles_per_mfma_per_simd": 32.0 * (2.0 * 32 * 32 * 16 * 1024 * pm * 1e6 / 32.0) / (tf * 1e12)})
    for r in rows:
        print("%-84s %s" % (r["variant"], "  ".join("%s=%.3f" % (k, v) for k, v in r.items() if isinstance(v, float))), flush=True)
    names = {7: "two-tile prototype (fragment reads + pack/ReLU + LDS-DMA pieces)", 6: "  without the LDS-DMA pieces",
             3: "  without the pack/ReLU epilogue", 2: "  fragment reads only", 0: "  bare MFMA stream (no reads, no epilogue, no pieces)"}
    import subprocess
    for kind in ("asm", "hipcc"):
        for fl in (7, 6, 3, 2, 0):          # one process per variant: a faulting prototype must not take the others with it
            row = {"variant": kind + " " + names[fl]}
            try:
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", "%s:%d" % (kind, fl), "--iters", str(a.iters)],
                                   capture_output=True, text=True, timeout=120)
                res = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
                if res:
                    ms, mhz, cyc = json.loads(res[-1][7:])
                    row.update({"ms": ms, "mhz": mhz, "cycles_per_mfma_per_simd": cyc, "ms_at_real_work": ms * 1336.0 / 1280.0})
                else:
                    row["error"] = "rc %d: %s" % (p.returncode, (p.stderr or p.stdout)[-300:].replace("\n", " | "))
            except subprocess.TimeoutExpired:
                row["error"] = "timeout (hang?)"
            rows.append(row)
            print("%-84s %s" % (row["variant"], "  ".join("%s=%.3f" % (k, v) for k, v in row.items() if isinstance(v, float)) or row.get("error", "")), flush=True)
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"device": torch.cuda.get_device_name(0), "samples_per_launch": S, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
