"""Per-chunk s_memtime timelines of the FUSED ping-pong MLP launch (k_mlp_pp<.., FUSE>), iteration PNR_TRACE_ITER of workgroup 0,
from -DPNR_TRACE=1 builds under build/ab (tools/build_ab.sh).  One tool, four views (round 4: mlp_trace_l / _lm / _m folded in):

  --view chunks (default; build -DPNR_TRACE_MASK=0x5c: stamps 2 = M start, 3 = M end, 4 = after the M->L barrier, 6 = L work done)
      period  = wave 0's M start of chunk c+1 - of chunk c  (what the chunk costs the workgroup: P's M + Q's M + 2 hand-overs)
      M_P/M_Q = M phase of group P (wave 0) / Q (wave 4);  bar_P/bar_Q = their wait at the M->L barrier
      Lw_P/Lw_Q = the L phase's own work (without the wait at the L->M barrier);  L_P = wave 0's L phase before this chunk
  --view l      (build -DPNR_TRACE_MASK=0x53: stamps 4 = L start, 0 = refill_begin, 1 = refill issued, 6 = arrival at the L->M barrier)
      per chunk, wave 0 / wave 4:  a = 4 -> 0 (early fragment requests), b = 0 -> 1 (refill pieces + epilogue), c = 1 -> 6 (hand-over,
      late fragment / bias requests, the drain of the phase's LDS reads)
  --view m      (build -DPNR_TRACE_MID=1 -DPNR_TRACE_MASK=0x8f: stamps 2 = M start, 0 / 7 / 1 = before MFMA NF/4, NF/2, 3NF/4, 3 = all issued)
      the four quarters of every chunk's M phase: which MFMAs of a phase are the slow ones
  --view lm <libL> <libM>   L work (from an 0x53 build) beside M phases (from an 0x5c build), one line per chunk

usage: python tools/mlp_trace_fused.py [--view chunks|l|m|lm] <lib name under build/ab> [<second lib for lm>]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCH, NST = 48, 8
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    os.environ["PNR_LIB_PATH"] = sys.argv[2]
    import torch
    from types import SimpleNamespace as NS
    dev = torch.device("cuda:0")
    trace = torch.zeros((8, NCH, NST), dtype=torch.int64, device=dev)
    os.environ["PNR_TRACE_PTR"] = str(trace.data_ptr())
    from panopticnerf_amd import benchlib, make_network, ops, synthetic
    torch.manual_seed(0)
    net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
    synthetic.trained_like_(net)
    net = net.to(dev)
    rays = synthetic.camera_rays()[:65536].to(dev)
    z = ops.stratified(rays, 192)
    desc, img = net.packed(1, dev, fused=os.environ.get('PNR_PLAN', '1') != '0')
    box, ids = (t.to(dev) for t in synthetic.random_boxes(64, 45, 32))
    h = ops.bbox_hits(rays, box, 8)
    ls, li = ops.sample_labels(z, h[0], h[1], h[2], ids)
    ms, mhz = benchlib.time_mlp_forward_tiles(desc, img, rays, z, 1)
    torch.cuda.synchronize()
    print("TRACE " + json.dumps({"t": trace.cpu().tolist(), "ms": ms, "mhz": mhz}))
    sys.exit(0)


def run_trace(name):
    lib = os.path.join(ROOT, "build", "ab", "libpnr_%s.so" % name)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], capture_output=True, text=True, timeout=240)
    line = [l for l in out.stdout.splitlines() if l.startswith("TRACE ")]
    if not line:
        print(out.stdout[-2000:], out.stderr[-2000:])
        sys.exit(1)
    return json.loads(line[0][6:])


def chunk_names():
    """chunk names of the 8x256 + 45/32 plan (pnr_mlp_plan.h order)"""
    plan1 = os.environ.get("PNR_PLAN", "1") != "0"
    names = ["trunk0"] * (1 if plan1 else 2)       # plan 1: layer 0 is one chunk
    for l in range(1, 8):
        names += ["trunk%d" % l] * 4
    if plan1:
        return names + ["feature"] * 4 + ["views"] * 2 + ["rgbsigma"] + ["sem0"] * 2 + ["inst0"] * 2 + ["logits"]
    return names + ["feature"] * 4 + ["views"] * 2 + ["rgbsigma"] + ["sem0"] * 2 + ["sem1"] * 2 + ["inst0"] * 2 + ["inst1"]


def view_chunks(name):
    D = run_trace(name)
    T, names = D["t"], chunk_names()
    n = len(names)
    print("traced launch (trace build): %.3f ms at %.0f MHz; %d chunks" % (D["ms"], D["mhz"], n))
    print("%3s %-9s %7s %6s %6s %6s %6s %6s %6s %6s" % ("c", "layer", "period", "M_P", "bar_P", "M_Q", "bar_Q", "L_P", "Lw_P", "Lw_Q"))
    tot, by = 0, {}
    for c in range(n):
        nxt = T[0][c + 1][2] if c + 1 < n else 0
        per = nxt - T[0][c][2] if nxt else 0
        mp, bp = T[0][c][3] - T[0][c][2], T[0][c][4] - T[0][c][3]
        mq, bq = T[4][c][3] - T[4][c][2], T[4][c][4] - T[4][c][3]
        lp = T[0][c][2] - T[0][c - 1][4] if c else 0
        # L work proper: after the M->L barrier of chunk c-1 up to the arrival at the L->M barrier of chunk c (stamp 6)
        lwp = T[0][c][6] - T[0][c - 1][4] if c else 0
        lwq = T[4][c][6] - T[4][c - 1][4] if c else 0
        print("%3d %-9s %7d %6d %6d %6d %6d %6d %6d %6d" % (c, names[c], per, mp, bp, mq, bq, lp, lwp, lwq))
        if per > 0:
            tot += per
            by[names[c]] = by.get(names[c], 0) + per
    print("sum of periods (chunks 0..%d): %d cycles" % (n - 2, tot))
    for k, v in by.items():
        print("   %-9s %7d  %5.1f %%" % (k, v, 100.0 * v / tot))


def view_l(name):
    D = run_trace(name)
    T = D["t"]
    print("trace build: %.3f ms @ %.0f MHz" % (D["ms"], D["mhz"]))
    print(" c |  P: a     b     c   total |  Q: a     b     c   total")
    for c in range(1, len(chunk_names())):
        row = []
        for w in (0, 4):
            t4, t0, t1, t6 = T[w][c - 1][4], T[w][c - 1][0], T[w][c - 1][1], T[w][c][6]
            row += [t0 - t4, t1 - t0, t6 - t1, t6 - t4]
        print("%2d | %5d %5d %5d %6d | %5d %5d %5d %6d" % tuple([c] + row))


def view_m(name):
    D = run_trace(name)
    T = D["t"]
    print("trace build: %.3f ms @ %.0f MHz" % (D["ms"], D["mhz"]))
    print(" c |  P: q1    q2    q3    q4  total |  Q: q1    q2    q3    q4  total")
    for c in range(0, len(chunk_names())):
        row = []
        for w in (0, 4):
            t2, t0, t7, t1, t3 = (T[w][c][k] for k in (2, 0, 7, 1, 3))
            row += [t0 - t2, t7 - t0, t1 - t7, t3 - t1, t3 - t2]
        print("%2d | %5d %5d %5d %5d %6d | %5d %5d %5d %5d %6d" % tuple([c] + row))


def view_lm(name_l, name_m):
    L, M = run_trace(name_l), run_trace(name_m)
    print("L build %.3f ms @%d MHz, M build %.3f ms @%d MHz" % (L["ms"], L["mhz"], M["ms"], M["mhz"]))
    tl, tm = L["t"], M["t"]
    print(" c   Lw_P  Lw_Q | M_P  M_Q  periodM")
    for c in range(1, len(chunk_names()) - 1):
        print("%2d  %5d %5d | %5d %5d %6d" % (c, tl[0][c][6] - tl[0][c - 1][4], tl[4][c][6] - tl[4][c - 1][4], tm[0][c][3] - tm[0][c][2],
                                              tm[4][c][3] - tm[4][c][2], tm[0][c + 1][2] - tm[0][c][2]))


if __name__ == "__main__":
    argv = sys.argv[1:]
    view = "chunks"
    if argv and argv[0] == "--view":
        view, argv = argv[1], argv[2:]
    {"chunks": view_chunks, "l": view_l, "m": view_m, "lm": view_lm}[view](*(argv or ["pptr"]))
