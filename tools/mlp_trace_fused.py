"""Per-chunk s_memtime timeline of the FUSED ping-pong MLP launch (k_mlp_pp<.., FUSE>) -- every chunk of one sample
group, from a -DPNR_TRACE=1 -DPNR_TRACE_MASK=0x5c build (stamps 2 = M start, after the L->M barrier; 3 = M end, MFMAs issued;
4 = after the M->L barrier; 6 = L work done, before the L->M barrier).  Iteration PNR_TRACE_ITER of workgroup 0.  One line per chunk:
  period  = wave 0's M start of chunk c+1 - of chunk c  (what the chunk costs the workgroup: P's M + Q's M + 2 hand-overs)
  M_P/M_Q = M phase of group P (wave 0) / Q (wave 4);  bar_P/bar_Q = their wait at the M->L barrier
  Lw_P/Lw_Q = the L phase's own work (without the wait at the L->M barrier)
  L_P     = wave 0's L phase before this chunk (after-barrier of chunk c-1 -> M start of c), which runs beside Q's M(c-1)
usage: python tools/mlp_trace_fused.py <lib name under build/ab>"""
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

# chunk names of the 8x256 + 45/32 plan (pnr_mlp_plan.h order)
PLAN1 = os.environ.get("PNR_PLAN", "1") != "0"
names = ["trunk0"] * (1 if PLAN1 else 2)       # plan 1: layer 0 is one chunk
for l in range(1, 8):
    names += ["trunk%d" % l] * 4
if PLAN1:
    names += ["feature"] * 4 + ["views"] * 2 + ["rgbsigma"] + ["sem0"] * 2 + ["inst0"] * 2 + ["logits"]
else:
    names += ["feature"] * 4 + ["views"] * 2 + ["rgbsigma"] + ["sem0"] * 2 + ["sem1"] * 2 + ["inst0"] * 2 + ["inst1"]
lib = os.path.join(ROOT, "build", "ab", "libpnr_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "pptr"))
out = subprocess.run([sys.executable, __file__, "--child", lib], capture_output=True, text=True, timeout=240)
line = [l for l in out.stdout.splitlines() if l.startswith("TRACE ")]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:])
    sys.exit(1)
D = json.loads(line[0][6:])
T = D["t"]
n = len(names)
print("traced launch (trace build): %.3f ms at %.0f MHz; %d chunks" % (D["ms"], D["mhz"], n))
print("%3s %-9s %7s %6s %6s %6s %6s %6s %6s %6s" % ("c", "layer", "period", "M_P", "bar_P", "M_Q", "bar_Q", "L_P", "Lw_P", "Lw_Q"))
tot = 0
by = {}
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
