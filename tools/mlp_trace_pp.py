"""Per-chunk s_memtime timeline of the ping-pong fused MLP (k_mlp_pp) from a -DPNR_TRACE=1 -DPNR_TRACE_MASK=0x3f build:
stamps 0 refill start (L, after the epilogue) | 1 refill issued | 2 M start (after the L->M barrier) | 3 M end (MFMAs
issued) | 4 after the M->L barrier | 5 after vmcnt(0).  Iteration 2 of workgroup 0; cycles relative to wave 0's stamp 2 of
the same chunk.  usage: PNR_MLP_VARIANT=1 python tools/mlp_trace_pp.py <lib name under build/ab>"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCH, NST = 48, 8
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    os.environ["PNR_LIB_PATH"] = sys.argv[2]
    os.environ["PNR_MLP_VARIANT"] = "1"
    import torch
    from types import SimpleNamespace as NS
    dev = torch.device("cuda:0")
    trace = torch.zeros((8, NCH, NST), dtype=torch.int64, device=dev)
    os.environ["PNR_TRACE_PTR"] = str(trace.data_ptr())
    from panopticnerf_amd import make_network, ops, synthetic
    net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
    rays = synthetic.camera_rays()[:65536].to(dev)
    z = ops.stratified(rays, 192)
    desc, img = net.packed(1, dev)
    raw = ops.alloc_raw(81, 65536 * 192, dev)
    ops.mlp_forward(desc, img, rays, z, out=raw)
    torch.cuda.synchronize()
    print("TRACE " + json.dumps(trace.cpu().tolist()))
    sys.exit(0)
lib = os.path.join(ROOT, "build", "ab", "libpnr_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "pptr"))
out = subprocess.run([sys.executable, __file__, "--child", lib], capture_output=True, text=True, timeout=180)
line = [l for l in out.stdout.splitlines() if l.startswith("TRACE ")]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:])
    sys.exit(1)
T = json.loads(line[0][6:])
names = ["refill0", "refill1", "M_start", "M_end", "afterB", "vmcnt0"]
for c in range(6, 16):
    base = T[0][c][2]
    print(f"chunk {c}: cycles relative to wave 0's M start; period(wave0 M start -> next chunk's) = {T[0][c + 1][2] - base}")
    for w in range(8):
        print(f"   wave {w}  " + "  ".join(f"{names[k]}={T[w][c][k] - base:6d}" for k in range(6)) +
              f"   M={T[w][c][3] - T[w][c][2]:5d}  L={T[w][c][2] - T[w][c - 1][4]:5d}")
