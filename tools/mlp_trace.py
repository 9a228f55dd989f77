"""Low-intrusion per-chunk timeline of the fused MLP.  A stamp (s_memtime) costs an lgkmcnt(0) wait, so every build
carries only three: chunk top (0), barrier release (6) and ONE more (k); the barrier release is common to all waves and
aligns them.  Build:  for k in 1 2 3 4 5: tools/build_ab.sh tr$k:"-DPNR_TRACE=1 -DPNR_TRACE_MASK=$((1|64|1<<k))"
Stamps: 1 after the DMA issue | 2 before the MFMA loop | 3 after the MFMA loop (issued) | 4 chunk end | 5 after vmcnt(0)
(arrival at the barrier).  Iteration 2 of workgroup 0.  usage: python tools/mlp_trace.py [prefix]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NCH, NST, N = 48, 8, 40
if len(sys.argv) > 2 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    os.environ["PNR_LIB_PATH"] = sys.argv[2]
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
prefix = sys.argv[1] if len(sys.argv) > 1 else "tr"
names = {2: "mma_start", 3: "mma_end", 4: "chunk_end", 5: "arrive", 1: "dma_issued"}
T = {}
for k in names:
    lib = os.path.join(ROOT, "build", "ab", f"libpnr_{prefix}{k}.so")
    if not os.path.exists(lib):
        continue
    out = subprocess.run([sys.executable, __file__, "--child", lib], capture_output=True, text=True, timeout=120).stdout
    line = [l for l in out.splitlines() if l.startswith("TRACE ")]
    if line:
        T[k] = json.loads(line[0][6:])
ks = [k for k in (1, 2, 3, 4, 5) if k in T]
k0 = ks[-1]
for c in (6, 7, 8, 9, 10, 11, 12, 13):
    print(f"chunk {c}: cycles relative to the release of this chunk's barrier (stamp 6)")
    for w in range(8):
        row = [f"top={T[k0][w][c][0] - T[k0][w][c][6]:6d}"]
        row += [f"{names[k]}={T[k][w][c][k] - T[k][w][c][6]:6d}" for k in ks]
        row.append(f"period={T[k0][w][c][6] - T[k0][w][c - 1][6]:5d}")
        print(f"   wave {w}  " + "  ".join(row))
