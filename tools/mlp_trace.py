"""Per-chunk cycle breakdown of one wave of the fused MLP (PNR_TRACE build, tools/build_ab.sh trace:-DPNR_TRACE=1).
Stamps per chunk: 0 chunk start, 1 after DMA issue, 2 after the chunk's MFMAs are issued, 3 before vmcnt wait,
4 after vmcnt wait, 5 after barrier."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PNR_LIB_PATH"] = os.path.join(ROOT, "build", "ab", "libpnr_trace.so")
os.environ.setdefault("PNR_MLP_VARIANT", "3")
import torch
from types import SimpleNamespace as NS
dev = torch.device("cuda:0")
n_chunks = 88
trace = torch.zeros((4, n_chunks, 8), dtype=torch.int64, device=dev)
os.environ["PNR_TRACE_PTR"] = str(trace.data_ptr())
os.environ["PNR_TRACE_WAVE"] = sys.argv[1] if len(sys.argv) > 1 else "0"
from panopticnerf_amd import make_network, ops, synthetic
net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
rays = synthetic.camera_rays()[:65536].to(dev)
z = ops.stratified(rays, 192)
desc, img = net.packed(1, dev)
raw = torch.empty((81, 65536 * 192), device=dev)
ops.mlp_forward(desc, img, rays, z, out=raw)
torch.cuda.synchronize()
t = trace.cpu().numpy()
it = 2
T = t[it]
print("iteration", it, "total cycles", int(T[-1, 5] - T[0, 0]), " (stamps are s_memtime = shader clock? 100MHz ref?)")
print("chunk  refill(WAR+vmcnt+DMA)  RAWwait+mma_issue  epilogue  advance   total")
tot = [0] * 5
for c in range(n_chunks):
    d = [int(T[c, 1] - T[c, 0]), int(T[c, 2] - T[c, 1]), int(T[c, 3] - T[c, 2]), int(T[c, 5] - T[c, 3])]
    nxt = int((T[c + 1, 0] if c + 1 < n_chunks else T[c, 5]) - T[c, 5])
    full = int(T[c, 5] - T[c, 0]) + nxt
    for i, v in enumerate(d + [nxt]):
        tot[i] += v
    if c < 26 or c > 60:
        print(f"{c:5d} {d[0]:14d} {d[1]:18d} {d[2]:9d} {d[3]:8d} {full:8d}  (+{nxt} to next chunk)")
print("sum    ", tot, "=", sum(tot))
