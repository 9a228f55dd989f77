"""Per-chunk timeline of k_mlp_bwd (and of the training forward) from a -DPNR_TRACE=1 build: s_memtime stamps of workgroup 0,
iteration 2.  Stamps: 0 chunk top | 1 DMA issued | 2 before the MFMA loop | 3 MFMAs issued | 4 epilogue + stores issued |
5 after the vmcnt wait (arrival at the barrier) | 6 barrier released.  Every stamp costs an lgkmcnt(0) wait: a rough picture.
usage: PNR_LIB_PATH=build/ab/libpnr_trb.so python tools/bwd_trace.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace as NS
NCH, NST = 48, 8
dev = torch.device("cuda:0")
trace = torch.zeros((8, NCH, NST), dtype=torch.int64, device=dev)
os.environ["PNR_TRACE_PTR"] = str(trace.data_ptr())
from panopticnerf_amd import make_network, ops, synthetic
net = make_network(NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32, precision="bf16")).to(dev).train()
R, N = 4096, 192
rays = synthetic.camera_rays()[::129][:R].contiguous().to(dev)
z = ops.stratified(rays, N)
desc, img = net.packed(1, dev)
_, img_b = net.packed_bwd(1, dev)
def show(tag, nchunks):
    T = trace.cpu().tolist()
    print("==", tag)
    print("chunk  period |  wave0: dma_issued mma_start mma_end epi_end arrive release | max over waves: arrive   (cycles after this chunk's top)")
    for c in range(1, nchunks):
        w0 = T[0][c]
        if w0[0] == 0: break
        per = w0[0] - T[0][c - 1][0]
        arr = max(T[w][c][5] - T[w][c][0] for w in range(8))
        print(f"{c:5d} {per:7d} | " + " ".join(f"{w0[k] - w0[0]:8d}" for k in (1, 2, 3, 4, 5, 6)) + f" | {arr:8d}")
raw, acts = ops.mlp_forward_train(desc, img, rays, z)
torch.cuda.synchronize()
show("training forward (k_mlp_fused<TRAIN>)", NCH)
trace.zero_()
d_raw = torch.randn_like(raw)
dys = ops.mlp_backward(desc, img_b, d_raw, acts, R, N)
torch.cuda.synchronize()
show("data-gradient pass (k_mlp_bwd)", NCH)
