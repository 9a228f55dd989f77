"""Mean shader clock DURING the training kernels (s_memtime cycles / s_memrealtime 100 MHz ticks of workgroup 0), next to their
duration: is the HBM write stream of the saved activations / gradients paid in clock (power) or in stalls?
usage: [PNR_LIB_PATH=build/ab/libpnr_X.so] python tools/clk_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
dev = torch.device("cuda:0")
clk = torch.zeros(2, dtype=torch.int64, device=dev)
from panopticnerf_amd import _lib, make_network, ops, synthetic
net = make_network(NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32, precision="bf16")).to(dev).train()
R, N = 4096, 192
rays = synthetic.camera_rays()[::129][:R].contiguous().to(dev)
z = ops.stratified(rays, N)
desc, img = net.packed(1, dev)
desc = _lib.MlpDesc.from_buffer_copy(bytes(desc))           # pnr_mlp_desc.clk_probe: the forward kernels launched with THIS copy
desc.clk_probe[0] = clk.data_ptr() & 0xffffffff            # of the descriptor stamp their clock here (int32 fields: wrap)
desc.clk_probe[1] = clk.data_ptr() >> 32

_, img_b = net.packed_bwd(1, dev)
def run(tag, fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    c = clk.cpu().tolist()
    print(f"{tag:28s} {ms:7.3f} ms   shader clock {c[0] / max(c[1], 1) * 100:7.0f} MHz   ({c[0] / 1e6:.2f} Mcycles)")
    return r
raw, acts = run("training forward", lambda: ops.mlp_forward_train(desc, img, rays, z))
d_raw = torch.randn_like(raw)
run("data-gradient pass", lambda: ops.mlp_backward(desc, img_b, d_raw, acts, R, N))
run("inference forward", lambda: ops.mlp_forward(desc, img, rays, z))
