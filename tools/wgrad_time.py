"""pnr_mlp_wgrad alone at the benched training geometry (4096 rays x 192 samples, 8x256 + 45/32 heads): hipEvent time of
20 launches, 6 repeats -> min / median.  A/B two builds with PNR_LIB_PATH (one process per build, interleaved by the caller)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, ops, synthetic
dev = torch.device("cuda:0")
cfg = NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32, precision="bf16")
net = make_network(cfg).to(dev).train()
R, N = 4096, 192
rays = synthetic.camera_rays()[::129][:R].contiguous().to(dev)
z = ops.stratified(rays, N)
nerf = net.nerf(1)
desc, img = net.packed(1, dev)
_, img_b = net.packed_bwd(1, dev)
raw, acts = ops.mlp_forward_train(desc, img, rays, z)
d_raw = torch.randn_like(raw)
dys = ops.mlp_backward(desc, img_b, d_raw, acts, R, N)
shapes = {k: v.shape for k, v in nerf.state_dict().items()}
fn = lambda: ops.mlp_wgrad(desc, acts, dys, R * N, shapes)
for _ in range(3): fn()
ts = []
for rep in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20)
ts.sort()
print(f"{os.path.basename(os.environ.get('PNR_LIB_PATH', 'libpnr.so')):24s} wgrad min {ts[0]:.4f} median {ts[3]:.4f} ms")
