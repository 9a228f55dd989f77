"""A/B timing of the fused-MLP kernel variants (PNR_MLP_VARIANT) on the bench's dominant launch:
fine level, 65536 rays x 192 samples, 8x256 + semantic(45) + instance(32) heads, bf16.
Each variant runs in its own process (the variant is latched at first use)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
import bench
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, ops, synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
synthetic.trained_like_(net)
rays = synthetic.camera_rays()[:65536].to(dev)
z = ops.stratified(rays, 192)
desc, img = net.packed(1, dev)
raw = torch.empty((81, 65536 * 192), device=dev)
ops.time_mlp_forward(desc, img, rays, z, raw, 2)
ms = min(ops.time_mlp_forward(desc, img, rays, z, raw, 5) for _ in range(3))
fl = 65536 * 192 * bench.mlp_flops_per_sample()
print("variant %%s: %%.3f ms  %%.1f TFLOP/s  %%.1f Msamples/s  checksum %%.6f" %% (
    sys.argv[1], ms, fl / ms / 1e9, 65536 * 192 / ms / 1e3, raw[:, ::100003].double().sum().item()))
''' % ROOT

for v in (sys.argv[1:] or ["0", "1", "2", "3"]):
    env = dict(os.environ, PNR_MLP_VARIANT=v)
    subprocess.run([sys.executable, "-c", CHILD, v], env=env, check=False)
