"""Run the fused MLP forward (fine level, 65536 rays x 192 samples) a few times -- a target for rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, ops, synthetic
dev = torch.device("cuda:0")
net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
rays = synthetic.camera_rays()[:65536].to(dev)
z = ops.stratified(rays, 192)
desc, img = net.packed(1, dev)
raw = ops.alloc_raw(81, 65536 * 192, dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ops.mlp_forward(desc, img, rays, z, out=raw)
torch.cuda.synchronize()
