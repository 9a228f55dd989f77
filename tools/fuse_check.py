"""Diagnostic: is pnr_mlp_forward_composite independent of which other rays share the launch?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, ops, synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
C, K = 19, 8
net = make_network(NS(N_importance=128, num_classes=C, num_instances=K)).to(dev).eval()
synthetic.trained_like_(net, 0.05)
rays = synthetic.camera_rays()[:70000].contiguous().to(dev)
for N in (64, 192):
    z = ops.stratified(rays, N)
    desc, img = net.packed(1, dev, "bf16")
    g = torch.Generator(device=dev).manual_seed(1)
    hit = torch.rand(z.shape, device=dev, generator=g) < 0.3
    ls = torch.where(hit, torch.randint(0, C, z.shape, device=dev, generator=g), -1).int()
    li = torch.where(hit, torch.randint(0, K, z.shape, device=dev, generator=g), -1).int()
    full = ops.mlp_forward_composite(desc, img, rays, z, ls, li, False, True)
    full2 = ops.mlp_forward_composite(desc, img, rays, z, ls, li, False, True)
    idx = torch.arange(0, 70000, 187, device=dev)
    sub = ops.mlp_forward_composite(desc, img, rays[idx].contiguous(), z[idx].contiguous(), ls[idx].contiguous(), li[idx].contiguous(), False, True)
    for k in full:
        d_rep = (full[k] != full2[k]).reshape(full[k].shape[0], -1).any(1)
        d_sub = (full[k][idx] != sub[k]).reshape(idx.numel(), -1).any(1)
        print(f"N={N} {k:14s} run-to-run rays differing {int(d_rep.sum()):6d}   full vs subset rays differing {int(d_sub.sum()):5d} of {idx.numel()}"
              f"   max abs diff {float((full[k][idx] - sub[k]).abs().max()):.3e}")
