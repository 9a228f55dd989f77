"""Kernel-level view of the training step: run a few steps (NetworkWrapper + backward + Adam) as the target of
`rocprofv3 --kernel-trace`; tools/prof_summary.py turns the trace into the per-kernel table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import NetworkWrapper, make_network, synthetic
dev = torch.device("cuda:0")
cfg = NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32, precision="bf16")
net = make_network(cfg).to(dev).train(); synthetic.trained_like_(net)
wrap = NetworkWrapper(net, cfg)
opt = torch.optim.Adam(net.parameters(), lr=5e-4, fused=True)       # as bench.py's training step
g = torch.Generator(device=dev).manual_seed(0)
rays = synthetic.camera_rays().to(dev); box, ids = synthetic.random_boxes(64, 45, 32)
R = 4096
idx = torch.randint(0, rays.shape[0], (R,), generator=g, device=dev)
tb = {"rays": rays[idx][None].contiguous(), "bbox": box.to(dev), "bbox_ids": ids.to(dev),
      "rgb": torch.rand((1, R, 3), generator=g, device=dev), "depth": torch.rand((1, R), generator=g, device=dev) * 60 - 10,
      "pseudo_label": torch.randint(-1, 45, (1, R), generator=g, device=dev), "instance_label": torch.randint(-1, 32, (1, R), generator=g, device=dev)}
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    opt.zero_grad(set_to_none=True)
    _, loss, _, _ = wrap(tb); loss.backward(); opt.step()
torch.cuda.synchronize()
