"""Which torch ops (not libpnr kernels) a training step launches, and from which source line: torch.profiler over a few steps of
tools/train_trace.py's loop, GPU kernels grouped by name with the Python frame that issued them.  The step's glue (fills, copies,
small elementwise kernels) is ~0.2 ms of ~8: this is the tool that says where each comes from.
usage: python tools/train_glue_profile.py [steps=4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
from torch.profiler import profile, ProfilerActivity
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


def step():
    opt.zero_grad(set_to_none=True)
    _, loss, _, _ = wrap(tb); loss.backward(); opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(n):
        step()
    torch.cuda.synchronize()
ev = prof.events()
# ATen ops that own device time, by (op, shapes, innermost repo frame)
rows = collections.defaultdict(lambda: [0, 0.0])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for e in ev:
    dt = getattr(e, "device_time_total", None)
    if dt is None:
        dt = getattr(e, "cuda_time_total", 0)
    if not dt or not e.name.startswith("aten::") or e.cpu_parent is not None and e.cpu_parent.name.startswith("aten::"):
        continue
    frame = next((f for f in (e.stack or []) if root in f and "tools/" not in f), (e.stack or ["?"])[0] if e.stack else "?")
    key = (e.name, str(e.input_shapes)[:60], frame.replace(root + "/", "")[:110])
    rows[key][0] += 1
    rows[key][1] += dt
tot = sum(v[1] for v in rows.values())
print("torch ops with device time, per step (%d steps): %.1f us in all" % (n, tot / n))
for (name, shp, frame), (cnt, dt) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%7.1f us %5.1f x  %-28s %-60s %s" % (dt / n, cnt / n, name, shp, frame))
