"""Host-side cost of the eager training step: cProfile over N steps of tools/train_trace.py's loop (no synchronisation inside the
loop), top functions by own time and by cumulative time, next to the wall time per step and the same step's HIP-graph replay
time from bench.py -- where the eager step is launch-bound, this says which Python is in the way.
usage: python tools/train_cpu_profile.py [steps=40]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import NetworkWrapper, make_network, synthetic
dev = torch.device("cuda:0")
cfg = NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32, precision="bf16")
net = make_network(cfg).to(dev).train(); synthetic.trained_like_(net)
wrap = NetworkWrapper(net, cfg)
opt = torch.optim.Adam(net.parameters(), lr=5e-4, fused=True)
g = torch.Generator(device=dev).manual_seed(0)
rays = synthetic.camera_rays().to(dev); box, ids = synthetic.random_boxes(64, 45, 32)
R = 4096
idx = torch.randint(0, rays.shape[0], (R,), generator=g, device=dev)
tb = {"rays": rays[idx][None].contiguous(), "bbox": box.to(dev), "bbox_ids": ids.to(dev),
      "rgb": torch.rand((1, R, 3), generator=g, device=dev), "depth": torch.rand((1, R), generator=g, device=dev) * 60 - 10,
      "pseudo_label": torch.randint(-1, 45, (1, R), generator=g, device=dev).int(), "instance_label": torch.randint(-1, 32, (1, R), generator=g, device=dev).int()}


def step():
    opt.zero_grad(set_to_none=True)
    _, loss, _, _ = wrap(tb); loss.backward(); opt.step()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
t_enqueue = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_wall = (time.perf_counter() - t0) / n
print("eager: %.3f ms per step wall, %.3f ms of host time to enqueue one (no sync inside the loop)" % (t_wall * 1e3, t_enqueue * 1e3))
# host time alone: the same loop with the GPU idle-waiting excluded is what cProfile sees below (it adds its own ~30 % overhead)
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(28)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
