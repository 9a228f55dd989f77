"""Where a training step's time goes (diagnostic): per-stage GPU time of one level, 4096 rays x 192 samples."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import _wgrad_ref as wref      # library-GEMM cross-check of pnr_mlp_wgrad
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, make_renderer, ops, synthetic, train
dev = torch.device("cuda:0")
cfg = NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32, precision="bf16")
net = make_network(cfg).to(dev).train()
rend = make_renderer(cfg, net)
R, N = 4096, 192
rays = synthetic.camera_rays()[::129][:R].contiguous().to(dev)
z = ops.stratified(rays, N)
nerf = net.nerf(1)
def T(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
t, (desc, img) = T(lambda: (nerf.rgb_linear.bias.data.add_(0.0), net.packed(1, dev))[1]); print(f"pack fwd (device)      {t:8.3f} ms")
t, (_, img_b) = T(lambda: (nerf.rgb_linear.bias.data.add_(0.0), net.packed_bwd(1, dev))[1]); print(f"pack bwd (device)      {t:8.3f} ms")
t, (raw, acts) = T(lambda: ops.mlp_forward_train(desc, img, rays, z)); print(f"mlp_forward_train      {t:8.3f} ms")
t, out = T(lambda: ops.composite(raw, z, rays, 45, 32, True)); print(f"composite              {t:8.3f} ms")
g = {"rgb": torch.randn(R, 3, device=dev), "semantic": torch.randn(R, 45, device=dev)}
t, d_raw = T(lambda: ops.composite_backward(raw, z, rays, 45, 32, g)); print(f"composite_backward     {t:8.3f} ms")
t, dys = T(lambda: ops.mlp_backward(desc, img_b, d_raw, acts, R, N)); print(f"mlp_backward (dgrad)   {t:8.3f} ms")
t, wg = T(lambda: wref.weight_grads(nerf, desc, acts, dys, d_raw, R * N)); print(f"weight_grads (torch)   {t:8.3f} ms")
shapes = {k: v.shape for k, v in nerf.state_dict().items()}
t, wk = T(lambda: ops.mlp_wgrad(desc, acts, dys, R * N, shapes), 20); print(f"pnr_mlp_wgrad (HIP)    {t:8.3f} ms")
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
for p in net.parameters(): p.grad = torch.zeros_like(p)
t, _ = T(lambda: opt.step()); print(f"Adam.step              {t:8.3f} ms")
