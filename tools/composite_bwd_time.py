"""pnr_composite_backward alone at the training step's geometry (4096 rays, 45 + 32 logit fields, bbox labels, every upstream
gradient and the 3D cross-entropy scales present), fine (N = 192) and coarse (N = 64) level; hipEvents, interleaved repeats; a
checksum of d_raw per build so that A/B builds (PNR_LIB_PATH=build/ab/libpnr_<name>.so) can be compared for bit identity.
usage: [PNR_LIB_PATH=...] python tools/composite_bwd_time.py [rays=4096]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panopticnerf_amd import ops, synthetic

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
C, K = 45, 32
g = torch.Generator(device=dev).manual_seed(0)
rays = synthetic.camera_rays().to(dev)
rays = rays[torch.randint(0, rays.shape[0], (R,), generator=g, device=dev)].contiguous()
box, ids = synthetic.random_boxes(64, C, K)
hits = ops.bbox_hits(rays, box.to(dev), 8)


def timed(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for N in (192, 64):
    z = ops.stratified(rays, N)
    ls, li = ops.sample_labels(z, hits[0], hits[1], hits[2], ids.to(dev))
    raw = torch.randn((4 + C + K, R * N), generator=g, device=dev)
    grads = {"rgb": torch.randn((R, 3), generator=g, device=dev), "depth": torch.randn((R,), generator=g, device=dev),
             "acc": torch.randn((R,), generator=g, device=dev), "semantic": torch.randn((R, C), generator=g, device=dev),
             "instance": torch.randn((R, K), generator=g, device=dev), "weights": torch.randn((R, N), generator=g, device=dev),
             "fix_semantic": torch.randn((R, C), generator=g, device=dev), "fix_instance": torch.randn((R, K), generator=g, device=dev)}
    ce = torch.full((1,), 1e-4, device=dev)
    run = lambda: ops.composite_backward(raw, z, rays, C, K, grads, None, ls, li, ce, ce, 0)
    ts = [timed(run) for _ in range(4)]
    d = run()
    lab = float(((ls >= 0).any(1)).float().mean())
    print("N=%3d  %s us  best %.1f   rays with a label %.2f   d_raw checksum %s" %
          (N, " ".join("%.1f" % t for t in ts), min(ts), lab, hex(int(d.view(torch.int32).to(torch.int64).sum().item()) & 0xffffffffffff)), flush=True)
