"""Synthetic KITTI-360-shaped inputs: there is no dataset in this
environment, so rays come from a pinhole camera with KITTI-360-like rectified intrinsics and
the 3D bbox prior from seeded random oriented boxes.  CPU tensors; callers move them."""
import math

import torch

KITTI_W, KITTI_H = 1408, 376
KITTI_F, KITTI_CX, KITTI_CY = 552.554261, 682.049453, 238.769549


# BASELINE.json `configs` as renderer / network config keys (SURVEY.md 8d "config mapping").  `bbox`: whether the 3D
# bbox prior is part of the workload.  C = 45 / K = 32 are this build's choices (the reference's are unverifiable,
# SURVEY.md 9 item 8); every consumer prints what it used.
BASELINE_CONFIGS = {
    1: dict(name="configs[0]: coarse-only 32 samples/ray, 4x128 MLP, appearance only",
            N_samples=32, N_importance=0, D=4, W=128, skips=[4], num_classes=0, num_instances=0, bbox=False),
    2: dict(name="configs[1]: coarse-only 64 samples/ray, 8x256 MLP, appearance only",
            N_samples=64, N_importance=0, D=8, W=256, skips=[4], num_classes=0, num_instances=0, bbox=False),
    3: dict(name="configs[2]: coarse+fine 64+128 (sample_pdf), 8x256 MLPs, appearance + depth",
            N_samples=64, N_importance=128, D=8, W=256, skips=[4], num_classes=0, num_instances=0, bbox=False),
    4: dict(name="configs[3]: + semantic head 45 (3D bbox prior + learned), panoptic logit compositing",
            N_samples=64, N_importance=128, D=8, W=256, skips=[4], num_classes=45, num_instances=0, bbox=True),
    5: dict(name="configs[4]: full panoptic (semantic 45 + instance 32 heads, 3D bbox prior)",
            N_samples=64, N_importance=128, D=8, W=256, skips=[4], num_classes=45, num_instances=32, bbox=True),
}


def baseline_cfg(n, **extra):
    """BASELINE config n (1..5) as an attribute-style cfg for make_network / make_renderer (+ overrides)."""
    from types import SimpleNamespace
    d = {k: v for k, v in BASELINE_CONFIGS[n].items() if k not in ("name", "bbox")}
    d.update(extra)
    return SimpleNamespace(**d)


def camera_rays(width=KITTI_W, height=KITTI_H, yaw=0.0, origin=(0.0, 1.55, 0.0), near=0.5, far=100.0):
    """(H*W, 8) rays: o(3), d(3) (unnormalised pixel directions, z forward), near, far."""
    j, i = torch.meshgrid(torch.arange(height, dtype=torch.float32), torch.arange(width, dtype=torch.float32),
                          indexing="ij")
    d = torch.stack([(i - KITTI_CX) / KITTI_F, (j - KITTI_CY) / KITTI_F, torch.ones_like(i)], -1).reshape(-1, 3)
    c, s = math.cos(yaw), math.sin(yaw)
    Rm = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
    d = d @ Rm.T
    o = torch.tensor(origin, dtype=torch.float32).expand_as(d)
    nf = torch.tensor([near, far], dtype=torch.float32).expand(d.shape[0], 2)
    return torch.cat([o, d, nf], -1).contiguous()


def random_boxes(n_box=64, n_sem=45, n_inst=32, seed=1):
    """Seeded oriented boxes: (M,15) = centre, rotation rows (yaw about y), half extents; ids (M,2) int32."""
    g = torch.Generator().manual_seed(seed)
    u = lambda lo, hi, *s: lo + (hi - lo) * torch.rand(*s, generator=g)
    ctr = torch.stack([u(-40, 40, n_box), u(-3, 3, n_box), u(2, 80, n_box)], -1)
    yaw = u(0, math.pi, n_box)
    c, s, z0, o1 = torch.cos(yaw), torch.sin(yaw), torch.zeros(n_box), torch.ones(n_box)
    rot = torch.stack([c, z0, s, z0, o1, z0, -s, z0, c], -1)
    ext = u(0.5, 4.0, n_box, 3)
    box = torch.cat([ctr, rot, ext], -1).float().contiguous()
    ids = torch.stack([torch.randint(0, max(n_sem, 1), (n_box,), generator=g),
                       torch.randint(0, max(n_inst, 1), (n_box,), generator=g)], -1).int().contiguous()
    return box, ids


def trained_like_(net, sigma_bias=0.03, seed=0):
    """Shift the density bias so a useful fraction of samples has alpha > 0 (a freshly
    initialised NeRF composites to almost nothing)."""
    with torch.no_grad():
        for n in (net.nerf_0, net.nerf_1):
            if n is not None:
                n.alpha_linear.bias.fill_(sigma_bias)
    return net
