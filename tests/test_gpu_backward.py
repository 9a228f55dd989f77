"""GPU parity tests of the backward path: HIP kernels vs torch autograd
through the oracle's fp32 restatement, on identical inputs."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import torch_oracle as to
from panopticnerf_amd import ops

pytestmark = pytest.mark.gpu


def _rays(rng, R, near=0.5, far=30.0):
    o = rng.normal(0, 1, (R, 3)) + np.array([0, 1.5, 0])
    d = rng.normal(0, 0.3, (R, 3)) + np.array([0, 0, 1.0])
    return np.concatenate([o, d, np.full((R, 1), near), np.full((R, 1), far)], 1).astype(np.float32)


@pytest.mark.parametrize("N", [4, 32, 64, 192, 256])
def test_composite_backward_matches_autograd(dev, N):
    rng = np.random.default_rng(N)
    R, C, K = 53, 6, 5
    rays = torch.tensor(_rays(rng, R))
    z = torch.tensor(co.stratified(rays.numpy(), N, t_rand=rng.random((R, N)).astype(np.float32)))
    raw = torch.tensor(rng.normal(0, 1, (R, N, 4 + C + K)).astype(np.float32))
    raw[..., 3] = torch.tensor(rng.normal(0.05, 0.15, (R, N)).astype(np.float32))
    noise = torch.tensor(rng.normal(0, 0.02, (R, N)).astype(np.float32))
    g = {"rgb": torch.tensor(rng.normal(size=(R, 3)).astype(np.float32)),
         "depth": torch.tensor(rng.normal(size=R).astype(np.float32)) * 0.1,
         "acc": torch.tensor(rng.normal(size=R).astype(np.float32)),
         "semantic": torch.tensor(rng.normal(size=(R, C)).astype(np.float32)),
         "instance": torch.tensor(rng.normal(size=(R, K)).astype(np.float32)),
         "weights": torch.tensor(rng.normal(size=(R, N)).astype(np.float32))}
    rr = raw.clone().requires_grad_(True)
    out = to.raw2outputs(rr, z, rays[:, 3:6], C, K, noise)
    loss = sum((out[k] * g[k]).sum() for k in g)
    loss.backward()
    ref = rr.grad.reshape(R * N, -1).T.contiguous()          # channel-major
    raw_cm = raw.reshape(R * N, -1).T.contiguous().to(dev)
    d_raw = ops.composite_backward(raw_cm, z.to(dev), rays.to(dev), C, K, {k: v.to(dev) for k, v in g.items()},
                                   noise=noise.to(dev))
    err = (d_raw.cpu() - ref).abs()
    scale = ref.abs().max().item()
    assert err.max().item() < 2e-4 * max(scale, 1.0), (err.max().item(), scale)
    # a subset of the upstream gradients (others NULL) and no noise
    rr = raw.clone().requires_grad_(True)
    out = to.raw2outputs(rr, z, rays[:, 3:6], C, K)
    ((out["rgb"] * g["rgb"]).sum() + (out["semantic"] * g["semantic"]).sum()).backward()
    d2 = ops.composite_backward(raw_cm, z.to(dev), rays.to(dev), C, K, {"rgb": g["rgb"].to(dev), "semantic": g["semantic"].to(dev)})
    ref2 = rr.grad.reshape(R * N, -1).T
    assert (d2.cpu() - ref2).abs().max().item() < 2e-4 * max(ref2.abs().max().item(), 1.0)
