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


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("geom", [(2, 128, [], 0, 0), (3, 128, [1], 5, 3), (8, 256, [4], 45, 32), (8, 256, [4], 0, 0)])
def test_mlp_backward_matches_autograd(dev, geom):
    """pnr_mlp_forward_train + pnr_mlp_backward + weight-gradient GEMMs vs torch autograd through the
    bf16-emulating oracle MLP (same rounded activations => same ReLU gates; against the fp32 forward the
    gates of near-zero units differ and the error grows ~1.5 % per layer of depth, which says nothing about
    the kernels) for the same upstream d_raw.  Per-tensor relative L2 error < 3 %."""
    from panopticnerf_amd import make_network, train
    from types import SimpleNamespace as NS
    D, W, skips, C, K = geom
    torch.manual_seed(D * 7 + W + C)
    net = make_network(NS(D=D, W=W, skips=skips, num_classes=C, num_instances=K))
    nerf = net.nerf_0
    rng = np.random.default_rng(D + W)
    R, N = 7, 41                                     # 287 samples: ragged last tile and last group
    rays = torch.tensor(_rays(rng, R, 0.5, 8.0))
    z = torch.tensor(co.stratified(rays.numpy(), N, t_rand=rng.random((R, N)).astype(np.float32)))
    ocfg = to.mlp_config(D=D, W=W, skips=tuple(skips), n_sem=C, n_inst=K, head_W=W // 2)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in nerf.state_dict().items()}
    raw_ref = to.run_network(params, ocfg, rays, z, emulate_bf16=True)    # (R,N,ch)
    d_raw = torch.tensor(rng.normal(size=raw_ref.shape).astype(np.float32))
    (raw_ref * d_raw).sum().backward()
    desc, img = net.packed(0, dev, "bf16")
    raw, acts = ops.mlp_forward_train(desc, img, rays.to(dev), z.to(dev))
    assert (raw.T.reshape(R, N, -1).cpu() - raw_ref.detach()).abs().max() < 6e-2      # forward unchanged by saving
    d_cm = d_raw.reshape(R * N, -1).T.contiguous().to(dev)
    _, img_b = net.packed_bwd(0, dev)
    dys = ops.mlp_backward(desc, img_b, d_cm, acts, R, N)
    g = train.weight_grads(nerf, desc, acts, dys, d_cm, R * N)
    assert set(g) == set(params)
    errs = {k: _rel(g[k].cpu(), params[k].grad) for k in params}
    print("rel L2 errors:", {k: round(v, 4) for k, v in errs.items()})
    for k, v in errs.items():
        assert v < 3e-2, (k, v)


def test_render_backward_end_to_end(dev):
    """Renderer.render under autograd: loss on rgb / depth / semantic maps of both levels; parameter gradients vs
    torch autograd through the oracle's render_rays fed the HIP path's own z (identical stage inputs)."""
    from panopticnerf_amd import make_network, make_renderer
    from types import SimpleNamespace as NS
    C, K = 6, 4
    cfg = NS(N_samples=32, N_importance=32, num_classes=C, num_instances=K, precision="bf16", D=4, W=128, skips=[1])
    torch.manual_seed(3)
    net = make_network(cfg).to(dev).train()
    with torch.no_grad():
        for lv in (0, 1):
            net.nerf(lv).alpha_linear.bias.fill_(0.3)
    rend = make_renderer(cfg, net)
    rng = np.random.default_rng(5)
    R = 64
    rays = torch.tensor(_rays(rng, R, 0.5, 6.0))
    tgt = {k: torch.tensor(rng.normal(size=s).astype(np.float32)) for k, s in
           (("rgb", (R, 3)), ("depth", (R,)), ("semantic", (R, C)), ("instance", (R, K)))}
    out = rend.render({"rays": rays[None].to(dev)})
    loss = sum(((out[f"{k}_{lv}"][0] - v.to(dev)) ** 2).mean() for lv in (0, 1) for k, v in tgt.items())
    loss.backward()
    # oracle: same z per level, fp32 autograd
    ocfg = to.mlp_config(D=4, W=128, skips=(1,), n_sem=C, n_inst=K, head_W=64)
    ref_loss = 0
    ref_params = {}
    for lv in (0, 1):
        prm = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.nerf(lv).state_dict().items()}
        ref_params[lv] = prm
        zz = out[f"z_vals_{lv}"][0].detach().cpu()
        o = to.raw2outputs(to.run_network(prm, ocfg, rays, zz), zz, rays[:, 3:6], C, K)
        ref_loss = ref_loss + sum(((o[k] - v) ** 2).mean() for k, v in tgt.items())
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item())
    for lv in (0, 1):
        for name, p in net.nerf(lv).named_parameters():
            assert p.grad is not None, name
            r = _rel(p.grad.cpu(), ref_params[lv][name].grad)
            assert r < 5e-2, (lv, name, r)
    # an optimiser step changes the parameters and the next render repacks them
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    before = out["rgb_1"].detach().clone()
    opt.step()
    out2 = rend.render({"rays": rays[None].to(dev)})
    assert (out2["rgb_1"].detach() - before).abs().max() > 0


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_device_packer_equals_host_packer(dev, prec):
    """pnr_mlp_pack_device writes the same image, bit for bit, as the host packers (forward and transposed)."""
    from panopticnerf_amd import make_network
    from types import SimpleNamespace as NS
    torch.manual_seed(1)
    net = make_network(NS(num_classes=45, num_instances=32))
    nerf = net.nerf_0
    desc = nerf.desc(prec)
    sd_cpu = {k: v.detach() for k, v in nerf.state_dict().items()}
    sd_dev = {k: v.to(dev) for k, v in sd_cpu.items()}
    host = ops.pack_mlp(desc, sd_cpu)
    devi, ws = ops.pack_mlp_device(desc, sd_dev, False)
    assert torch.equal(devi.cpu(), host)
    if prec == "bf16":
        host_b = ops.pack_mlp_bwd(desc, sd_cpu)
        dev_b, _ = ops.pack_mlp_device(desc, sd_dev, True)
        assert torch.equal(dev_b.cpu(), host_b)
    # buffers are reused and the image follows parameter updates
    sd_dev["rgb_linear.bias"].add_(1.0)
    again, ws2 = ops.pack_mlp_device(desc, sd_dev, False, devi, ws)
    assert again.data_ptr() == devi.data_ptr() and ws2.data_ptr() == ws.data_ptr()
    sd_cpu["rgb_linear.bias"] = sd_cpu["rgb_linear.bias"] + 1.0
    assert torch.equal(again.cpu(), ops.pack_mlp(desc, sd_cpu))
