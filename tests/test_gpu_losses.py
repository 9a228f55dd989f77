"""GPU parity tests of the loss wrapper: pnr_losses / pnr_ce3d values and gradients against the
torch oracle (autograd), the fixed-field and 3D cross-entropy gradients through pnr_composite_backward2, and the
NetworkWrapper end to end.  The oracle is parity-unpinned (no reference code in the mount), see oracle/torch_oracle.py."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import torch_oracle as to
from panopticnerf_amd import ops

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.mark.parametrize("R,C,K,l2", [(1, 3, 2, False), (257, 45, 32, False), (1000, 19, 0, True), (4096, 45, 32, True)])
def test_losses_match_oracle(dev, R, C, K, l2):
    g = torch.Generator().manual_seed(R + C)
    maps = {"rgb": torch.rand(R, 3, generator=g), "depth": torch.rand(R, generator=g) * 20}
    tg = {"rgb": torch.rand(R, 3, generator=g), "depth": torch.rand(R, generator=g) * 20 - 4}      # ~20 % invalid (<= 0)
    if C:
        maps["semantic"] = torch.randn(R, C, generator=g) * 3
        maps["fix_semantic"] = torch.softmax(torch.randn(R, C, generator=g), 1) * torch.rand(R, 1, generator=g)
        tg["semantic"] = torch.randint(-1, C, (R,), generator=g, dtype=torch.int32)
    if K:
        maps["instance"] = torch.randn(R, K, generator=g) * 3
        maps["fix_instance"] = torch.softmax(torch.randn(R, K, generator=g), 1) * torch.rand(R, 1, generator=g)
        tg["instance"] = torch.randint(-1, K, (R,), generator=g, dtype=torch.int32)
    w = {"rgb": 1.0, "depth": 0.1, "semantic": 0.7, "fix_semantic": 0.3, "instance": 0.5, "fix_instance": 0.2}
    leaf = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
    terms, total = to.losses(leaf, tg, w, C, K, l2, 1e-5)
    total.backward()
    out, grads = ops.losses(w, {k: v.to(dev) for k, v in maps.items()}, {k: v.to(dev) for k, v in tg.items()}, C, K, l2, 1e-5)
    out = out.cpu()
    order = ("rgb", "depth", "semantic", "fix_semantic", "instance", "fix_instance")
    for i, k in enumerate(order):
        if k in terms:
            assert abs(out[i].item() - terms[k].item()) <= 2e-5 * max(1.0, abs(terms[k].item())), (k, out[i].item(), terms[k].item())
    assert abs(out[6].item() - total.item()) <= 2e-5 * max(1.0, abs(total.item()))
    for k in maps:
        ref = leaf[k].grad if leaf[k].grad is not None else torch.zeros_like(maps[k])
        assert (grads[k].cpu() - ref).abs().max() <= 1e-6 + 2e-5 * ref.abs().max(), k
    out2, _ = ops.losses(w, {k: v.to(dev) for k, v in maps.items()}, {k: v.to(dev) for k, v in tg.items()}, C, K, l2, 1e-5)
    assert torch.equal(out2.cpu(), out)                      # fixed-order reductions: deterministic


def test_losses_all_ignored_and_partial_inputs(dev):
    R, C = 300, 7
    maps = {"rgb": torch.rand(R, 3), "semantic": torch.randn(R, C)}
    tg = {"rgb": torch.rand(R, 3), "semantic": torch.full((R,), -1, dtype=torch.int32)}
    out, grads = ops.losses({"rgb": 2.0, "semantic": 1.0}, {k: v.to(dev) for k, v in maps.items()},
                            {k: v.to(dev) for k, v in tg.items()}, C, 0)
    assert out[2].item() == 0.0 and grads["semantic"].abs().max().item() == 0.0      # no labelled ray: zero term, zero grad
    assert abs(out[6].item() - 2.0 * ((maps["rgb"] - tg["rgb"]) ** 2).mean().item()) < 1e-5
    assert set(grads) == {"rgb", "semantic"}


def _level_inputs(R, N, C, K, seed):
    g = torch.Generator().manual_seed(seed)
    raw = torch.randn(R, N, 4 + C + K, generator=g)
    raw[..., 3] = raw[..., 3] * 0.5 + 0.2
    z = torch.sort(torch.rand(R, N, generator=g) * 5 + 0.5, 1).values
    rays = torch.cat([torch.zeros(R, 3), torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=1) * 1.3,
                      torch.full((R, 1), 0.5), torch.full((R, 1), 6.0)], 1)
    ls = torch.randint(-1, C, (R, N), generator=g, dtype=torch.int32)
    li = torch.randint(-1, K, (R, N), generator=g, dtype=torch.int32)
    return raw, z, rays, ls, li


@pytest.mark.parametrize("N", [16, 64, 192])
def test_ce3d_and_fixed_field_gradients_through_compositing(dev, N):
    """d_raw of pnr_composite_backward2 with fixed-field map gradients and the per-sample 3D CE vs torch autograd of
    the oracle's raw2outputs + ce3d on the same raw."""
    R, C, K = 37, 6, 4
    raw, z, rays, ls, li = _level_inputs(R, N, C, K, N)
    g = torch.Generator().manual_seed(1)
    up = {"rgb": torch.randn(R, 3, generator=g), "depth": torch.randn(R, generator=g), "semantic": torch.randn(R, C, generator=g),
          "fix_semantic": torch.randn(R, C, generator=g), "fix_instance": torch.randn(R, K, generator=g)}
    w3s, w3i = 0.8, -0.3
    leaf = raw.clone().requires_grad_(True)
    o = to.raw2outputs(leaf, z, rays[:, 3:6], C, K, label_sem=ls, label_inst=li)
    ces, ns = to.ce3d(leaf[..., 4:4 + C].reshape(-1, C), ls.reshape(-1))
    cei, ni = to.ce3d(leaf[..., 4 + C:].reshape(-1, K), li.reshape(-1))
    (sum((o[k] * v).sum() for k, v in up.items()) + w3s * ces + w3i * cei).backward()
    raw_cm = raw.reshape(R * N, -1).T.contiguous().to(dev)
    got_s, got_i = ops.ce3d(raw_cm, 4, C, ls.to(dev)).cpu(), ops.ce3d(raw_cm, 4 + C, K, li.to(dev)).cpu()
    assert abs(got_s[0].item() - ces.item()) < 2e-5 * max(1, abs(ces.item())) and int(got_s[1].item()) == ns
    assert abs(got_i[0].item() - cei.item()) < 2e-5 * max(1, abs(cei.item())) and int(got_i[1].item()) == ni
    d = ops.composite_backward(raw_cm, z.to(dev), rays.to(dev), C, K, {k: v.to(dev) for k, v in up.items()}, None, ls.to(dev),
                               li.to(dev), torch.tensor([w3s / max(ns, 1)], device=dev), torch.tensor([w3i / max(ni, 1)], device=dev))
    ref = leaf.grad.reshape(R * N, -1).T
    assert _rel(d.cpu(), ref) < 2e-4, _rel(d.cpu(), ref)
    # without the extra sources the old entry point's result is unchanged
    base = {k: v for k, v in up.items() if not k.startswith("fix_")}
    d0 = ops.composite_backward(raw_cm, z.to(dev), rays.to(dev), C, K, {k: v.to(dev) for k, v in base.items()})
    d1 = ops.composite_backward(raw_cm, z.to(dev), rays.to(dev), C, K, {k: v.to(dev) for k, v in base.items()}, None, ls.to(dev), li.to(dev))
    assert torch.equal(d0, d1)


def test_network_wrapper_end_to_end(dev):
    """NetworkWrapper(batch): loss value and every parameter gradient vs the oracle (same z per level, bf16-emulating
    oracle forward, as in test_mlp_backward_matches_autograd)."""
    from panopticnerf_amd import NetworkWrapper, make_network, synthetic
    C, K = 6, 4
    cfg = NS(N_samples=32, N_importance=32, num_classes=C, num_instances=K, precision="bf16", D=4, W=128, skips=[1],
             w_rgb=1.0, w_depth=0.05, w_sem=0.5, w_fix_sem=0.25, w_inst=0.5, w_fix_inst=0.25, w_sem3d=0.2, w_inst3d=0.1)
    torch.manual_seed(11)
    net = make_network(cfg).to(dev).train()
    with torch.no_grad():
        for lv in (0, 1):
            net.nerf(lv).alpha_linear.bias.fill_(0.2)
    wrap = NetworkWrapper(net, cfg)
    R = 96
    rays = synthetic.camera_rays()[::5519][:R].contiguous()
    box, ids = synthetic.random_boxes(24, C, K, seed=3)
    g = torch.Generator().manual_seed(2)
    batch = {"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev),
             "rgb": torch.rand(1, R, 3, generator=g).to(dev), "depth": (torch.rand(1, R, generator=g) * 30 - 5).to(dev),
             "pseudo_label": torch.randint(-1, C, (1, R), generator=g).to(dev), "instance_label": torch.randint(-1, K, (1, R), generator=g).to(dev)}
    ret, loss, stats, _ = wrap(batch)
    loss.backward()
    assert "fix_semantic_1" in ret and "ce3d_semantic_1" in ret and torch.isfinite(loss)
    # oracle on the same z, labels from the oracle's own bbox code
    ocfg = to.mlp_config(D=4, W=128, skips=(1,), n_sem=C, n_inst=K, head_W=64)
    w = wrap.weights
    tg = {"rgb": batch["rgb"][0].cpu(), "depth": batch["depth"][0].cpu(), "semantic": batch["pseudo_label"][0].cpu().int(),
          "instance": batch["instance_label"][0].cpu().int()}
    hits = to.bbox_hits(rays, box, 8)
    ref_total = 0
    ref_params = {}
    for lv in (0, 1):
        prm = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.nerf(lv).state_dict().items()}
        ref_params[lv] = prm
        zz = ret[f"z_vals_{lv}"][0].detach().cpu()
        ls, li = to.sample_labels(zz, hits[0], hits[1], hits[2], ids)
        raw = to.run_network(prm, ocfg, rays, zz, emulate_bf16=True)     # same rounded activations => same ReLU gates
        o = to.raw2outputs(raw, zz, rays[:, 3:6], C, K, label_sem=ls, label_inst=li)
        _, tot = to.losses({k: o[k] for k in ("rgb", "depth", "semantic", "fix_semantic", "instance", "fix_instance")}, tg, w, C, K)
        ces, _ = to.ce3d(raw[..., 4:4 + C].reshape(-1, C), ls.reshape(-1))
        cei, _ = to.ce3d(raw[..., 4 + C:].reshape(-1, K), li.reshape(-1))
        ref_total = ref_total + tot + 0.2 * ces + 0.1 * cei
    ref_total.backward()
    assert abs(loss.item() - ref_total.item()) < 3e-2 * abs(ref_total.item()), (loss.item(), ref_total.item())
    for lv in (0, 1):
        for name, p in net.nerf(lv).named_parameters():
            assert p.grad is not None, name
            r = _rel(p.grad.cpu(), ref_params[lv][name].grad)
            assert r < 6e-2, (lv, name, r)
    assert set(stats) >= {"loss", "rgb_loss_0", "fix_semantic_loss_1", "ce3d_semantic_loss_1"}


def test_chunked_training_render_weights_the_3d_ce_by_labelled_samples(dev):
    """the per-level 3D cross-entropy of a render split into ray chunks must be the mean over ALL labelled
    samples of the batch (chunk means weighted by their labelled-sample counts), not the mean of chunk means -- chunks
    see very different numbers of in-box samples.  One chunk vs ragged chunks: same scalars, same loss, same gradients."""
    from panopticnerf_amd import NetworkWrapper, make_network, synthetic
    C, K = 5, 3
    base = dict(N_samples=32, N_importance=32, num_classes=C, num_instances=K, precision="bf16", D=4, W=128, skips=[1],
                w_sem3d=0.3, w_inst3d=0.2)
    torch.manual_seed(5)
    net = make_network(NS(**base)).to(dev).train()
    with torch.no_grad():
        for lv in (0, 1):
            net.nerf(lv).alpha_linear.bias.fill_(0.2)
    R = 300
    rays = synthetic.camera_rays()[::1733][:R].contiguous()
    box, ids = synthetic.random_boxes(12, C, K, seed=4)
    box[:6, 0] -= 60.0                                   # half of the boxes far left: in-box samples concentrate in some rays
    g = torch.Generator().manual_seed(3)
    batch = {"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev), "rgb": torch.rand(1, R, 3, generator=g).to(dev)}
    res = {}
    for chunk in (4096, 77):
        wrap = NetworkWrapper(net, NS(chunk_size=chunk, **base))
        net.zero_grad(set_to_none=True)
        ret, loss, stats, _ = wrap(batch)
        loss.backward()
        res[chunk] = (ret, loss.detach(), {n: p.grad.clone() for n, p in net.named_parameters()})
    one, many = res[4096], res[77]
    for lv in (0, 1):
        for f in ("semantic", "instance"):
            k, kn = f"ce3d_{f}_{lv}", f"ce3d_{f}_n_{lv}"
            assert float(one[0][kn]) > 0 and float(one[0][kn]) == float(many[0][kn])
            assert abs(float(one[0][k].detach()) - float(many[0][k].detach())) < 2e-5 * abs(float(one[0][k].detach())), k
    assert abs(float(one[1]) - float(many[1])) < 1e-5 * abs(float(one[1]))
    for n in one[2]:
        assert _rel(many[2][n], one[2][n]) < 2e-3, n


@pytest.mark.parametrize("N", [16, 192])
def test_softmax_compositing_backward(dev, N):
    """sem_mode 1 (softmax(logits) composited per sample): d_raw of pnr_composite_backward3 vs torch autograd of the
    oracle's raw2outputs(sem_mode=1), together with every other gradient source; and the probability-map loss terms."""
    R, C, K = 29, 7, 5
    raw, z, rays, ls, li = _level_inputs(R, N, C, K, 100 + N)
    g = torch.Generator().manual_seed(4)
    up = {"rgb": torch.randn(R, 3, generator=g), "depth": torch.randn(R, generator=g), "acc": torch.randn(R, generator=g),
          "semantic": torch.randn(R, C, generator=g), "instance": torch.randn(R, K, generator=g),
          "fix_semantic": torch.randn(R, C, generator=g), "weights": torch.randn(R, N, generator=g)}
    leaf = raw.clone().requires_grad_(True)
    o = to.raw2outputs(leaf, z, rays[:, 3:6], C, K, label_sem=ls, label_inst=li, sem_mode=1)
    sum((o[k] * v).sum() for k, v in up.items()).backward()
    raw_cm = raw.reshape(R * N, -1).T.contiguous().to(dev)
    fwd = ops.composite(raw_cm, z.to(dev), rays.to(dev), C, K, True, None, ls.to(dev), li.to(dev), 1)
    assert (fwd["semantic"].cpu() - o["semantic"].detach()).abs().max() < 2e-5
    d = ops.composite_backward(raw_cm, z.to(dev), rays.to(dev), C, K, {k: v.to(dev) for k, v in up.items()}, None, ls.to(dev),
                               li.to(dev), None, None, 1)
    ref = leaf.grad.reshape(R * N, -1).T
    assert _rel(d.cpu(), ref) < 2e-4, _rel(d.cpu(), ref)
    # probability maps: NLL terms
    maps = {"semantic": fwd["semantic"].cpu(), "fix_semantic": fwd["fix_semantic"].cpu()}
    tg = {"semantic": torch.randint(-1, C, (R,), generator=g, dtype=torch.int32)}
    w = {"semantic": 0.7, "fix_semantic": 0.3}
    leafm = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
    terms, total = to.losses(leafm, tg, w, C, 0, maps_are_prob=True)
    total.backward()
    out, grads = ops.losses(w, {k: v.to(dev) for k, v in maps.items()}, {k: v.to(dev) for k, v in tg.items()}, C, 0, maps_are_prob=True)
    assert abs(out[2].item() - terms["semantic"].item()) < 2e-5 * max(1, abs(terms["semantic"].item()))
    assert abs(out[6].item() - total.item()) < 2e-5 * max(1, abs(total.item()))
    for k in maps:
        assert (grads[k].cpu() - leafm[k].grad).abs().max() <= 1e-6 + 2e-5 * leafm[k].grad.abs().max(), k
