"""The drop-in boundary, tested the way the reference uses it: the adapter files
under integration/ are loaded BY PATH from config strings, `Network()` is built with no arguments, `Renderer(net)` /
`NetworkWrapper(net)` / `Evaluator()` likewise, and the reference's own training-loop shape -- including an unchanged
`DistributedDataParallel` wrap -- runs on top.  Construction on CPU; render / train on the GPU."""
import os
import socket

import pytest
import torch

import _ref_harness as ref
from oracle import torch_oracle as to
from panopticnerf_amd import Network, NetworkWrapper, Renderer, synthetic
from panopticnerf_amd import make_network as direct_make_network
from panopticnerf_amd import make_renderer as direct_make_renderer
from panopticnerf_amd.evaluate import Evaluator


def test_plugins_resolve_by_path_with_zero_arg_network():
    cfg = ref.load_cfg(num_classes=7, num_instances=5)
    net = ref.make_network(cfg)
    assert isinstance(net, Network) and type(net).__module__ == cfg.network_module
    # the zero-arg Network() saw the global cfg: fine NeRF and both heads exist (INTEGRATION.md r1 bound a bare class here)
    assert net.nerf_1 is not None and net.nerf_0.n_sem == 7 and net.nerf_1.n_inst == 5 and net.precision == "bf16"
    rend = ref.make_renderer(cfg, net)
    assert isinstance(rend, Renderer) and rend.N_samples == 64 and rend.N_importance == 128 and rend.max_hits == 8
    wrap = ref.make_network_wrapper(cfg, net)
    assert isinstance(wrap, NetworkWrapper) and wrap.net is net and list(wrap.parameters())
    ev = ref.make_evaluator(cfg)
    assert isinstance(ev, Evaluator) and ev.n_classes == 7
    # cascade_samples instead of N_importance: both sides fall back the same way
    d = dict(vars(ref.load_cfg(num_classes=0, num_instances=0)))
    d.pop("N_importance")
    cfg2 = ref.load_cfg(**{k: v for k, v in d.items()}, cascade_samples=32)
    delattr(cfg2, "N_importance")
    net2 = ref.make_network(cfg2)
    assert net2.nerf_1 is not None and ref.make_renderer(cfg2, net2).N_importance == 32
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rend.render({"rays": torch.zeros(1, 4, 8)})


def _batch(dev, C, K, R=512, targets=False, seed=0):
    g = torch.Generator().manual_seed(seed)
    rays = synthetic.camera_rays()[:: (1408 * 376) // R][:R].contiguous()
    box, ids = synthetic.random_boxes(24, C, K, seed=3)
    b = {"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    if targets:
        b.update({"rgb": torch.rand(1, R, 3, generator=g).to(dev), "depth": (torch.rand(1, R, generator=g) * 60 - 10).to(dev),
                  "pseudo_label": torch.randint(-1, C, (1, R), generator=g).to(dev),
                  "instance_label": torch.randint(-1, K, (1, R), generator=g).to(dev)})
    return b, rays, box, ids


@pytest.mark.gpu
def test_plugins_render_like_the_direct_path_and_the_oracle(dev):
    C, K = 6, 4
    cfg = ref.load_cfg(num_classes=C, num_instances=K)
    torch.manual_seed(1)
    net = ref.make_network(cfg).eval()
    synthetic.trained_like_(net, 0.05)
    net = net.to(dev)
    rend = ref.make_renderer(cfg, net)
    batch, rays, box, ids = _batch(dev, C, K)
    with torch.no_grad():
        out = rend.render(batch)
    assert out["rgb_1"].shape == (1, 512, 3) and out["semantic_1"].shape == (1, 512, C) and out["fix_instance_1"].shape == (1, 512, K)
    # the same weights through the package's own factories: bit-identical
    net_d = direct_make_network(cfg).eval()
    net_d.load_state_dict(net.state_dict())
    with torch.no_grad():
        out_d = direct_make_renderer(cfg, net_d.to(dev)).render(batch)
    for k in out:
        assert torch.equal(out[k], out_d[k]), k
    # and against the oracle (bf16 emulation of the same weights)
    oc = to.mlp_config(n_sem=C, n_inst=K)
    params = {"coarse": {k: v.cpu() for k, v in net.nerf_0.state_dict().items()},
              "fine": {k: v.cpu() for k, v in net.nerf_1.state_dict().items()}}
    want = to.render_rays(params, oc, rays, 64, 128, box=box, box_ids=ids, emulate_bf16=True)
    assert (out["rgb_1"][0].cpu() - want["rgb_1"]).abs().max() < 2e-2
    assert (out["fix_semantic_1"][0].cpu() - want["fix_semantic_1"]).abs().max() < 2e-2
    # evaluator plugin consumes the renderer's output
    ev = ref.make_evaluator(cfg)
    res = ev.evaluate(out, {"rgb": want["rgb_1"][None].to(dev), "pseudo_label": torch.zeros(1, 512, dtype=torch.int32, device=dev)})
    s = ev.summarize()
    assert res["panoptic_id"].shape == (512,) and s["psnr"] > 35.0 and 0.0 <= s["pixel_acc"] <= 1.0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_ddp_wraps_the_wrapper_unchanged(dev):
    """The reference trainer does DistributedDataParallel(NetworkWrapper(net), device_ids=[local_rank]).  The HIP path
    has no torch forward, but the parameters enter LevelFn.apply as autograd inputs, so DDP's AccumulateGrad hooks fire
    like for any module: the wrapped step must leave the same gradients as the un-wrapped one (world size 1 here;
    tests/test_gpu_multi.py does the 2-rank average)."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    C, K = 5, 3
    cfg = ref.load_cfg(num_classes=C, num_instances=K)
    torch.manual_seed(2)
    net = ref.make_network(cfg)
    synthetic.trained_like_(net, 0.05)
    net = net.to(dev).train()
    wrap = ref.make_network_wrapper(cfg, net)
    batch, *_ = _batch(dev, C, K, R=256, targets=True)
    _, loss, stats, _ = wrap(batch)
    loss.backward()
    want = {n: p.grad.clone() for n, p in net.named_parameters()}
    assert all(g.abs().sum() > 0 for g in want.values())
    net.zero_grad(set_to_none=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ddp = DDP(wrap, device_ids=[dev.index])
        for _ in range(2):                                # twice: DDP's reducer must be re-armed by a complete first backward
            net.zero_grad(set_to_none=True)
            _, loss2, stats2, _ = ddp(batch)
            loss2.backward()
        assert torch.equal(loss2.detach(), loss.detach())
        for n, p in net.named_parameters():
            assert p.grad is not None and torch.equal(p.grad, want[n]), n
    finally:
        dist.destroy_process_group()
