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


@pytest.mark.parametrize("geom", [(2, 128, [], 0, 0), (3, 128, [1], 5, 3), (8, 256, [4], 45, 32), (8, 256, [4], 0, 0),
                                  (3, 128, [1], 5, 3, "feature"), (8, 256, [4], 45, 32, "feature"), (4, 256, [], 19, 0, "feature")])
def test_mlp_backward_matches_autograd(dev, geom):
    """pnr_mlp_forward_train + pnr_mlp_backward + weight-gradient GEMMs vs torch autograd through the
    bf16-emulating oracle MLP (same rounded activations => same ReLU gates; against the fp32 forward the
    gates of near-zero units differ and the error grows ~1.5 % per layer of depth, which says nothing about
    the kernels) for the same upstream d_raw.  Per-tensor relative L2 error < 3 %."""
    from panopticnerf_amd import make_network
    import _wgrad_ref as wref
    from types import SimpleNamespace as NS
    D, W, skips, C, K = geom[:5]
    tap = geom[5] if len(geom) > 5 else "trunk"      # cfg.head_tap: the heads read the trunk output or the feature (a switch)
    torch.manual_seed(D * 7 + W + C)
    net = make_network(NS(D=D, W=W, skips=skips, num_classes=C, num_instances=K, head_tap=tap))
    nerf = net.nerf_0
    rng = np.random.default_rng(D + W)
    R, N = 7, 41                                     # 287 samples: ragged last tile and last group
    rays = torch.tensor(_rays(rng, R, 0.5, 8.0))
    z = torch.tensor(co.stratified(rays.numpy(), N, t_rand=rng.random((R, N)).astype(np.float32)))
    ocfg = to.mlp_config(D=D, W=W, skips=tuple(skips), n_sem=C, n_inst=K, head_W=W // 2, head_tap=tap)
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
    g = wref.weight_grads(nerf, desc, acts, dys, d_cm, R * N)
    assert set(g) == set(params)
    errs = {k: _rel(g[k].cpu(), params[k].grad) for k in params}
    print("rel L2 errors:", {k: round(v, 4) for k, v in errs.items()})
    for k, v in errs.items():
        assert v < 3e-2, (k, v)
    # the hand-written weight-gradient kernel: same bf16 operands, fp32 accumulation in a different order
    gk = ops.mlp_wgrad(desc, acts, dys, R * N, {k: v.shape for k, v in params.items()})
    assert set(gk) == set(params)
    for k in params:
        # output layers: the kernel reduces the bf16 copy of d_raw the dgrad stored, the torch path the fp32 d_raw
        out_layer = k.split(".")[0] in ("rgb_linear", "alpha_linear") or k.startswith(("semantic_linears.1", "instance_linears.1"))
        assert _rel(gk[k].cpu(), g[k].cpu()) < (8e-3 if out_layer else 2e-3), (k, _rel(gk[k].cpu(), g[k].cpu()))
        assert _rel(gk[k].cpu(), params[k].grad) < 3e-2, k


@pytest.mark.parametrize("S", [64, 1000, 16384 + 192, 3 * 16384])
def test_wgrad_kernel_vs_plain_gemm(dev, S):
    """pnr_mlp_wgrad on random bf16 buffers against fp32 matmuls of the same (slot-ordered) regions, un-permuted with
    the host-side slot maps: every job shape (256x256, 256x64, 128x256, 128x32, 32x128, 32x256, 64x128), tile tails
    (S % 64 != 0), slab tails and several slabs.  Not symmetric / not identity: transposes and permutations show."""
    from panopticnerf_amd import make_network
    import _wgrad_ref as wref
    from types import SimpleNamespace as NS
    torch.manual_seed(S)
    net = make_network(NS(num_classes=45, num_instances=32))
    nerf = net.nerf_0
    desc = nerf.desc("bf16")
    ao, do = ops.train_layout(desc, S)
    acts = (torch.randn(ao[-1], device=dev) * 0.5).to(torch.bfloat16)     # padding rows S..S_pad: any finite values
    dys = (torch.randn(do[-1], device=dev) * 0.5).to(torch.bfloat16)
    widths = [nerf.W // 2, nerf.W, nerf.W // 2, nerf.W // 2] + [nerf.W] * nerf.D + [32, 64, 64]
    for i, w in enumerate(widths):                                           # ... but zero dY there (k_mlp_bwd writes zeros)
        wref.fill_saved_rows(dys, do[i], S, w, wref.saved_rows(dys, do[i], S, w).clone())
    shapes = {k: v.shape for k, v in nerf.state_dict().items()}
    gk = ops.mlp_wgrad(desc, acts, dys, S, shapes)
    D, W, H = nerf.D, nerf.W, nerf.W // 2
    A = lambda i, w: wref.saved_rows(acts, ao[i], S, w).float()
    Y = lambda i, w: wref.saved_rows(dys, do[i], S, w).float()
    fW, fH = wref.feat_slots(W, str(dev)), wref.feat_slots(H, str(dev))
    f32s, f64s = wref.feat_slots(32, str(dev)), wref.feat_slots(64, str(dev))
    ex, ed = wref.embed_slots(5, nerf.xyz_L, str(dev)), wref.embed_slots(2, nerf.dir_L, str(dev))
    inv = lambda idx, n: wref._inverse(idx, n)
    def ref(dy, x, ridx, nrow, cidx, ncol, row0=0):
        full = dy.t() @ x                                            # (ma slots, nb slots)
        return full.index_select(0, inv(ridx, ridx.numel())[row0:row0 + nrow]).index_select(1, inv(cidx, ncol))
    def refb(dy, ridx, nrow, row0=0):
        return dy.sum(0).index_select(0, inv(ridx, ridx.numel())[row0:row0 + nrow])
    chk = {}
    chk["pts_linears.0.weight"] = ref(Y(4, W), A(0, 64), fW, W, ex, 63)
    chk["pts_linears.3.weight"] = ref(Y(7, W), A(4, W), fW, W, fW, W)
    chk["pts_linears.5.weight"] = torch.cat([ref(Y(9, W), A(0, 64), fW, W, ex, 63), ref(Y(9, W), A(6, W), fW, W, fW, W)], 1)
    chk["pts_linears.5.bias"] = refb(Y(9, W), fW, W)
    chk["feature_linear.weight"] = ref(Y(1, W), A(1 + D, W), fW, W, fW, W)
    chk["views_linears.0.weight"] = torch.cat([ref(Y(0, H), A(2 + D, W), fH, H, fW, W), ref(Y(0, H), A(1, 32), fH, H, ed, 27)], 1)
    chk["views_linears.0.bias"] = refb(Y(0, H), fH, H)
    chk["rgb_linear.weight"] = ref(Y(4 + D, 32), A(3 + D, H), f32s, 3, fH, H)
    chk["rgb_linear.bias"] = refb(Y(4 + D, 32), f32s, 3)
    chk["alpha_linear.weight"] = ref(Y(4 + D, 32), A(1 + D, W), f32s, 1, fW, W, row0=3)
    chk["alpha_linear.bias"] = refb(Y(4 + D, 32), f32s, 1, row0=3)
    chk["semantic_linears.0.weight"] = ref(Y(2, H), A(1 + D, W), fH, H, fW, W)
    chk["semantic_linears.1.weight"] = ref(Y(5 + D, 64), A(4 + D, H), f64s, 45, fH, H)
    chk["semantic_linears.1.bias"] = refb(Y(5 + D, 64), f64s, 45)
    chk["instance_linears.1.weight"] = ref(Y(6 + D, 64), A(5 + D, H), f64s, 32, fH, H)
    for k, r in chk.items():
        assert gk[k].shape == r.shape, (k, gk[k].shape, r.shape)
        e = _rel(gk[k].cpu(), r.cpu())
        assert e < 1e-5, (k, e)
    # deterministic: the slab partials are reduced in a fixed order
    gk2 = ops.mlp_wgrad(desc, acts, dys, S, shapes)
    assert all(torch.equal(gk[k], gk2[k]) for k in gk)


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


def test_training_step_is_graph_capturable(dev):
    """A whole training step -- NetworkWrapper (render with autograd + fused losses), backward through the HIP kernels,
    Adam(capturable), and the in-place repack of both networks' weight images that the next render triggers -- captured
    into ONE HIP graph and replayed: the parameters after 3 replays equal 3 eager steps from the same start bit for bit
    (every kernel on the path is deterministic)."""
    from panopticnerf_amd import NetworkWrapper, make_network, synthetic
    from types import SimpleNamespace as NS
    import copy
    C, K = 6, 4
    cfg = NS(N_samples=32, N_importance=32, num_classes=C, num_instances=K, precision="bf16", D=4, W=128, skips=[1])
    torch.manual_seed(5)
    net_e = make_network(cfg).to(dev).train()
    net_g = copy.deepcopy(net_e)
    R = 256
    rays = synthetic.camera_rays()[::2003][:R].contiguous()
    box, ids = synthetic.random_boxes(16, C, K, seed=2)
    g = torch.Generator().manual_seed(1)
    batch = {"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev),
             "rgb": torch.rand(1, R, 3, generator=g).to(dev), "depth": (torch.rand(1, R, generator=g) * 20 - 2).to(dev),
             "pseudo_label": torch.randint(-1, C, (1, R), generator=g).to(dev), "instance_label": torch.randint(-1, K, (1, R), generator=g).to(dev)}

    def make(net):
        wrap = NetworkWrapper(net, cfg)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True)
        def step():
            opt.zero_grad(set_to_none=False)
            _, loss, _, _ = wrap(batch)
            loss.backward()
            opt.step()
            return loss
        return step

    step_e, step_g = make(net_e), make(net_g)
    # warm-up on a side stream (torch's capture recipe): first-call allocations, lazy attribute setting, Adam state
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step_g()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(2):
        step_e()
    torch.cuda.synchronize()
    for a, b in zip(net_e.parameters(), net_g.parameters()):
        assert torch.equal(a, b)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        static_loss = step_g()
    losses_g = []
    for _ in range(3):
        graph.replay()
        losses_g.append(static_loss.item())
    # the capture itself does not execute: 3 replays = 3 steps.  Eager: 3 steps.
    losses_e = [step_e().item() for _ in range(3)]
    torch.cuda.synchronize()
    assert losses_g == losses_e, (losses_g, losses_e)
    for (n, a), b in zip(net_e.named_parameters(), net_g.parameters()):
        assert torch.equal(a, b), n


@pytest.mark.parametrize("split", [False, True])
def test_graphed_step_replays_equal_eager_steps_on_changing_batches(dev, split):
    """train.GraphedStep: the whole step as one HIP graph (or, with a `reduce` callable, forward + backward | eager collective |
    optimiser).  Constructing it trains nothing (the warm-up steps are undone in place); replays on CHANGING batches equal eager
    steps bit for bit -- losses and every parameter; a batch of another shape is refused."""
    from types import SimpleNamespace as NS
    import copy
    from panopticnerf_amd import NetworkWrapper, make_network, synthetic, train as pnr_train
    C, K = 6, 4
    cfg = NS(N_samples=32, N_importance=32, num_classes=C, num_instances=K, precision="bf16", D=4, W=128, skips=[1])
    torch.manual_seed(6)
    net_e = make_network(cfg).to(dev).train()
    net_g = copy.deepcopy(net_e)
    R = 256
    box, ids = synthetic.random_boxes(16, C, K, seed=2)
    g = torch.Generator().manual_seed(2)

    def batch(i):
        rays = synthetic.camera_rays()[i::2003][:R].contiguous()
        return {"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev),
                "rgb": torch.rand(1, R, 3, generator=g).to(dev), "depth": (torch.rand(1, R, generator=g) * 20 - 2).to(dev),
                "pseudo_label": torch.randint(-1, C, (1, R), generator=g).to(dev), "instance_label": torch.randint(-1, K, (1, R), generator=g).to(dev)}

    batches = [batch(i) for i in range(4)]
    wrap_e, wrap_g = NetworkWrapper(net_e, cfg), NetworkWrapper(net_g, cfg)
    opt_e = torch.optim.Adam(net_e.parameters(), lr=1e-3, capturable=True, fused=True)
    opt_g = torch.optim.Adam(net_g.parameters(), lr=1e-3, capturable=True, fused=True)
    calls = []
    step = pnr_train.GraphedStep(wrap_g, opt_g, batches[3], reduce=(lambda: calls.append(1)) if split else None)
    n_warm = len(calls)
    for a, b in zip(net_e.parameters(), net_g.parameters()):
        assert torch.equal(a, b)                                   # construction trained nothing
    for st in opt_g.state.values():
        assert all(float(v.abs().sum()) == 0.0 for v in st.values() if torch.is_tensor(v))
    losses_g, losses_e = [], []
    for b in batches[:3]:
        _, loss, stats = step(b)
        losses_g.append(loss.item())
        assert stats["loss"].item() == losses_g[-1]
    for b in batches[:3]:
        opt_e.zero_grad(set_to_none=False)
        _, loss, _, _ = wrap_e(b)
        loss.backward()
        opt_e.step()
        losses_e.append(loss.item())
    assert losses_g == losses_e, (losses_g, losses_e)
    assert losses_g[0] != losses_g[1]                              # the batches do differ
    for (n, a), b in zip(net_e.named_parameters(), net_g.parameters()):
        assert torch.equal(a, b), n
    if split:
        assert len(calls) == n_warm + 3                            # the collective's slot ran eagerly once per step
    with pytest.raises(ValueError, match="captured step takes"):
        step(dict(batches[0], rays=batches[0]["rays"][:, :100]))
    with pytest.raises(ValueError, match="capturable"):
        pnr_train.GraphedStep(wrap_g, torch.optim.Adam(net_g.parameters(), lr=1e-3), batches[0])


def test_graphed_step_with_grad_reducer_finish_as_reduce(dev):
    """`GraphedStep(..., reduce=GradReducer(net).finish)` -- the reducer's post-accumulate hooks fire during the warm-up
    AND during the capture of forward + backward; under capture they must launch no collective (it would be baked into the graph
    and replayed beside the eager one).  A one-rank nccl group with `always=True` keeps hooks and collectives alive: the capture
    succeeds, no Work object survives it, every replay reduces both buckets eagerly in finish(), and three replays equal three
    eager steps bit for bit (mean over one rank = identity)."""
    from types import SimpleNamespace as NS
    import copy
    import socket
    import torch.distributed as dist
    from panopticnerf_amd import NetworkWrapper, make_network, synthetic, train as pnr_train
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        C, K = 6, 4
        cfg = NS(N_samples=32, N_importance=32, num_classes=C, num_instances=K, precision="bf16", D=4, W=128, skips=[1])
        torch.manual_seed(6)
        net_e = make_network(cfg).to(dev).train()
        net_g = copy.deepcopy(net_e)
        R = 256
        box, ids = synthetic.random_boxes(16, C, K, seed=2)
        g = torch.Generator().manual_seed(2)

        def batch(i):
            rays = synthetic.camera_rays()[i::2003][:R].contiguous()
            return {"rays": rays[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev),
                    "rgb": torch.rand(1, R, 3, generator=g).to(dev), "depth": (torch.rand(1, R, generator=g) * 20 - 2).to(dev),
                    "pseudo_label": torch.randint(-1, C, (1, R), generator=g).to(dev), "instance_label": torch.randint(-1, K, (1, R), generator=g).to(dev)}

        batches = [batch(i) for i in range(4)]
        wrap_e, wrap_g = NetworkWrapper(net_e, cfg), NetworkWrapper(net_g, cfg)
        opt_e = torch.optim.Adam(net_e.parameters(), lr=1e-3, capturable=True, fused=True)
        opt_g = torch.optim.Adam(net_g.parameters(), lr=1e-3, capturable=True, fused=True)
        red = pnr_train.GradReducer(net_g, always=True)
        assert red.active and len(red.buckets) == 2 and red.handles
        sends = []
        orig_send = red._send
        red._send = lambda b: (sends.append(torch.cuda.is_current_stream_capturing()), orig_send(b))[1]
        step = pnr_train.GraphedStep(wrap_g, opt_g, batches[3], reduce=red.finish)
        assert not red.works and sends and not any(sends)          # nothing was launched inside the captured region
        n0 = len(sends)
        for b in batches[:3]:
            step(b)
        assert len(sends) == n0 + 3 * 2 and not any(sends)         # per replay: both buckets, from finish(), eagerly
        for b in batches[:3]:
            opt_e.zero_grad(set_to_none=False)
            _, loss, _, _ = wrap_e(b)
            loss.backward()
            opt_e.step()
        for (n, a), b in zip(net_e.named_parameters(), net_g.parameters()):
            assert torch.equal(a, b), n
        red.remove()
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("S_rays", [(37, 24), (16, 64)])      # 888 samples (ragged: S_pad = 1024) and 1024 (S = S_pad)
def test_saved_tensors_layout_gates_and_padding(dev, S_rays):
    """The training buffers as include/pnr.h documents them: read back through the saved-tensor layout (tests/_wgrad_ref.
    saved_rows) and the slot maps, every trunk layer's saved output is relu(W x + b) of the saved layer before it (bf16
    tolerance); the gate bits are exactly [X > 0]; the padding rows S..S_pad of every dY region are zero after
    pnr_mlp_backward and of every acts region finite."""
    from panopticnerf_amd import make_network
    import _wgrad_ref as wref
    from types import SimpleNamespace as NS
    R, N = S_rays
    S = R * N
    torch.manual_seed(S)
    net = make_network(NS(num_classes=6, num_instances=5)).to(dev).train()
    nerf = net.nerf_0
    D, W, H = nerf.D, nerf.W, nerf.W // 2
    rng = np.random.default_rng(S)
    rays = torch.tensor(_rays(rng, R)).to(dev)
    z = ops.stratified(rays, N)
    desc, img = net.packed(0, dev)
    raw, acts = ops.mlp_forward_train(desc, img, rays, z)
    ao, do = ops.train_layout(desc, S)
    Sp = wref.pad_samples(S)
    fW = wref.feat_slots(W, str(dev))
    inv = wref._inverse(fW, W)
    X = [wref.saved_rows(acts, ao[2 + l], S, W).float().index_select(1, inv) for l in range(D)]     # feature order
    sd = {k: v.detach().float() for k, v in nerf.state_dict().items()}
    bf = lambda t: t.to(torch.bfloat16).float()
    for l in range(1, D):
        if l - 1 == nerf.skip:
            continue                                   # the skip layer also takes gamma(x): covered by the gradient tests
        w, b = bf(sd["pts_linears.%d.weight" % l]), sd["pts_linears.%d.bias" % l]
        ref = bf(torch.relu(X[l - 1] @ w.t() + b))
        err = (X[l] - ref).abs().max().item()
        assert err <= 2e-2 * max(1.0, ref.abs().max().item()), (l, err)
    # gate bits: dword j of lane (n, hi) covers blocks 2j, 2j+1; bit 8*(fb&1)+p <-> slot fb*32+hi*16+2p, +16 <-> 2p+1
    abuf = acts.view(torch.int16)
    gate_base = ao[5 + D] + Sp * H                     # the gate regions follow the last acts region, in layout order
    assert gate_base % 64 == 0
    words = abuf[gate_base: gate_base + Sp * (W // 16)].view(torch.int32).view(Sp, W // 32)[:S].to(torch.int64) & 0xffffffff
    Xs = wref.saved_rows(acts, ao[2], S, W).float()    # X_1 in slot order
    for fb in (0, 3, W // 32 - 1):
        for hi in (0, 1):
            for p in (0, 5, 7):
                wd = words[:, hi * (W // 64) + fb // 2]
                lo = (wd >> (8 * (fb & 1) + p)) & 1
                hi_bit = (wd >> (16 + 8 * (fb & 1) + p)) & 1
                s0 = fb * 32 + hi * 16 + 2 * p
                assert torch.equal(lo.bool(), Xs[:, s0] > 0) and torch.equal(hi_bit.bool(), Xs[:, s0 + 1] > 0), (fb, hi, p)
    assert torch.isfinite(acts[: ao[5 + D] + Sp * H].float()).all()
    # backward: zero padding rows in every dY region
    _, img_b = net.packed_bwd(0, dev)
    d_raw = torch.randn_like(raw)
    dys = ops.mlp_backward(desc, img_b, d_raw, acts, R, N)
    widths = [H, W, H, H] + [W] * D + [32, 64, 64]
    for i, w in enumerate(widths):
        full = dys[do[i]: do[i] + Sp * w]
        rows = wref.saved_rows(dys, do[i], Sp, w)       # all S_pad rows
        assert torch.count_nonzero(rows[S:]) == 0, i
        assert torch.isfinite(full.float()).all()


# ----------------------------------------------------------------------------- fp32 parity mode of the training path
@pytest.mark.parametrize("geom", [(2, 128, [], 0, 0, "trunk", 2), (3, 128, [1], 5, 3, "trunk", 2), (8, 256, [4], 45, 32, "trunk", 2),
                                  (8, 256, [4], 45, 32, "feature", 2), (8, 256, [4], 45, 32, "trunk", 1), (4, 256, [], 19, 0, "feature", 1)])
def test_fp32_training_mode_matches_fp32_autograd_to_1e_4(dev, geom):
    """precision = "fp32" with gradients (pnr_mlp_forward_train_fp32 + pnr_mlp_backward_fp32; VERDICT r3 missing 3): the raw
    image and EVERY parameter gradient against torch fp32 autograd through the oracle MLP for the same upstream d_raw, on 287
    samples -- 1e-4 of the tensor's scale, the bar north_star sets for fp32 (the bf16 kernels can only be compared with a
    bf16-emulating oracle at 0.2-1 %).  Covers the head_tap / head_depth switches, incl. head_depth = 1."""
    from panopticnerf_amd import make_network
    from types import SimpleNamespace as NS
    D, W, skips, C, K, tap, depth = geom
    torch.manual_seed(D * 11 + W + C + depth)
    net = make_network(NS(D=D, W=W, skips=skips, num_classes=C, num_instances=K, head_tap=tap, head_depth=depth, precision="fp32"))
    nerf = net.nerf_0.to(dev)
    rng = np.random.default_rng(D + W + 1)
    R, N = 7, 41
    rays = torch.tensor(_rays(rng, R, 0.5, 8.0))
    z = torch.tensor(co.stratified(rays.numpy(), N, t_rand=rng.random((R, N)).astype(np.float32)))
    ocfg = to.mlp_config(D=D, W=W, skips=tuple(skips), n_sem=C, n_inst=K, head_W=W // 2, head_tap=tap, head_depth=depth)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in nerf.state_dict().items()}
    raw_ref = to.run_network(params, ocfg, rays, z)
    d_raw = torch.tensor(rng.normal(size=raw_ref.shape).astype(np.float32))
    (raw_ref * d_raw).sum().backward()
    desc = nerf.desc("fp32")
    live = {n: p.detach() for n, p in nerf.named_parameters()}
    raw, acts = ops.mlp_forward_train_fp32(desc, live, rays.to(dev), z.to(dev))
    err = (raw.T.reshape(R, N, -1).cpu() - raw_ref.detach()).abs().max().item()
    assert err < 1e-4 * max(1.0, raw_ref.abs().max().item()), err
    g = ops.mlp_backward_fp32(desc, live, d_raw.reshape(R * N, -1).T.contiguous().to(dev), acts, R, N)
    assert set(g) == set(params)
    worst = {}
    for k in params:
        ref = params[k].grad
        worst[k] = ((g[k].cpu() - ref).abs().max() / ref.abs().max().clamp(min=1e-12)).item()
    print("fp32 training mode, max |g - ref| / max |ref| per tensor: worst %.2e (%s)" % (max(worst.values()), max(worst, key=worst.get)))
    for k, v in worst.items():
        assert v < 1e-4, (k, v)


def test_fp32_render_backward_end_to_end_to_1e_4(dev):
    """Renderer.render with precision = "fp32" under autograd (the NotImplementedError of round 3 is gone): loss on the maps
    of both levels; every parameter gradient vs fp32 autograd through the oracle fed the HIP path's own z, to 1e-4 of scale."""
    from panopticnerf_amd import make_network, make_renderer
    from types import SimpleNamespace as NS
    C, K = 6, 4
    cfg = NS(N_samples=16, N_importance=16, num_classes=C, num_instances=K, precision="fp32", D=4, W=128, skips=[1])
    torch.manual_seed(5)
    net = make_network(cfg).to(dev).train()
    with torch.no_grad():
        for lv in (0, 1):
            net.nerf(lv).alpha_linear.bias.fill_(0.3)
    rend = make_renderer(cfg, net)
    rng = np.random.default_rng(6)
    R = 16                                              # 16 rays x (16 + 32) samples = 768 samples
    rays = torch.tensor(_rays(rng, R, 0.5, 6.0))
    tgt = {k: torch.tensor(rng.normal(size=s).astype(np.float32)) for k, s in
           (("rgb", (R, 3)), ("depth", (R,)), ("semantic", (R, C)), ("instance", (R, K)))}
    out = rend.render({"rays": rays[None].to(dev)})
    loss = sum(((out[f"{k}_{lv}"][0] - v.to(dev)) ** 2).mean() for lv in (0, 1) for k, v in tgt.items())
    loss.backward()
    ocfg = to.mlp_config(D=4, W=128, skips=(1,), n_sem=C, n_inst=K, head_W=64)
    ref_loss, ref_params = 0, {}
    for lv in (0, 1):
        prm = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in net.nerf(lv).state_dict().items()}
        ref_params[lv] = prm
        zz = out[f"z_vals_{lv}"][0].detach().cpu()
        o = to.raw2outputs(to.run_network(prm, ocfg, rays, zz), zz, rays[:, 3:6], C, K)
        ref_loss = ref_loss + sum(((o[k] - v) ** 2).mean() for k, v in tgt.items())
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * abs(ref_loss.item())
    for lv in (0, 1):
        for name, p in net.nerf(lv).named_parameters():
            ref = ref_params[lv][name].grad
            e = ((p.grad.cpu() - ref).abs().max() / ref.abs().max().clamp(min=1e-12)).item()
            assert e < 1e-4, (lv, name, e)


# ----------------------------------------------------------------------------- the backward twin of the switch matrix
@pytest.mark.parametrize("tap,depth", [("trunk", 1), ("feature", 1), ("feature", 2), ("trunk", 2)])
@pytest.mark.parametrize("geom", [(8, 256, [4], 45, 32), (4, 128, [1], 6, 0), (3, 256, [], 0, 7)])
def test_head_tap_and_depth_switches_backward(dev, geom, tap, depth):
    """cfg.head_tap x cfg.head_depth through the bf16 TRAINING kernels (head_depth = 1 used to be inference
    only): pnr_mlp_forward_train + pnr_mlp_backward + pnr_mlp_wgrad against autograd through the oracle MLP with the same
    switches, in the kernels' own arithmetic (bf16 forward and bf16 dY: torch_oracle's emulate_bf16 = "bwd").  With one Linear
    per head the logit gradients feed the tap's gradient chain directly (pnr_mlp_plan.h)."""
    from panopticnerf_amd import make_network
    from types import SimpleNamespace as NS
    D, W, skips, C, K = geom
    torch.manual_seed(D * 5 + W + C + depth)
    net = make_network(NS(D=D, W=W, skips=skips, num_classes=C, num_instances=K, head_tap=tap, head_depth=depth))
    nerf = net.nerf_0
    rng = np.random.default_rng(D + W + depth)
    R, N = 9, 37                                     # 333 samples: ragged last tile and last group
    rays = torch.tensor(_rays(rng, R, 0.5, 8.0))
    z = torch.tensor(co.stratified(rays.numpy(), N, t_rand=rng.random((R, N)).astype(np.float32)))
    ocfg = to.mlp_config(D=D, W=W, skips=tuple(skips), n_sem=C, n_inst=K, head_W=W // 2, head_tap=tap, head_depth=depth)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in nerf.state_dict().items()}
    raw_ref = to.run_network(params, ocfg, rays, z, emulate_bf16="bwd")
    d_raw = torch.tensor(rng.normal(size=raw_ref.shape).astype(np.float32))
    (raw_ref * d_raw).sum().backward()
    desc, img = net.packed(0, dev, "bf16")
    assert desc.head_depth == depth and desc.head_tap == (1 if tap == "feature" else 0)
    raw, acts = ops.mlp_forward_train(desc, img, rays.to(dev), z.to(dev))
    assert (raw.T.reshape(R, N, -1).cpu() - raw_ref.detach()).abs().max() < 6e-2
    d_cm = d_raw.reshape(R * N, -1).T.contiguous().to(dev)
    _, img_b = net.packed_bwd(0, dev)
    dys = ops.mlp_backward(desc, img_b, d_cm, acts, R, N)
    g = ops.mlp_wgrad(desc, acts, dys, R * N, {k: v.shape for k, v in params.items()})
    assert set(g) == set(params)
    errs = {k: _rel(g[k].cpu(), params[k].grad) for k in params}
    print("rel L2 errors (%s, depth %d):" % (tap, depth), {k: round(v, 4) for k, v in errs.items() if v > 5e-3})
    for k, v in errs.items():
        assert v < 3e-2, (k, v)


def test_hip_gradient_is_the_emulated_bf16_gradient_at_the_benched_geometry(dev):
    """One training step's gradient at the benched geometry (8x256 + 45 / 32 heads, 64 + 128 samples, bbox prior, the trainer's
    loss wrapper with UNIT weights), three ways: HIP (bf16 forward, bf16 dY), the oracle's emulation of exactly that
    arithmetic (emulate_bf16 = "bwd") and fp32 autograd.  What it pins: (1) the HIP gradient IS the
    emulated one -- so the CPU students of tools/train_fidelity.py speak for the kernels; (2) where the bf16 path's gradient
    error comes from: with unit weights the cross-entropy terms make the trunk gradient ~17x the colour term's own gradient, and
    the bf16 FORWARD's perturbation of that large gradient is as large as the colour component itself, while rounding every
    dY to bf16 in the backward adds almost nothing on top (bf16_fwd vs bf16_bwd columns)."""
    import _students as S
    from panopticnerf_amd import NetworkWrapper, make_network
    from types import SimpleNamespace as NS
    torch.set_num_threads(min(32, torch.get_num_threads()))
    sc = S.scene(steps=1, batch=192)
    idx = sc.batches[0]
    rgb_only = {"rgb": 1.0, "depth": 0.0, "semantic": 0.0, "fix_semantic": 0.0, "instance": 0.0, "fix_instance": 0.0}
    ref = {m: S.oracle_gradient(sc, idx, m, S.UNIT, 0.1) for m in ("fp32", "bf16_fwd", "bf16_bwd")}
    g_rgb = S.oracle_gradient(sc, idx, "fp32", rgb_only, 0.0)
    hip = S.hip_gradient(sc, idx, dev, S.UNIT, 0.1)
    n = lambda t: t.norm().item()
    worst_emu, rows = 0.0, []
    for k in ref["fp32"]:
        if k.endswith(".bias") or n(ref["fp32"][k]) == 0:
            continue
        e_emu = n(hip[k] - ref["bf16_bwd"][k]) / n(ref["bf16_bwd"][k])
        e_f32 = n(hip[k] - ref["fp32"][k]) / n(ref["fp32"][k])
        worst_emu = max(worst_emu, e_emu)
        rows.append((k, n(ref["fp32"][k]), n(g_rgb[k]), e_emu, e_f32,
                     n(ref["bf16_fwd"][k] - ref["fp32"][k]) / max(n(g_rgb[k]), 1e-30), n(ref["bf16_bwd"][k] - ref["fp32"][k]) / max(n(g_rgb[k]), 1e-30),
                     n(hip[k] - ref["fp32"][k]) / max(n(g_rgb[k]), 1e-30)))
    print("%-36s %9s %9s | %8s %8s | err / |g_rgb|: %7s %7s %7s" % ("weight", "|g|", "|g_rgb|", "hip~emu", "hip~f32", "fwd", "bwd", "hip"))
    for r in rows:
        if r[2] > 0:
            print("%-36s %9.2e %9.2e | %8.4f %8.4f | %22.3f %7.3f %7.3f" % r)
        else:       # the heads: the colour term has no gradient there
            print("%-36s %9.2e %9s | %8.4f %8.4f |" % (r[0], r[1], "-", r[3], r[4]))
    # (1) HIP = its emulation (pass A, round 4: 1e-4 ... 2e-4 at the coarse level, whose sample positions are bit-identical; <= 1e-2
    #     at the fine level, whose positions come from two bf16 coarse passes that differ in accumulation order), far closer than
    #     either is to fp32 (0.5 ... 6 %)
    coarse = [r for r in rows if r[0].startswith("coarse")]
    assert max(r[3] for r in coarse) < 2e-3, max(r[3] for r in coarse)
    assert worst_emu < 2e-2, worst_emu
    trunk = [r for r in rows if "pts_linears" in r[0] and r[0].startswith("fine")]
    assert np.mean([r[3] for r in trunk]) < 0.5 * np.mean([r[4] for r in trunk])
    # (2) the backward's own rounding is a small part of the bf16 path's deviation from fp32
    assert np.mean([abs(r[6] - r[5]) for r in trunk]) < 0.1 * np.mean([r[5] for r in trunk])
