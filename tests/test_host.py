"""CPU tests of the host logic: the plugin surface (make_network / make_renderer), the
state_dict key layout, loud failure off-GPU, synthetic inputs, and the ray-sharding path at
world_size 2 over gloo (the oracle stands in for the GPU renderer -- tests may use it)."""
import os
import socket
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import torch_oracle as to
from panopticnerf_amd import Network, Renderer, make_network, make_renderer, shard, synthetic


def test_plugin_surface_and_state_dict_keys():
    cfg = NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32)
    net = make_network(cfg)
    assert isinstance(net, Network) and isinstance(net, torch.nn.Module)
    rend = make_renderer(cfg, net)
    assert isinstance(rend, Renderer) and callable(rend.render)
    keys = set(net.state_dict())
    for lv in (0, 1):
        for name in ("pts_linears.0", "pts_linears.7", "alpha_linear", "feature_linear", "views_linears.0",
                     "rgb_linear", "semantic_linears.0", "semantic_linears.1", "instance_linears.1"):
            assert f"nerf_{lv}.{name}.weight" in keys and f"nerf_{lv}.{name}.bias" in keys
    sd = net.nerf_0.state_dict()
    assert sd["pts_linears.0.weight"].shape == (256, 63)
    assert sd["pts_linears.5.weight"].shape == (256, 319)        # skip: [gamma(x), h]
    assert sd["views_linears.0.weight"].shape == (128, 283)      # [feature, gamma(d)]
    assert sd["semantic_linears.1.weight"].shape == (45, 128)
    # same names/shapes as the oracle's parameter dict => checkpoints map key for key
    ref = to.init_params(to.mlp_config(n_sem=45, n_inst=32))
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v.shape) for k, v in ref.items()}
    # coarse-only config has no fine network
    assert make_network(NS(N_importance=0)).nerf_1 is None


def test_packed_cache_tracks_parameter_updates():
    net = make_network(NS(D=2, W=128, skips=[]))
    d1, img1 = net.packed(0, "cpu")
    d2, img2 = net.packed(0, "cpu")
    assert img1 is img2
    with torch.no_grad():
        net.nerf_0.rgb_linear.bias.add_(1.0)
    _, img3 = net.packed(0, "cpu")
    assert img3 is not img1 and not torch.equal(img3, img1)


def test_packed_cache_eval_vs_invalidate():
    net = make_network(NS(D=2, W=128, skips=[])).eval()
    _, img1 = net.packed(0, "cpu")
    net.nerf_0.rgb_linear.bias.data.add_(1.0)            # .data write: the tensor version does NOT move
    assert net.packed(0, "cpu")[1] is img1                # ... so the cache cannot see it
    net.invalidate_packed()
    _, img2 = net.packed(0, "cpu")
    assert img2 is not img1 and not torch.equal(img2, img1)


def test_fine_level_needs_a_fine_network_unless_sharing_is_asked_for():
    # Network() / Network(cfg without N_importance) used to serve level 1 from the coarse NeRF silently
    net0 = make_network(None)
    assert net0.nerf_1 is None
    with pytest.raises(RuntimeError, match="no fine NeRF"):
        net0.nerf(1)
    with pytest.raises(ValueError, match="fine pass"):
        make_renderer(NS(N_samples=64, N_importance=128), net0)
    with pytest.raises(ValueError, match="fine pass"):
        make_renderer(NS(N_samples=64, cascade_samples=128), make_network(NS(N_samples=64)))
    # the same key fallback on both sides: cascade_samples alone builds the fine NeRF
    net = make_network(NS(N_samples=64, cascade_samples=128))
    assert net.nerf_1 is not None and make_renderer(NS(N_samples=64, cascade_samples=128), net).N_importance == 128
    # explicit weight sharing
    shared = make_network(NS(N_samples=64, N_importance=128, share_coarse_fine=True))
    assert shared.nerf_1 is None and shared.nerf(1) is shared.nerf_0
    make_renderer(NS(N_samples=64, N_importance=128), shared)


def test_renderer_fails_loudly_without_gpu():
    cfg = NS(N_samples=8, N_importance=0)
    rend = make_renderer(cfg, make_network(cfg).eval())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rend.render({"rays": torch.zeros(1, 4, 8)})
    with pytest.raises(RuntimeError):
        make_network(cfg)(torch.zeros(1))


def test_synthetic_inputs_are_kitti_shaped():
    rays = synthetic.camera_rays()
    assert rays.shape == (1408 * 376, 8) and rays.dtype == torch.float32
    assert torch.all(rays[:, 6] == 0.5) and torch.all(rays[:, 7] == 100.0)
    box, ids = synthetic.random_boxes(64, 45, 32)
    assert box.shape == (64, 15) and ids.shape == (64, 2) and ids.dtype == torch.int32
    Rm = box[:, 3:12].reshape(-1, 3, 3)
    assert torch.allclose(Rm @ Rm.transpose(1, 2), torch.eye(3).expand(64, 3, 3), atol=1e-6)
    hits = to.bbox_hits(rays[::997], box, 8)
    assert hits[2].sum() > 0          # the prior is exercised by the synthetic scene


def test_chunk_plan_is_balanced_and_whole_rounds():
    """Renderer.render's chunks (renderer.chunk_plan): they cover the rays exactly once in order; a rank's 66,176-ray share of the
    1408 x 376 frame (strong scaling over 8 ranks) is ONE chunk; the full frame is 8 chunks that are all whole rounds of the
    MLP kernels' persistent grid at 64 and at 192 samples per ray; tiny inputs and tiny chunk sizes still work."""
    from panopticnerf_amd.renderer import chunk_plan, CHUNK_QUANTUM
    assert chunk_plan(0, 65536) == [] and chunk_plan(5, 65536) == [(0, 5)]
    assert chunk_plan(529408 // 8, 65536) == [(0, 66176)]
    full = chunk_plan(529408, 65536)
    assert len(full) == 8 and all((e - s) % CHUNK_QUANTUM == 0 for s, e in full)
    for n_samples in (64, 192):
        assert all(((e - s) * n_samples) % (256 * 256) == 0 for s, e in full)
    rng = np.random.default_rng(0)
    for _ in range(200):
        R, c = int(rng.integers(1, 700000)), int(rng.choice([1, 7, 384, 4096, 65536, 100000]))
        plan = chunk_plan(R, c)
        assert plan[0][0] == 0 and plan[-1][1] == R and all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
        assert all(0 < e - s <= c + c // 8 + (CHUNK_QUANTUM if c >= 8 * CHUNK_QUANTUM else 0) for s, e in plan)   # chunk_size bounds the working set: at most 1/8 over (+ one quantum)
        assert len(plan) <= R // c + 1


def test_shard_roundtrip_single_process():
    rays = torch.arange(11 * 8, dtype=torch.float32).reshape(11, 8)
    parts = [shard.shard_rays(rays, r, 3) for r in range(3)]
    assert sum(p.shape[0] for p in parts) == 11
    assert torch.equal(parts[1][0], rays[1]) and torch.equal(parts[2][1], rays[5])
    assert shard.gather_maps({"a": rays}, 11, 0, 1)["a"] is rays


def _ar_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from panopticnerf_amd import train
    torch.manual_seed(0)
    net = make_network(NS(D=2, W=128, skips=[], num_classes=3, N_importance=8))
    for i, p in enumerate(net.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    net.nerf_0.rgb_linear.bias.grad = None                     # a parameter without gradient is skipped
    train.allreduce_grads(net, world)
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1)))
             for i, p in enumerate(net.parameters()) if p.grad is not None)
    ok = ok and net.nerf_0.rgb_linear.bias.grad is None
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def _reducer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from panopticnerf_amd import train
    ok = True
    for share in (False, True):
        torch.manual_seed(0)
        net = make_network(NS(D=2, W=128, skips=[], num_classes=3, N_importance=8, share_coarse_fine=share))
        red = train.GradReducer(net, world)
        assert len(red.buckets) == (1 if share else 2)
        sent_in_backward = []
        for step in range(2):                                  # twice: the reducer re-arms itself in finish()
            for p in net.parameters():
                p.grad = None
            unused = net.nerf_0.rgb_linear.bias                # never enters the loss: its bucket completes only in finish()
            loss = 0.0
            for lv in ((1, 0) if not share else (0, 0)):       # a shared NeRF is used by both levels
                for i, p in enumerate(net.nerf(lv).parameters()):
                    if p is not unused:
                        loss = loss + (p * float((rank + 1) * (i + 1) * (lv + 1))).sum()
            loss.backward()
            sent_in_backward.append([b["sent"] for b in red.buckets])
            red.finish()
            for lv in ((1, 0) if not share else (0,)):
                for i, p in enumerate(net.nerf(lv).parameters()):
                    if p is unused:
                        ok = ok and p.grad is None
                        continue
                    uses = 2 if share else 1
                    want = 1.5 * (i + 1) * (lv + 1) * uses     # mean over ranks 1, 2 of (rank + 1) * ...
                    ok = ok and torch.allclose(p.grad, torch.full_like(p, want))
        # gradient accumulation: two backward() calls before one finish() -- the early bucket is stale and everything is reduced again
        for p in net.parameters():
            p.grad = None
        for _rep in range(2):
            loss = 0.0
            for lv in ((1, 0) if not share else (0, 0)):
                for i, p in enumerate(net.nerf(lv).parameters()):
                    if p is not net.nerf_0.rgb_linear.bias:
                        loss = loss + (p * float((rank + 1) * (i + 1) * (lv + 1))).sum()
            loss.backward()
        red.finish()
        for lv in ((1, 0) if not share else (0,)):
            for i, p in enumerate(net.nerf(lv).parameters()):
                if p is not net.nerf_0.rgb_linear.bias:
                    ok = ok and torch.allclose(p.grad, torch.full_like(p, 2 * 1.5 * (i + 1) * (lv + 1) * (2 if share else 1)))
        # the fine bucket (complete: every parameter used) went out DURING backward; the coarse one (an unused parameter) in finish()
        ok = ok and (sent_in_backward == ([[False]] * 2 if share else [[True, False]] * 2))
        # under stream capture (GraphedStep captures forward + backward with `reduce=reducer.finish`) a hook must launch NOTHING --
        # a collective inside the captured region would be baked into the graph -- and finish() inside the capture is
        # refused; called eagerly afterwards it sends every bucket and the means come out as before
        train.GradReducer._capturing = staticmethod(lambda: True)
        for p in net.parameters():
            p.grad = None
        loss = 0.0
        for lv in ((1, 0) if not share else (0, 0)):
            for i, p in enumerate(net.nerf(lv).parameters()):
                loss = loss + (p * float((rank + 1) * (i + 1) * (lv + 1))).sum()
        loss.backward()
        ok = ok and not any(b["sent"] for b in red.buckets) and not red.works
        try:
            red.finish()
            ok = False
        except RuntimeError as e:
            ok = ok and "capture" in str(e)
        train.GradReducer._capturing = staticmethod(lambda: False)
        red.finish()
        for lv in ((1, 0) if not share else (0,)):
            for i, p in enumerate(net.nerf(lv).parameters()):
                ok = ok and torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1) * (lv + 1) * (2 if share else 1)))
        red.remove()
    one = train.GradReducer(make_network(NS(D=2, W=128, skips=[])), 1)
    one.finish()                                               # world 1: nothing registered, nothing to do
    ok = ok and not one.handles
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_grad_reducer_overlapped_buckets_world2_gloo():
    """train.GradReducer (the fine network's all-reduce beside the coarse level's backward): per-NeRF buckets launched
    from post-accumulate hooks in a rank-independent order, the rest in finish(); means equal the flat-bucket form's."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_flat_bucket_grad_allreduce_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ar_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def test_weight_grad_slot_maps_are_permutations():
    import _wgrad_ref as wref
    for w in (128, 256):
        idx = wref.feat_slots(w, "cpu")
        assert sorted(idx.tolist()) == list(range(w))
    ex = wref.embed_slots(5, 10, "cpu")
    assert sorted(i for i in ex.tolist() if i >= 0) == list(range(63)) and (ex < 0).sum() == 1
    ed = wref.embed_slots(2, 4, "cpu")
    assert sorted(i for i in ed.tolist() if i >= 0) == list(range(27)) and (ed < 0).sum() == 5
    assert sorted(i for i in wref.embed_slots(5, 6, "cpu").tolist() if i >= 0) == list(range(39))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rays, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    cfg = to.mlp_config(D=2, W=32, skips=(), n_sem=3, n_inst=2, head_W=16)
    params = {"coarse": to.init_params(cfg, 1, sigma_bias=0.1), "fine": to.init_params(cfg, 2, sigma_bias=0.1)}
    rays = synthetic.camera_rays()[:: (1408 * 376) // n_rays][:n_rays].contiguous()

    def render(r):
        if r.shape[0] == 0:      # an empty shard (fewer rays than ranks): what Renderer._empty_outputs returns on the GPU
            return {"rgb_1": torch.empty(0, 3), "depth_1": torch.empty(0), "semantic_1": torch.empty(0, 3), "z_vals_1": torch.empty(0, 32)}
        o = to.render_rays(params, cfg, r, 16, 16)
        return {k: o[k] for k in ("rgb_1", "depth_1", "semantic_1", "z_vals_1")}

    full = shard.render_sharded(render, rays, rank, world, gather=True)
    # reduce before the gather (here: a CPU argmax standing in for pnr_panoptic_labels) and gather selected keys only
    red = lambda o: {"rgb_1": o["rgb_1"], "semantic_label": o["semantic_1"].argmax(-1).int()}
    lab = shard.render_sharded(render, rays, rank, world, gather=True, keys=("semantic_label",), reduce_fn=red)
    assert set(lab) == {"semantic_label"} and lab["semantic_label"].dtype == torch.int32
    assert torch.equal(lab["semantic_label"], full["semantic_1"].argmax(-1).int())
    local = shard.render_sharded(render, rays, rank, world, gather=False)
    assert local["rgb_1"].shape[0] == len(range(rank, n_rays, world))
    # maps of other widths travel through exact carriers (bool / uint8 / int64 / bf16 / float64), with and without the flat
    # collective (the one-bucket gather used to refuse anything but 4-byte maps)
    def odd(r):
        o = render(r)
        return {"mask": o["depth_1"] > o["depth_1"].mean(), "u8": (o["rgb_1"] * 255).to(torch.uint8), "i64": (o["depth_1"] * 1e6).long(),
                "bf": o["rgb_1"].bfloat16(), "f64": o["semantic_1"].double() / 3, "rgb_1": o["rgb_1"]}
    ref_odd = odd(rays)
    for flat in (True, False):
        keep = dist.all_gather_into_tensor
        if not flat:
            def _no_flat(*a, **k):
                raise NotImplementedError("no all_gather_into_tensor in this backend")
            dist.all_gather_into_tensor = _no_flat
        try:
            got = shard.render_sharded(odd, rays, rank, world, gather=True)
        finally:
            dist.all_gather_into_tensor = keep
        assert all(got[k].dtype == ref_odd[k].dtype and got[k].shape == ref_odd[k].shape for k in ref_odd)
        for k in ("mask", "u8", "i64", "bf", "f64"):
            a, b = got[k], ref_odd[k]
            # whole-frame vs per-shard renders differ in the last bits; exactness of the CARRIER is checked on this rank's own rows
            mine = odd(shard.shard_rays(rays, rank, world))[k]
            assert torch.equal(a[rank::world], mine), (k, flat)
            assert a.shape == b.shape
    ref = render(rays)
    ok = all(torch.allclose(full[k], ref[k], atol=1e-6) for k in ref)
    # every rank holds the same complete frame
    chk = full["rgb_1"].double().sum().reshape(1)
    both = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(both, chk)
    ok = ok and all(torch.equal(b, both[0]) for b in both)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


@pytest.mark.parametrize("n_rays", [10, 7, 1])      # even and ragged split over 2 ranks; one ray: rank 1's shard is empty
def test_sharded_render_world2_gloo(n_rays):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rays, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r for r, _ in res) == [0, 1] and all(ok for _, ok in res)


def test_reference_checkpoint_key_map():
    """a checkpoint with wrapper prefixes / alternative names loads through the key map; wrong shapes and
    unknown keys are loud."""
    cfg = NS(D=2, W=128, skips=[], num_classes=3, N_importance=8)
    src, dst = make_network(cfg), make_network(cfg)
    with torch.no_grad():
        for p in src.parameters():
            p.normal_()
    ren = lambda k: "module.net." + k.replace("nerf_0.", "coarse.").replace("nerf_1.", "fine.").replace("alpha_linear", "sigma_linear")
    sd = {ren(k): v.clone() for k, v in src.state_dict().items()}
    rep = dst.load_reference_state_dict(sd)
    assert not rep["missing"] and not rep["unexpected"] and len(rep["loaded"]) == len(sd)
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
    with pytest.raises(KeyError):
        dst.load_reference_state_dict({**sd, "module.net.coarse.extra.weight": torch.zeros(1)})
    bad = dict(sd)
    k0 = next(k for k in bad if k.endswith("rgb_linear.weight"))
    bad[k0] = torch.zeros(3, 7)
    with pytest.raises(ValueError):
        dst.load_reference_state_dict(bad)
    rep = dst.load_reference_state_dict({k: v for k, v in sd.items() if "fine." not in k}, strict=False)
    assert rep["missing"] and all(m.startswith("nerf_1.") for m in rep["missing"])


def test_reference_checkpoint_concat_order_switches():
    """a checkpoint of a network that concatenates [h, gamma(x)] at the skip layer and [gamma(d), feature] at
    the view layer loads into this ([gamma(x), h] / [feature, gamma(d)]) network with the columns rotated -- and only those."""
    cfg = NS(D=4, W=128, skips=[1], num_classes=3, N_importance=8)
    src, dst = make_network(cfg), make_network(cfg)
    with torch.no_grad():
        for p in src.parameters():
            p.normal_()
    W, ex, ed = 128, 63, 27
    ref = {k: v.clone() for k, v in src.state_dict().items()}
    for lv in ("nerf_0", "nerf_1"):
        k = f"{lv}.pts_linears.2.weight"                  # the layer behind skip = 1: 63 + 128 columns
        assert ref[k].shape == (W, ex + W)
        ref[k] = torch.cat([ref[k][:, ex:], ref[k][:, :ex]], 1)            # what a [h | gamma(x)] reference would hold
        k = f"{lv}.views_linears.0.weight"
        ref[k] = torch.cat([ref[k][:, W:], ref[k][:, :W]], 1)              # [gamma(d) | feature]
    rep = dst.load_reference_state_dict(ref, key_map=(), skip_concat="hidden_first", views_concat="dir_first")
    assert not rep["missing"] and not rep["unexpected"]
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
    dst.load_reference_state_dict(ref, key_map=())        # default order: those two weights differ, nothing else
    diff = [k for k, a, b in zip(src.state_dict(), src.state_dict().values(), dst.state_dict().values()) if not torch.equal(a, b)]
    assert sorted(diff) == sorted(f"{lv}.{n}" for lv in ("nerf_0", "nerf_1") for n in ("pts_linears.2.weight", "views_linears.0.weight"))
    with pytest.raises(ValueError):
        dst.load_reference_state_dict(ref, key_map=(), skip_concat="sideways")


# ----------------------------------------------------------------------------- bench.py launch contract
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _bench_line(cmd, timeout=300):
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cp = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
    assert cp.returncode == 0 and len(lines) == 1, (cp.returncode, cp.stdout[-800:], cp.stderr[-800:])
    return json.loads(lines[0])


@pytest.mark.parametrize("how", ["plain", "torchrun"])
def test_bench_starts_its_own_ranks_and_reports_the_process_group(how):
    """`python bench.py --gpus 2` started as a PLAIN process (no WORLD_SIZE: how the driver starts the N = 1 run) must bring up
    its own 2 ranks instead of dying on an assertion, and rank 0 must print exactly ONE JSON line; under
    `python -m torch.distributed.run` it must join the ranks it was given.  Run with --fake-render (gloo on CPU, a stub in
    place of the renderer): this checks the launch / sharding / collective / JSON plumbing, not a measurement.  The line
    carries machine-checkable proof of the group (`rccl`: backend, world size, one device per rank, a summed all-reduce) and
    BOTH scaling forms."""
    import socket
    args = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--fake-render", "--cpu-seconds", "0", "--train-steps", "0"]
    if how == "plain":
        line = _bench_line([os.path.join(ROOT, "bench.py")] + args)
        head = "strong"                     # the default headline with --gpus > 1 (round 5): ONE frame sharded over the ranks
    else:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        line = _bench_line(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args + ["--scaling", "weak"])
        head = "weak"
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == head
    assert line["data"].startswith("fake") and line["higher_is_better"] is True and line["unit"] == "Msamples/s"
    r = line["rccl"]
    assert r["backend"] == "gloo" and r["world_size"] == 2 and len(set(r["devices"])) == 2 and r["allreduce_sum_ok"] is True
    assert r["allreduce_ms"] > 0 and r["allreduce_bytes"] > 4_000_000
    m = line["scaling_modes"]
    assert set(m) == {"weak", "strong"} and m["weak"]["frames_per_step"] == 2 and m["strong"]["frames_per_step"] == 1
    assert m["strong"]["gather_equals_single_rank"] is True
    frames = 2 if head == "weak" else 1
    assert line["value"] == m[head]["value"] and line["config"]["frames_per_step"] == frames and line["config"]["keep_weights"] is False
    # value = whole-job samples / time: rays x 256 MLP samples x frames x steps
    n = line["config"]["rays_per_frame"] * line["config"]["mlp_samples_per_ray"]
    assert abs(m["weak"]["value"] - n * 2 / (m["weak"]["ms_per_step"] * 1e-3) / 1e6) <= 0.02 * m["weak"]["value"]
    assert abs(m["strong"]["value"] - n / (m["strong"]["ms_per_step"] * 1e-3) / 1e6) <= 0.02 * m["strong"]["value"]


def test_bench_eight_ranks_strong_scaling_headline_on_gloo():
    """The 8-rank form the driver's scaling bench launches -- `bench.py --gpus 8` -- before an 8-GPU node has ever been seen
    (VERDICT r4 item 7): eight ranks on gloo with the stub renderer.  The headline is the STRONG form (one frame, rays interleaved
    over 8 ranks, per-ray maps all-gathered inside the timed region), the weak value stays in `scaling_modes`; the process group
    is proven (world size 8, eight distinct devices, a summed all-reduce) and the gathered frame equals the frame one rank
    renders alone, bit for bit."""
    line = _bench_line([os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--fake-render",
                        "--cpu-seconds", "0", "--train-steps", "0"], timeout=600)
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["data"].startswith("fake")
    r = line["rccl"]
    assert r["backend"] == "gloo" and r["world_size"] == 8 and len(set(r["devices"])) == 8 and r["allreduce_sum_ok"] is True
    m = line["scaling_modes"]
    assert m["strong"]["frames_per_step"] == 1 and m["weak"]["frames_per_step"] == 8
    assert m["strong"]["gather_equals_single_rank"] is True
    assert line["value"] == m["strong"]["value"] and line["config"]["frames_per_step"] == 1
    assert "interleaved over 8 rank" in line["config"]["parallelism"]


def _reducer8_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from panopticnerf_amd import train
    torch.manual_seed(0)
    net = make_network(NS(D=2, W=128, skips=[], num_classes=3, N_importance=8))
    red = train.GradReducer(net, world)
    # an UNEVEN split: 1003 "rays" interleaved over 8 ranks (ranks 0..2 hold 126, the others 125).  Each rank's loss is its
    # local SUM divided by the rank-mean count R / world, so that the mean over ranks is the batch mean whatever the split
    R = 1003
    x = torch.arange(R, dtype=torch.float64).add(1.0).div(R)
    mine = shard.shard_rays(x[:, None], rank, world)[:, 0]
    coef = float(mine.sum() / (R / world))
    loss = 0.0
    for lv in (1, 0):
        for i, p in enumerate(net.nerf(lv).parameters()):
            loss = loss + (p * (coef * (i + 1) * (lv + 1))).sum()
    loss.backward()
    red.finish()
    want = float(x.mean())                                      # single-rank gradient of the whole batch's mean
    ok = True
    for lv in (1, 0):
        for i, p in enumerate(net.nerf(lv).parameters()):
            ok = ok and torch.allclose(p.grad, torch.full_like(p, want * (i + 1) * (lv + 1)), rtol=1e-5, atol=0)
    ok = ok and mine.shape[0] == (126 if rank < 3 else 125)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_grad_reducer_world8_gloo_uneven_shards():
    """train.GradReducer on EIGHT ranks with an uneven ray split (1003 rays: three ranks hold one ray more): the reduced
    gradient is the single-rank gradient of the whole batch when every rank normalises its local sum by R / world."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_reducer8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(res) == 8 and all(ok for _, ok in res)


def test_bench_refuses_a_rank_count_that_is_not_the_launchers():
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--fake-render", "--cpu-seconds", "0"],
                        capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert cp.returncode != 0 and "--gpus 2" in cp.stderr + cp.stdout


def test_evaluator_refuses_a_class_zero_thing():
    """Panoptic id = class * 1000 + instance on things, class on stuff: a thing of class 0 would be encoded as its bare instance
    index and counted as the stuff class of that number.  The evaluator refuses such a label set up front."""
    from panopticnerf_amd.evaluate import Evaluator
    with pytest.raises(ValueError, match="class 0"):
        Evaluator(n_classes=4, is_thing=[1, 0, 1, 0])
    Evaluator(n_classes=4, is_thing=[0, 1, 1, 0])


def test_train_eval_switch_drops_the_packed_images():
    """Network.train(mode) invalidates the packed-image cache on every train <-> eval switch: an eval-mode
    render after a training phase must never serve an image keyed on tensor versions that HIP-graph replays / .data writes
    do not bump.  The buffers are kept (graph-captured pointers stay valid); only the version stamp goes."""
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network
    net = make_network(NS(N_importance=0, D=2, W=128))
    buf = object()
    net._packed[("fwd", 0, "cuda:0", "bf16", 0)] = (net._version(0), "desc", buf, None, ())
    net.train(True)                                   # already training: nothing to drop
    assert net._packed[("fwd", 0, "cuda:0", "bf16", 0)][0] is not None
    net.eval()
    hit = net._packed[("fwd", 0, "cuda:0", "bf16", 0)]
    assert hit[0] is None and hit[2] is buf
    net._packed[("fwd", 0, "cuda:0", "bf16", 0)] = (net._version(0),) + tuple(hit[1:])
    net.train()
    assert net._packed[("fwd", 0, "cuda:0", "bf16", 0)][0] is None
