"""Multi-GPU readiness: nccl (= RCCL on ROCm) ranks, one process per GPU.

  * sharded render == single-rank render, bit for bit (rays are independent; shard.render_sharded + gather);
  * gradients: the flat-bucket all-reduce (train.allreduce_grads) and an unchanged DistributedDataParallel wrap both
    leave on every rank the gradient of the CONCATENATED batch's mean loss (what a single rank computes on all rays).

world = 2 needs two GPUs (skipped otherwise -- the driver's 8-GPU node runs it); world = 1 runs the very same worker on
one GPU so that the code path is exercised on every box."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

C, K, R = 5, 3, 510          # R divisible by 2 and ragged against the 256-sample groups of the MLP kernel


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup(dev):
    from types import SimpleNamespace as NS
    from panopticnerf_amd import make_network, synthetic
    cfg = NS(N_samples=64, N_importance=128, num_classes=C, num_instances=K, precision="bf16", w_depth=0.0,
             w_sem3d=0.0, w_inst3d=0.0)      # per-sample 3D CE is a mean over each rank's LABELLED samples: not a rank-average
    torch.manual_seed(0)
    net = make_network(cfg)
    synthetic.trained_like_(net, 0.05)
    g = torch.Generator().manual_seed(1)
    rays = synthetic.camera_rays()[:: (1408 * 376) // R][:R].contiguous()
    box, ids = synthetic.random_boxes(24, C, K, seed=3)
    full = {"rays": rays, "rgb": torch.rand(R, 3, generator=g), "pseudo_label": torch.randint(0, C, (R,), generator=g).int(),
            "instance_label": torch.randint(0, K, (R,), generator=g).int()}
    return cfg, net.to(dev), full, box.to(dev), ids.to(dev)


def _batch(full, box, ids, sel, dev):
    b = {k: v[sel][None].to(dev) for k, v in full.items()}
    b.update(bbox=box, bbox_ids=ids)
    return b


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from panopticnerf_amd import NetworkWrapper, make_renderer, shard, train
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        cfg, net, full, box, ids = _setup(dev)
        rays = full["rays"].to(dev)
        ok = {}
        # ---- inference: sharded render + gather == the whole frame rendered by one rank
        rend = make_renderer(cfg, net.eval())
        fn = lambda r: {k: v[0] for k, v in rend.render({"rays": r[None], "bbox": box, "bbox_ids": ids}).items()}
        with torch.no_grad():
            whole = fn(rays)
            got = shard.render_sharded(fn, rays, rank, world, gather=True)
            ok["render"] = set(got) == set(whole) and all(torch.equal(got[k], whole[k]) for k in whole)
            lab = shard.render_sharded(fn, rays, rank, world, gather=True, reduce_fn=shard.label_maps(level=1))
            ok["labels"] = torch.equal(lab["semantic_label"], whole["semantic_1"].argmax(-1).int()) and lab["rgb_1"].shape == (R, 3)
        # ---- training: single-rank gradient of the concatenated batch ...
        net.train()
        wrap = NetworkWrapper(net, cfg)
        net.zero_grad(set_to_none=True)
        wrap(_batch(full, box, ids, slice(None), dev))[1].backward()
        want = {n: p.grad.clone() for n, p in net.named_parameters()}
        # ... equals the rank-average of the per-shard gradients: flat bucket (one RCCL all-reduce)
        sel = shard.shard_indices(R, rank, world)
        net.zero_grad(set_to_none=True)
        wrap(_batch(full, box, ids, sel, dev))[1].backward()
        train.allreduce_grads(net, world)
        err = max(((p.grad - want[n]).norm() / want[n].norm().clamp(min=1e-12)).item() for n, p in net.named_parameters())
        ok["flat_bucket"] = err < (1e-6 if world == 1 else 2e-2)        # world > 1: bf16 kernels on different ray subsets
        ok["flat_bucket_err"] = err
        flat = {n: p.grad.clone() for n, p in net.named_parameters()}
        # ... the overlapped per-level buckets (train.GradReducer: the fine NeRF's all-reduce beside the coarse backward) give
        # the same means as the flat bucket (same kernels on the same shard: only the collective's split differs)
        red = train.GradReducer(net, world)
        net.zero_grad(set_to_none=True)
        wrap(_batch(full, box, ids, sel, dev))[1].backward()
        if world > 1:
            ok["reducer_fine_bucket_sent_in_backward"] = red.buckets[0]["sent"]
        red.finish()
        red.remove()
        err = max(((p.grad - flat[n]).norm() / flat[n].norm().clamp(min=1e-12)).item() for n, p in net.named_parameters())
        ok["reducer"] = err < 1e-6
        ok["reducer_err"] = err
        # ... and an unchanged DDP wrap (the reference trainer's form)
        from torch.nn.parallel import DistributedDataParallel as DDP
        ddp = DDP(wrap, device_ids=[rank])
        net.zero_grad(set_to_none=True)
        ddp(_batch(full, box, ids, sel, dev))[1].backward()
        err = max(((p.grad - want[n]).norm() / want[n].norm().clamp(min=1e-12)).item() for n, p in net.named_parameters())
        ok["ddp"] = err < (1e-6 if world == 1 else 2e-2)
        ok["ddp_err"] = err
        # every rank holds the same averaged gradient
        chk = torch.stack([p.grad.double().sum() for p in net.parameters()]).sum().reshape(1)
        both = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(both, chk)
        ok["same_on_all_ranks"] = all(torch.allclose(b, both[0], rtol=1e-9) for b in both)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, ok))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, {"exception": traceback.format_exc() or repr(e)}))


@pytest.mark.parametrize("world", [1, 2])
def test_nccl_sharded_render_and_gradient_average(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, ok in res:
        assert "exception" not in ok, ok.get("exception")
        bad = {k: v for k, v in ok.items() if v is False}
        assert not bad, (rank, ok)
