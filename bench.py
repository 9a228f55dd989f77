"""bench.py -- BASELINE.json's metric on MI355X: Msamples/s (coarse+fine) of the render_rays
path on synthetic KITTI-360-shaped frames (1408x376 rays, 64 + 128 hierarchical samples,
8x256 NeRF MLPs with semantic + instance heads, 3D bbox prior).

A "step" is one pass of the whole hot path (Renderer.render) over one full frame per rank:
stratified sampler -> bbox hits/labels -> coarse MLP -> compositing -> sample_pdf ->
fine MLP -> compositing.  Inputs (rays, boxes, packed weights) are resident in HBM before the
timed region.  MLP sample evaluations per frame = rays * (64 + 192).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (fused fine-level MLP, MFMA-bound): algorithmic FLOP per
                  launch / mean launch duration from hipEvents on the launch stream;
  cpu_baseline -- the oracle's PyTorch CPU restatement of the same path timed on the host
                  cores on a bounded ray sample (a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace as NS

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W_IMG, H_IMG = 1408, 376
N_C, N_F = 64, 128
N_SEM, N_INST = 45, 32           # KITTI-360 label ids 0..44; 32 instance slots (reference values unverifiable)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def mlp_flops_per_sample(D=8, W=256, ex=63, ed=27, n_sem=N_SEM, n_inst=N_INST):
    """Algorithmic (unpadded) forward FLOP per sample, SURVEY.md 8d: 2 * MACs."""
    mac = ex * W + (D - 1) * W * W + ex * W          # trunk incl. the skip layer's extra gamma(x) columns
    mac += W + W * W + (W + ed) * (W // 2) + (W // 2) * 3
    for n in (n_sem, n_inst):
        if n:
            mac += W * (W // 2) + (W // 2) * n
    return 2 * mac


def traffic(kernel, n_rays_launch):
    """HBM bytes per launch of `kernel` from the last committed rocprofv3 PMC passes of this same bench
    command (profiles/latest_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs,
    FETCH_SIZE doubled per MI355X_MICROARCH.md).  PMC counters cannot be read from inside the timed
    process, so this is the recorded value for the 65,536-ray fine-level launch, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as f:
            t = json.load(f)[kernel]
        return int(t["hbm_bytes_per_launch"]) if n_rays_launch == 65536 else None
    except (OSError, KeyError, ValueError):
        return None


def graph_step_child(args):
    """Child process of the training-step measurement: the same step (NetworkWrapper render + fused losses, backward through
    the HIP kernels, Adam(capturable), in-place repack of both weight images) captured into ONE HIP graph; prints
    {"graph_ms": ms per replayed step}.  Separate process: see the call site."""
    from panopticnerf_amd import NetworkWrapper, make_network, synthetic
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = NS(N_samples=N_C, N_importance=N_F, num_classes=N_SEM, num_instances=N_INST, precision=args.precision)
    torch.manual_seed(0)
    tnet = make_network(cfg).to(dev).train()
    synthetic.trained_like_(tnet)
    wrap = NetworkWrapper(tnet, cfg)
    opt = torch.optim.Adam(tnet.parameters(), lr=5e-4, capturable=True)
    g = torch.Generator(device=dev).manual_seed(0)
    rays = synthetic.camera_rays().to(dev)
    box, ids = synthetic.random_boxes(64, N_SEM, N_INST)
    idx = torch.randint(0, rays.shape[0], (args.train_rays,), generator=g, device=dev)
    tb = {"rays": rays[idx][None].contiguous(), "bbox": box.to(dev), "bbox_ids": ids.to(dev),
          "rgb": torch.rand((1, args.train_rays, 3), generator=g, device=dev),
          "depth": torch.rand((1, args.train_rays), generator=g, device=dev) * 60.0 - 10.0,
          "pseudo_label": torch.randint(-1, N_SEM, (1, args.train_rays), generator=g, device=dev),
          "instance_label": torch.randint(-1, N_INST, (1, args.train_rays), generator=g, device=dev)}

    def step():
        opt.zero_grad(set_to_none=False)
        _, loss, _, _ = wrap(tb)
        loss.backward()
        opt.step()
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        graph.replay()
    torch.cuda.synchronize()
    print(json.dumps({"graph_ms": round((time.perf_counter() - t0) / args.train_steps * 1e3, 3)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--chunk", type=int, default=65536)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=3, help="secondary training-step measurement (0 = skip)")
    ap.add_argument("--train-rays", type=int, default=4096)
    ap.add_argument("--graph-step-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.graph_step_child:
        return graph_step_child(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from panopticnerf_amd import make_network, make_renderer, ops, synthetic

    cfg = NS(N_samples=N_C, N_importance=N_F, num_classes=N_SEM, num_instances=N_INST, precision=args.precision,
             chunk_size=args.chunk, keep_weights=False)
    torch.manual_seed(0)
    net = make_network(cfg).eval()
    synthetic.trained_like_(net)
    net = net.to(dev)
    rend = make_renderer(cfg, net)
    # weak scaling: every rank renders its own full frame (a different camera yaw); rays are independent,
    # so there is no data-path collective
    rays = synthetic.camera_rays(yaw=0.05 * rank).to(dev)
    box, ids = synthetic.random_boxes(64, N_SEM, N_INST)
    batch = {"rays": rays.reshape(H_IMG, W_IMG, 8), "bbox": box.to(dev), "bbox_ids": ids.to(dev)}
    n_rays = rays.shape[0]
    samples_per_frame = n_rays * (N_C + (N_C + N_F))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            rend.render(batch)
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            rend.render(batch)
        sync()
        dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = samples_per_frame * world * args.steps / dt / 1e6

    roofline = None
    extra = {}
    if rank == 0 and not args.no_roofline:
        # dominant kernel: the fused fine-level MLP on one renderer chunk
        with torch.no_grad():
            rc = rays[: args.chunk].contiguous()
            z = ops.stratified(rc, N_C + N_F)
            desc, img = net.packed(1, dev)
            raw = ops.alloc_raw(4 + N_SEM + N_INST, rc.shape[0] * (N_C + N_F), dev)   # as Renderer allocates it
            ops.time_mlp_forward(desc, img, rc, z, raw, 1)
            ms, kernel_mhz = ops.time_mlp_forward_clk(desc, img, rc, z, raw, 5)
            # what the matrix pipe of THIS device sustains (register-only MFMA loop): with constant operands, and with
            # random operands that change from MFMA to MFMA (the toggle rate of real data: the chip lowers its clock)
            pk_const, mhz_const = ops.probe_mfma_peak(False, 12000, dev)
            pk_rand, mhz_rand = ops.probe_mfma_peak(True, 12000, dev)
            S = rc.shape[0] * (N_C + N_F)
            flops = S * mlp_flops_per_sample()
            ach = flops / (ms * 1e-3) / 1e12
            peak = MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16" else 157.3
            roofline = {"kernel": "k_mlp_fused (fine level, %d rays x %d samples)" % (rc.shape[0], N_C + N_F),
                        "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": traffic("k_mlp_fused", rc.shape[0]),
                        "ms_per_launch": round(ms, 4), "flop_per_launch": flops,
                        "shader_mhz_during_kernel": round(kernel_mhz, 0),
                        "mfma_sustained": {"note": "register-only bf16 MFMA loop on every SIMD of this device, measured in this run",
                                           "constant_operands_tflops": round(pk_const, 1), "constant_operands_mhz": round(mhz_const, 0),
                                           "random_operands_tflops": round(pk_rand, 1), "random_operands_mhz": round(mhz_rand, 0)},
                        "frac_of_sustained_random_operand_peak": round(ach / pk_rand, 4) if pk_rand > 0 else None}
            # secondary: compositing scan (HBM-bound), algorithmic bytes per SURVEY.md 8d
            lab = torch.zeros((rc.shape[0], N_C + N_F), device=dev, dtype=torch.int32)
            for _ in range(2):
                ops.composite(raw, z, rc, N_SEM, N_INST, True, None, lab, lab, 0, False, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.composite(raw, z, rc, N_SEM, N_INST, True, None, lab, lab, 0, False, True)
            e1.record()
            torch.cuda.synchronize()
            cms = e0.elapsed_time(e1) / 5
            N = N_C + N_F
            ch = 4 + N_SEM + N_INST
            bytes_ray = 4 * N * (ch + 1) + 2 * 4 * N + 4 * N + 4 * (5 + 2 * (N_SEM + N_INST)) + 32
            gbs = rc.shape[0] * bytes_ray / (cms * 1e-3) / 1e9
            read_gbs = ops.probe_raw_read(raw, rc.shape[0], N, 5)     # same image, same order, no arithmetic
            extra["roofline_composite"] = {"kernel": "k_composite<channel-major>", "bound": "hbm",
                                           "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic("k_composite", rc.shape[0]),
                                           "ms_per_launch": round(cms, 4), "bytes_per_ray": bytes_ray,
                                           "pure_read_same_pattern_gbs": round(read_gbs, 1),
                                           "frac_of_pure_read": round(gbs / read_gbs, 4) if read_gbs > 0 else None}

    # secondary measurement (never the headline value): one training step on a 4096-ray batch per rank --
    # render with autograd, the loss wrapper (RGB / depth / 2D CE on learned and fixed fields / 3D CE; fused HIP),
    # backward through the HIP kernels (compositing, dgrad, wgrad), flat-bucket
    # gradient all-reduce over RCCL, Adam.  Guarded: a failure here must not lose the headline line.
    train_info = None
    if args.train_steps > 0:
        try:
            from panopticnerf_amd import NetworkWrapper, train as pnr_train
            tnet = make_network(cfg).to(dev).train()
            synthetic.trained_like_(tnet)
            wrap = NetworkWrapper(tnet, cfg)       # the trainer's loss wrapper: render + fused losses
            opt = torch.optim.Adam(tnet.parameters(), lr=5e-4)
            g = torch.Generator(device=dev).manual_seed(rank)
            idx = torch.randint(0, n_rays, (args.train_rays,), generator=g, device=dev)
            tb = {"rays": rays[idx][None].contiguous(), "bbox": box.to(dev), "bbox_ids": ids.to(dev),
                  "rgb": torch.rand((1, args.train_rays, 3), generator=g, device=dev),
                  "depth": torch.rand((1, args.train_rays), generator=g, device=dev) * 60.0 - 10.0,     # <= 0: no stereo depth
                  "pseudo_label": torch.randint(-1, N_SEM, (1, args.train_rays), generator=g, device=dev),
                  "instance_label": torch.randint(-1, N_INST, (1, args.train_rays), generator=g, device=dev)}

            def step():
                opt.zero_grad(set_to_none=True)
                _, loss, _, _ = wrap(tb)
                loss.backward()
                pnr_train.allreduce_grads(tnet, world)
                opt.step()
                return loss

            l0 = step().item()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.train_steps):
                ll = step()
            sync()
            tdt = (time.perf_counter() - t0) / args.train_steps
            if world > 1:
                tt = torch.tensor([tdt], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tdt = float(tt.item())
            graph_ms = None
            if world == 1:
                # the same step captured into ONE HIP graph (fresh process: torch's capture wants a network whose autograd
                # nodes were created under the capture-side stream, and a crash there must not lose the headline line)
                import subprocess
                try:
                    cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--graph-step-child", "--train-rays", str(args.train_rays),
                                         "--train-steps", str(max(args.train_steps, 5)), "--precision", args.precision],
                                        capture_output=True, text=True, timeout=180)
                    line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
                    graph_ms = json.loads(line[-1])["graph_ms"] if line else "failed: rc %d" % cp.returncode
                except Exception as e:      # noqa: BLE001
                    graph_ms = "failed: %s" % type(e).__name__
            train_info = {"ms_per_step": round(tdt * 1e3, 3), "ms_per_step_as_one_hip_graph": graph_ms, "rays_per_rank": args.train_rays,
                          "Msamples_per_s_fwd_bwd": round(args.train_rays * world * (N_C + N_C + N_F) / tdt / 1e6, 2),
                          "loss_first": round(l0, 5), "loss_last": round(ll.item(), 5),
                          "grad_allreduce": "flat bucket, %s" % ("RCCL (nccl)" if world > 1 else "single rank: skipped"),
                          "losses": "NetworkWrapper: rgb, depth, semantic/instance 2D CE on learned + fixed fields, 3D CE"}
        except Exception as e:      # noqa: BLE001
            train_info = {"error": "%s: %s" % (type(e).__name__, e)}

    cpu_baseline = None
    if rank == 0 and args.cpu_seconds > 0:
        from oracle import torch_oracle as to
        oc = to.mlp_config(n_sem=N_SEM, n_inst=N_INST)
        params = {"coarse": {k: v.detach().cpu() for k, v in net.nerf_0.state_dict().items()},
                  "fine": {k: v.detach().cpu() for k, v in net.nerf_1.state_dict().items()}}
        rays_c = rays.cpu()
        stride = n_rays // 512
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1

        def run(sub):
            t0 = time.perf_counter()
            r = to.render_rays(params, oc, sub, N_C, N_F, box=box, box_ids=ids)
            return r, time.perf_counter() - t0

        # pick the thread count that is fastest on this host (all logical CPUs is often NOT: SMT +
        # oversubscribed OpenMP teams), then spend the time budget at that setting
        probe = rays_c[::stride][:256].contiguous()
        best_n, best_t = 1, float("inf")
        with torch.no_grad():
            for n in sorted({min(c, avail) for c in (8, 16, 32, 64, 128, avail)}):
                torch.set_num_threads(n)
                run(probe[:64])
                _, t = run(probe)
                if t < best_t:
                    best_n, best_t = n, t
                if t > 20.0:
                    break
            torch.set_num_threads(best_n)
            done, t_cpu, k = 0, 0.0, 0
            psnr = None
            while t_cpu < args.cpu_seconds and k < stride:
                sub = rays_c[k::stride][:512].contiguous()
                ref, t = run(sub)
                t_cpu += t
                done += sub.shape[0]
                if k == 0:
                    out = rend.render({"rays": sub[None].to(dev), "bbox": box.to(dev), "bbox_ids": ids.to(dev)})
                    mse = torch.mean((out["rgb_1"][0].cpu() - ref["rgb_1"]) ** 2).item()
                    psnr = -10.0 * torch.log10(torch.tensor(max(mse, 1e-20))).item()
                k += 1
        cpu_val = done * (N_C + N_C + N_F) / t_cpu / 1e6
        cpu_baseline = {"value": round(cpu_val, 4), "unit": "Msamples/s", "cores": torch.get_num_threads(),
                        "kind": "port",
                        "sample": "%d rays of the same frame (every %d-th ray), full coarse+fine path, fp32, "
                                  "oracle/torch_oracle.py, %.1f s, %d of %d host CPUs (fastest setting probed)"
                                  % (done, stride, t_cpu, best_n, avail)}
        extra["psnr_db_hip_vs_oracle_fp32_render"] = None if psnr is None else round(psnr, 2)

    if rank == 0:
        line = {"metric": "Msamples/sec (coarse+fine), KITTI-360 1408x376", "value": round(value, 2),
                "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
                "config": {"workload": "BASELINE configs[4] per GPU: full panoptic (semantic %d + instance %d heads, "
                                       "3D bbox prior), %dx%d frame, %d+%d samples/ray, 8x256 MLPs; one frame per rank"
                                       % (N_SEM, N_INST, W_IMG, H_IMG, N_C, N_F),
                           "rays_per_rank": n_rays, "samples_per_ray": N_C + N_C + N_F, "chunk_rays": args.chunk,
                           "parallelism": "rays sharded, %d rank(s), no data-path collective" % world},
                "roofline": roofline, "cpu_baseline": cpu_baseline}
        line.update(extra)
        if train_info is not None:
            line["train_step"] = train_info
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
