"""bench.py -- BASELINE.json's metric on MI355X: Msamples/s (coarse+fine) of the render_rays
path on synthetic KITTI-360-shaped frames (1408x376 rays).

A "step" is one pass of the whole hot path (Renderer.render) over one full frame:
stratified sampler -> [bbox hits/labels] -> coarse MLP + compositing -> [sample_pdf ->
fine MLP + compositing].  Inputs (rays, boxes, packed weights) are resident in HBM before the
timed region.  MLP sample evaluations per ray = N_samples + (N_samples + N_importance).

  python bench.py [--gpus N --steps K --warmup W] [--config 1..5] [--scaling weak|strong]

--gpus N > 1 started as a PLAIN process (no WORLD_SIZE in the environment) launches its own N ranks, one per GPU, under
torch.distributed.run (nccl = RCCL); started by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it
joins the ranks it was given.  Rank 0 prints the ONE JSON line either way.

--config n   BASELINE.json configs[n-1] (panopticnerf_amd/synthetic.py::BASELINE_CONFIGS); default 5 = the
             configuration the metric is quoted on (full panoptic, 64+128 samples, 8x256 MLPs, 45+32 heads, bbox prior).
--scaling    which form the headline `value` is: strong = ONE frame, rays sharded over the ranks (shard.render_sharded), fine-level
             label maps + rgb + depth all-gathered inside the timed region (BASELINE configs[4]: "rays sharded over 8 GPUs";
             the DEFAULT with --gpus > 1: north_star's ">= 6x ray-throughput at 8 GPUs" is a statement about this form);
             weak = every rank renders its own full frame, no collective on the data path (the default on one GPU, where the
             two forms are the same frame).  BOTH forms are timed in every run and reported in `scaling_modes`.

Rank 0 prints ONE JSON line (contract in the task statement) with extra objects:
  roofline                  -- the dominant kernel (fused top-level MLP, MFMA-bound): algorithmic FLOP per launch /
                               mean launch duration from hipEvents on the launch stream (libpnr_bench.so);
  roofline_composite[_coarse] -- the standalone compositing scan (HBM-bound; training / two-kernel path) at the top / coarse level;
  rccl                      -- machine-checkable proof of the process group: backend, world size, one device per rank, the
                               time of the gradient bucket's all-reduce;
  scaling_modes             -- weak and strong whole-job values of this run;
  cpu_baseline              -- the oracle's PyTorch CPU restatement of the same workload timed on the host cores on a
                               bounded ray sample, in 8192-ray chunks (a reported baseline, not the target);
  cpu_baseline_config1      -- BASELINE configs[0] (the reference's CPU-runnable case) on the host;
  train_step                -- secondary: one training step on a ray batch per rank (never the headline value).
"""
import argparse
import json
import os
import sys
import time

# the host driver of the GPU boxes only supports dmabuf IPC: without this, RCCL's (and torch's) cross-process buffer sharing fails with
# `hipIpcGetMemHandle: invalid argument`.  Exported on the boxes already; set here too so that a bare `torchrun bench.py` works
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W_IMG, H_IMG = 1408, 376
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense, MI355X_MICROARCH.md chip table
MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0


def mlp_flops_per_sample(D=8, W=256, skip=4, ex=63, ed=27, n_sem=45, n_inst=32):
    """Algorithmic (unpadded) forward FLOP per sample, SURVEY.md 8d: 2 * MACs."""
    mac = ex * W + (D - 1) * W * W                   # trunk
    if 0 <= skip < D - 1:
        mac += ex * W                                # the skip layer's extra gamma(x) columns
    mac += W + W * W + (W + ed) * (W // 2) + (W // 2) * 3
    for n in (n_sem, n_inst):
        if n:
            mac += W * (W // 2) + (W // 2) * n
    return 2 * mac


def composite_bytes_per_ray(N, C, K, labels, weights):
    """Algorithmic HBM bytes per ray of k_composite: raw + z read, [one int32 label array per labelled
    field read], [weights written], maps written (learned + fixed fields), the ray record."""
    n_lab = ((1 if C else 0) + (1 if K else 0)) if labels else 0
    fixed = (C + K) if labels else 0
    return 4 * N * (4 + C + K + 1) + n_lab * 4 * N + (4 * N if weights else 0) + 4 * (5 + C + K + fixed) + 32


def traffic(kernel, n_rays_launch, config):
    """HBM bytes per launch of `kernel` from the last committed rocprofv3 PMC passes of this same bench
    command (profiles/latest_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs,
    FETCH_SIZE doubled per MI355X_MICROARCH.md).  PMC counters cannot be read from inside the timed
    process, so this is the recorded value for the 65,536-ray fine-level launch of config 5, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as f:
            t = json.load(f)[kernel]
        # recorded for the profiled frame's largest dispatch (rays_per_launch; 65536 in older records); per-ray traffic is constant
        # (records + per-sample quadruples + z), so the timed launch's figure is the recorded one scaled by its ray count
        return int(t["hbm_bytes_per_launch"] * n_rays_launch / t.get("rays_per_launch", 65536)) if config == 5 else None
    except (OSError, KeyError, ValueError):
        return None


def rocprof_recorded(kernel, bytes_per_launch):
    """The rocprofv3 kernel-trace average of a standalone compositing kernel from the last committed profile of this bench
    command (profiles/latest_traffic.json, written by tools/update_traffic.py), quoted beside the run's own hipEvent timing;
    {} when no profile recorded it."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_traffic.json")) as f:
            t = json.load(f)[kernel]
        us = float(t["rocprof_avg_us"])
        return {"rocprof_avg_us_recorded": us, "rocprof_frac_recorded": round(bytes_per_launch / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                "rocprof_source": t.get("rocprof_source")}
    except (OSError, KeyError, ValueError):
        return {}


class LaunchProbe:
    """frame_accounting: stands in for the ctypes library object during ONE untimed frame and brackets every
    launching entry point of libpnr.so with a pair of events on the launch stream (torch's current stream IS the stream the ops
    pass to the library).  pnr_mlp_forward_composite is issued as the two halves it consists of (include/pnr.h:
    pnr_mlp_forward_tiles + pnr_composite_combine, same workspace, same order), so the MLP launch and the per-ray combine are
    timed apart.  The product path is untouched: the probe exists only while bench.py installs it."""
    QUIET = ("_bytes", "_plan", "pnr_version", "pnr_last_error", "pnr_device_check", "pnr_mlp_train_layout", "pnr_mlp_pack")

    def __init__(self, lib):
        self.lib, self.marks = lib, []

    def _timed(self, name, fn, n_samples, *a):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*a)
        e1.record()
        self.marks.append((name, n_samples, e0, e1))
        return rc

    def __getattr__(self, name):
        fn = getattr(self.lib, name)
        if any(q in name for q in self.QUIET):
            return fn
        if name == "pnr_mlp_forward_composite":
            def both(desc, packed, rays, z, R, N, ls, li, white, rgb, depth, acc, w, sem, inst, fs, fi, ws, st):
                rc = self._timed("pnr_mlp_forward_tiles", self.lib.pnr_mlp_forward_tiles, N, desc, packed, rays, z, R, N, ws, st)
                if rc != 0:
                    return rc
                return self._timed("pnr_composite_combine", self.lib.pnr_composite_combine, N, desc, ws, z, R, N, ls, li, white, rgb,
                                   depth, acc, w, sem, inst, fs, fi, st)
            return both
        n_pos = {"pnr_mlp_forward": 5, "pnr_composite": 9, "pnr_sample_pdf": 4, "pnr_sample_pdf_labels": 4, "pnr_ray_setup": 6,
                 "pnr_stratified": 2, "pnr_sample_labels": 2}.get(name)
        return lambda *a: self._timed(name, fn, int(a[n_pos]) if n_pos is not None else 0, *a)


def frame_accounting(frame, n_coarse):
    """One untimed frame under LaunchProbe -> where its time goes: fine / coarse MLP launches, the small kernels by entry point, and
    what is left (torch-side copies / allocations between the library's launches, launch gaps).  Times are hipEvent differences on
    the launch stream: a kernel's figure includes its own launch gap."""
    from panopticnerf_amd import _lib
    lib = _lib.load()
    probe = LaunchProbe(lib)
    _lib._lib = probe
    try:
        with torch.no_grad():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frame()
            host_ms = (time.perf_counter() - t0) * 1e3        # host time to ENQUEUE the frame (no sync inside)
            torch.cuda.synchronize()
    finally:
        _lib._lib = lib
    m = probe.marks
    by = {}
    for name, N, e0, e1 in m:
        key = name[4:] + ("" if not N else "[N=%d]" % N)
        d = by.setdefault(key, [0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
    span = m[0][2].elapsed_time(m[-1][3])
    inside = sum(v[1] for v in by.values())
    mlp = {k: v for k, v in by.items() if k.startswith("mlp_forward")}
    fine = sum(v[1] for k, v in mlp.items() if "[N=%d]" % n_coarse not in k)
    coarse = sum(v[1] for k, v in mlp.items() if "[N=%d]" % n_coarse in k)
    small = {k: {"launches": v[0], "ms": round(v[1], 3)} for k, v in sorted(by.items()) if k not in mlp}
    return {"note": "ONE untimed frame, every libpnr.so launch bracketed by hipEvents on the launch stream (bench.py LaunchProbe)",
            "frame_ms_first_launch_to_last": round(span, 3), "host_enqueue_ms": round(host_ms, 3),
            "library_launches_per_frame": len(m),
            "fine_mlp_ms": round(fine, 3), "coarse_mlp_ms": round(coarse, 3),
            "small_kernels_ms": round(inside - fine - coarse, 3), "small_kernels": small,
            "between_launches_ms": round(span - inside, 3),
            "between_launches_is": "torch-side work between the library's launches (frame-map allocation / first-chunk copy, workspace "
                                   "allocations) and idle queue time"}


def make_train_batch(cfg, rays, box, ids, n_rays, C, K, dev, seed):
    """A training batch whose targets are LEARNABLE: a teacher network (same architecture, different seed) renders the
    batch's rays; its fine-level colour / depth are the rgb / stereo-depth targets and the argmax of its composited
    logits the 2D pseudo labels.  (Round 1 used uniform-random targets: nothing to learn, the loss of a fixed batch
    wandered upwards.)"""
    from panopticnerf_amd import make_network, make_renderer, synthetic
    g = torch.Generator(device=dev).manual_seed(seed)
    idx = torch.randint(0, rays.shape[0], (n_rays,), generator=g, device=dev)
    tb = {"rays": rays[idx][None].contiguous()}
    if box is not None:
        tb.update(bbox=box, bbox_ids=ids)
    state = torch.random.get_rng_state()
    torch.manual_seed(1234)
    teacher = make_network(cfg).eval()
    torch.random.set_rng_state(state)
    synthetic.trained_like_(teacher, 0.06)
    with torch.no_grad():
        t = make_renderer(cfg, teacher.to(dev)).render(tb)
    top = 1 if "rgb_1" in t else 0
    tb["rgb"] = t[f"rgb_{top}"].clone()
    tb["depth"] = t[f"depth_{top}"].clone()
    if C:
        tb["pseudo_label"] = t[f"semantic_{top}"].argmax(-1).int()
    if K:
        tb["instance_label"] = t[f"instance_{top}"].argmax(-1).int()
    return tb


def graph_step_child(args):
    """Child process of the training-step measurement: the same step (NetworkWrapper render + fused losses, backward through
    the HIP kernels, Adam(capturable, fused), in-place repack of both weight images) as panopticnerf_amd.train.GraphedStep runs
    it -- captured into ONE HIP graph, replayed per step after copying the batch into its static buffers; prints
    {"graph_ms": ms per replayed step}.  Separate process: see the call site."""
    from panopticnerf_amd import NetworkWrapper, make_network, synthetic
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    c = synthetic.BASELINE_CONFIGS[args.config]
    cfg = synthetic.baseline_cfg(args.config, precision=args.precision)
    torch.manual_seed(0)
    tnet = make_network(cfg).to(dev).train()
    synthetic.trained_like_(tnet)
    wrap = NetworkWrapper(tnet, cfg)
    opt = torch.optim.Adam(tnet.parameters(), lr=5e-4, capturable=True, fused=True)
    rays = synthetic.camera_rays().to(dev)
    box = ids = None
    if c["bbox"]:
        box, ids = (t.to(dev) for t in synthetic.random_boxes(64, c["num_classes"], max(c["num_instances"], 1)))
    tb = make_train_batch(cfg, rays, box, ids, args.train_rays, c["num_classes"], c["num_instances"], dev, 0)

    from panopticnerf_amd import train as pnr_train
    step = pnr_train.GraphedStep(wrap, opt, tb)          # warm-up on a side stream, undone in place; then the capture
    step(tb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        step(tb)                                         # copies the batch into the static buffers, replays the graph
    torch.cuda.synchronize()
    print(json.dumps({"graph_ms": round((time.perf_counter() - t0) / args.train_steps * 1e3, 3)}), flush=True)


def event_ms(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_child(argv):
    """One timed run of the oracle's CPU path in a FRESH process whose affinity was narrowed to n CPUs before torch (and its
    OpenMP team) started: `taskset`-visible pinning.  Walks 8192-ray chunks in frame order from chunk k0 for `seconds`
    (at least one chunk).  Prints one JSON line."""
    payload, n, k0, seconds, chunk = argv[0], int(argv[1]), int(argv[2]), float(argv[3]), int(argv[4])
    cpus = sorted(os.sched_getaffinity(0))[:n]
    os.sched_setaffinity(0, cpus)
    os.environ["OMP_NUM_THREADS"] = str(n)
    os.environ["MKL_NUM_THREADS"] = str(n)
    import torch as t
    t.set_num_threads(n)
    from oracle import torch_oracle as to
    from panopticnerf_amd import synthetic
    P = t.load(payload)
    c = synthetic.BASELINE_CONFIGS[P["config"]]
    oc = to.mlp_config(D=c["D"], W=c["W"], skips=tuple(c["skips"]), n_sem=c["num_classes"], n_inst=c["num_instances"], head_W=c["W"] // 2)
    Nc, Nf = c["N_samples"], c["N_importance"]
    per_ray = Nc + (Nc + Nf if Nf else 0)
    rays = P["rays"]
    box, ids = (P["box"], P["ids"]) if c["bbox"] else (None, None)
    done, t_cpu, k, first = 0, 0.0, k0, None
    with t.no_grad():
        to.render_rays(P["params"], oc, rays[:64], Nc, Nf, box=box, box_ids=ids)        # thread team up, code paths warm
        while t_cpu < seconds or done == 0:
            sub = rays[k * chunk:(k + 1) * chunk]
            if sub.shape[0] == 0:
                break
            t0 = time.perf_counter()
            ref = to.render_rays(P["params"], oc, sub, Nc, Nf, box=box, box_ids=ids)
            t_cpu += time.perf_counter() - t0
            if first is None and P.get("keep_first"):
                top = 1 if Nf else 0
                t.save({"k": k, "rgb": ref[f"rgb_{top}"]}, payload + ".first")
                first = True
            done += sub.shape[0]
            k += 1
    print(json.dumps({"msamples": done * per_ray / max(t_cpu, 1e-9) / 1e6, "rays": done, "seconds": t_cpu, "threads": t.get_num_threads(),
                      "affinity": len(os.sched_getaffinity(0)), "next_chunk": k,
                      "parallel_info": [l.strip() for l in t.__config__.parallel_info().splitlines() if "threads" in l.lower()][:4]}), flush=True)


def cpu_leg(config, params, rays_c, box, ids, seconds, keep_first, chunk=8192):
    """The oracle's PyTorch CPU path on `config`, time-bounded.  The thread count is PROBED on the chunk size that is then
    timed (8192-ray chunks): each candidate renders one whole chunk in its own pinned process (cpu_child), the
    fastest setting then walks further chunks for `seconds`.  Returns the cpu_baseline object (+ the first chunk's rgb)."""
    import subprocess
    import tempfile
    from panopticnerf_amd import synthetic
    avail = len(os.sched_getaffinity(0))
    cc = synthetic.BASELINE_CONFIGS[config]
    tmp = tempfile.mkdtemp(prefix="pnr_cpu_")
    payload = os.path.join(tmp, "payload.pt")
    torch.save({"config": config, "params": params, "rays": rays_c, "box": box, "ids": ids, "keep_first": keep_first}, payload)

    def child(n, k0, secs):
        cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", payload, str(n), str(k0), str(secs), str(chunk)],
                            capture_output=True, text=True, timeout=900)
        line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
        if not line:
            raise RuntimeError("cpu child failed: " + cp.stderr[-400:])
        return json.loads(line[-1])

    # thread counts up to half of the host's CPUs ("the node's own host cores"), probed in ascending order; the ladder stops once
    # a count is more than 10 % SLOWER than the best so far (on the pool's EPYC 9575F hosts 32 threads already lose to 16, and one
    # 8192-ray chunk at 64 / 128 threads takes 21 s / 60 s: measured once, profiles/README.md round 4 -- not paid in every run)
    cands = sorted({min(k, avail) for k in (8, 16, 32, 64, 128)})
    probe, k, stopped = {}, 0, None
    for n in cands:
        r = child(n, k, 0.0)                   # exactly one chunk
        probe[n] = r
        k = r["next_chunk"]
        if r["msamples"] < 0.9 * max(v["msamples"] for v in probe.values()):
            stopped = "%d threads were %.0f %% slower than the best so far: larger counts not run" % (
                n, 100 * (1 - r["msamples"] / max(v["msamples"] for v in probe.values())))
            break
    cands = [n for n in cands if n in probe]
    best = max(probe, key=lambda n: probe[n]["msamples"])
    tot_rays, tot_s = probe[best]["rays"], probe[best]["seconds"]
    if seconds > tot_s:
        r = child(best, k, seconds - tot_s)
        tot_rays += r["rays"]
        tot_s += r["seconds"]
    per_ray = cc["N_samples"] + (cc["N_samples"] + cc["N_importance"] if cc["N_importance"] else 0)
    first = None
    if keep_first and os.path.exists(payload + ".first"):
        first = torch.load(payload + ".first")
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return {"value": round(tot_rays * per_ray / tot_s / 1e6, 4), "unit": "Msamples/s", "cores": best, "kind": "port",
            "sample": "%d rays of the frame in %d-ray chunks (frame order), %s, fp32, oracle/torch_oracle.py, %.1f s at the fastest of the "
                      "probed thread counts; every run in its own process pinned (sched_setaffinity) to `cores` of the %d host CPUs"
                      % (tot_rays, chunk, cc["name"], tot_s, avail),
            "thread_probe_msamples": {str(n): round(probe[n]["msamples"], 4) for n in cands}, "thread_probe_stopped": stopped,
            "torch_threads": probe[best]["threads"], "parallel_info": probe[best]["parallel_info"], "cpu_model": cpu_model(),
            "host_cpus": avail}, first


def self_launch(args):
    """`python bench.py --gpus N` without a process group in the environment: start the N ranks ourselves."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def train_bytes_per_sample(c, split=False):
    """HBM bytes per sample of the three training kernels: forward writes the saved activations (bf16) + one gate
    bit per ReLU output + raw; the data-gradient pass reads gate bits + d_raw and writes every dY (bf16); the weight-gradient
    kernel reads both sets back (the trunk output h by two jobs since round 6: feature with the alpha row riding in it, and the stacked
    first Linears of the two heads -- i.e. ONE re-read of h; the re-reads of dY_skip, dY_views, gamma(x) and the [rgb, sigma] block that
    the job structure also has, 1.4 KB per sample, are NOT counted as algorithmic)."""
    D, W, H, ch = c["D"], c["W"], c["W"] // 2, 4 + c["num_classes"] + c["num_instances"]
    acts = 2 * (64 + 32 + (D + 1) * W + 3 * H)
    gates = (D * W + 3 * H) // 8
    dys = 2 * ((D + 1) * W + 3 * H + 160)
    h_rereads = 1 if (c["num_classes"] or c["num_instances"]) else 0
    if split:
        return {"forward_train": acts + gates + 4 * ch, "mlp_bwd": gates + 4 * ch + dys, "wgrad": acts + dys + h_rereads * 2 * W}
    return (acts + gates + 4 * ch) + (gates + 4 * ch + dys) + (acts + dys + h_rereads * 2 * W)


def train_kernel_table(tnet, c, rays_b, levels, dev):
    """The three training kernels by themselves at the step's own geometry: hipEvents around 10 launches of
    each, both levels, on the step's ray batch -- {ms (both levels), algorithmic GB, TB/s, frac of the 8 TB/s HBM peak}."""
    from panopticnerf_amd import ops
    split = train_bytes_per_sample(c, split=True)
    ms = {k: 0.0 for k in split}
    S_tot = 0
    with torch.no_grad():
        for lv, N in levels:
            desc, img = tnet.packed(lv, dev, "bf16")
            _, img_b = tnet.packed_bwd(lv, dev)
            z = ops.stratified(rays_b, N)
            R = rays_b.shape[0]
            raw, acts = ops.mlp_forward_train(desc, img, rays_b, z)
            d_raw = torch.randn_like(raw) * 1e-3
            dys = ops.mlp_backward(desc, img_b, d_raw, acts, R, N)
            shapes = {n: p.shape for n, p in tnet.nerf(lv).named_parameters()}
            ms["forward_train"] += event_ms(lambda: ops.mlp_forward_train(desc, img, rays_b, z), 10, 2)
            ms["mlp_bwd"] += event_ms(lambda: ops.mlp_backward(desc, img_b, d_raw, acts, R, N), 10, 2)
            ms["wgrad"] += event_ms(lambda: ops.mlp_wgrad(desc, acts, dys, R * N, shapes), 10, 2)
            S_tot += R * N
    out = {}
    for k in split:
        gb = split[k] * S_tot / 1e9
        tbs = gb / ms[k]                       # GB / ms = TB/s
        out[k] = {"ms": round(ms[k], 4), "GB": round(gb, 3), "TBps": round(tbs, 3), "frac": round(tbs * 1e3 / HBM_PEAK_GBS, 4)}
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-child":
        return cpu_child(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=5, choices=[1, 2, 3, 4, 5], help="BASELINE.json configs[n-1]")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="headline form; default: strong with --gpus > 1, else weak")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--semantic-activation", default="logits", choices=["logits", "softmax"],
                    help="what the learned semantic / instance fields composite (the reference's cfg.semantic_activation)")
    ap.add_argument("--chunk", type=int, default=65536)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget (0 = skip)")
    ap.add_argument("--cpu1-seconds", type=float, default=None,
                    help="time budget of the config-1 CPU leg (default: 10 s with --config 5 or 1, else 0)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--train-steps", type=int, default=None, help="secondary training-step measurement (0 = skip; "
                    "default 3 with --config 5, else 0)")
    ap.add_argument("--train-rays", type=int, default=4096)
    ap.add_argument("--graph-step-child", action="store_true", help=argparse.SUPPRESS)
    # launch / collective plumbing test (tests/test_host.py): no GPU, gloo, a stub in place of the renderer -- never a measurement
    ap.add_argument("--fake-render", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = "strong" if args.gpus > 1 else "weak"
    if args.train_steps is None:
        args.train_steps = 3 if args.config == 5 else 0
    if args.cpu1_seconds is None:
        args.cpu1_seconds = 10.0 if (args.config in (1, 5) and args.cpu_seconds > 0) else 0.0
    if args.graph_step_child:
        return graph_step_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    fake = args.fake_render
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    import torch.distributed as dist
    if fake:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if fake:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from panopticnerf_amd import shard, synthetic
    from panopticnerf_amd.renderer import chunk_plan

    c = synthetic.BASELINE_CONFIGS[args.config]
    N_C, N_F, N_SEM, N_INST = c["N_samples"], c["N_importance"], c["num_classes"], c["num_instances"]
    N_TOP = N_C + N_F
    per_ray = N_C + (N_TOP if N_F else 0)
    top = 1 if N_F else 0
    KEEP_W = False      # fine-level per-sample weights are not written by the timed frames (nothing downstream reads them)
    if fake:
        rays = synthetic.camera_rays()[::64].contiguous()
        net = rend = box = ids = None

        def render_dict(r):        # stub with the renderer's output keys and shapes: exercises sharding / gather / JSON only
            n = r.shape[0]
            out = {f"rgb_{top}": r[:, :3] * 0.5, f"depth_{top}": r[:, 6].clone()}
            if N_SEM:
                out[f"semantic_{top}"] = r[:, :1].repeat(1, N_SEM)
            return out
    else:
        from panopticnerf_amd import benchlib, make_network, make_renderer, ops
        cfg = synthetic.baseline_cfg(args.config, precision=args.precision, chunk_size=args.chunk, keep_weights=KEEP_W,
                                     semantic_activation=args.semantic_activation)
        torch.manual_seed(0)
        net = make_network(cfg).eval()
        synthetic.trained_like_(net)
        net = net.to(dev)
        rend = make_renderer(cfg, net)
        box = ids = None
        if c["bbox"]:
            box, ids = (t.to(dev) for t in synthetic.random_boxes(64, N_SEM, max(N_INST, 1)))
    # weak scaling: every rank renders its own full frame (a different camera yaw); strong: every rank holds the SAME
    # frame and renders its interleaved share of the rays.  Rays are independent: no data-path collective;
    # strong scaling adds the all-gather of the per-ray output maps.
    if not fake:
        rays_weak = synthetic.camera_rays(yaw=0.05 * rank).to(dev)
        rays_strong = rays_weak if world == 1 else synthetic.camera_rays(yaw=0.0).to(dev)
    else:
        rays_weak = rays_strong = rays
    n_rays = rays_weak.shape[0]

    def bdict(r):
        b = {"rays": r}
        if box is not None:
            b.update(bbox=box, bbox_ids=ids)
        return b

    if fake:
        def fake_reduce(local):
            out = {k: v for k, v in local.items() if not k.startswith("semantic")}
            if N_SEM:
                out["semantic_label"] = local[f"semantic_{top}"].argmax(-1).int()
            return out

        def frame_strong():
            return shard.render_sharded(render_dict, rays_strong, rank, world, gather=True, reduce_fn=fake_reduce)

        def frame_weak():
            return render_dict(rays_weak)
    else:
        reduce_fn = shard.label_maps(level=top) if N_SEM else None
        keys = None if N_SEM else (f"rgb_{top}", f"depth_{top}")

        def frame_strong():
            return shard.render_sharded(lambda r: {k: v[0] for k, v in rend.render(bdict(r[None])).items()}, rays_strong, rank, world,
                                        gather=True, keys=keys, reduce_fn=reduce_fn)
        full = bdict(rays_weak.reshape(H_IMG, W_IMG, 8))

        def frame_weak():
            return rend.render(full)

    def sync():
        if world > 1:
            dist.barrier()
        if not fake:
            torch.cuda.synchronize()

    def timed(frame):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        with torch.no_grad():
            for _ in range(args.warmup):
                frame()
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                frame()
            sync()
            dt = time.perf_counter() - t0
        # every rank's own wall time between the two barriers' END points (its GPU work finished somewhere inside): the max is
        # the contract's time; min / max per rank say how uneven the ranks are (power-clock spread across busy GPUs, chunk tails)
        mine_ms = dt
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, mine_ms

    def rank_busy_ms(frame):
        """Per-rank GPU-busy time of ONE step WITHOUT the closing barrier: hipEvents around the step on this rank's stream, all
        ranks started together.  (min, max) over ranks."""
        with torch.no_grad():
            sync()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            frame()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
        if world > 1:
            t = torch.tensor([ms, -ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return round(-float(t[1].item()), 3), round(float(t[0].item()), 3)
        return round(ms, 3), round(ms, 3)

    modes = {}
    order = ["weak", "strong"] if args.scaling == "weak" else ["strong", "weak"]
    for mode in order:          # the headline form first
        fr = frame_weak if mode == "weak" else frame_strong
        dt, _ = timed(fr)
        frames = world if mode == "weak" else 1
        modes[mode] = {"value": round(n_rays * per_ray * frames * args.steps / dt / 1e6, 2), "unit": "Msamples/s",
                       "ms_per_step": round(dt / args.steps * 1e3, 3), "frames_per_step": frames}
        if not fake:
            # where a shortfall against ideal scaling would come from: the spread of the ranks' own busy time
            # for one step, and -- strong form -- the gather's own time
            lo, hi = rank_busy_ms(fr)
            modes[mode].update(rank_busy_ms_min=lo, rank_busy_ms_max=hi)
            if mode == "strong":
                with torch.no_grad():
                    local = shard.render_sharded(lambda r: {k: v[0] for k, v in rend.render(bdict(r[None])).items()}, rays_strong, rank,
                                                 world, gather=False, keys=keys, reduce_fn=reduce_fn)
                    sync()
                    g_ms = event_ms(lambda: shard.gather_maps(local, rays_strong.shape[0], rank, world), 5, 2)
                if world > 1:
                    t = torch.tensor([g_ms], device=dev, dtype=torch.float64)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    g_ms = float(t.item())
                modes[mode].update(gather_ms=round(g_ms, 4), gather_bytes_per_rank=int(sum(v.numel() * v.element_size() for v in local.values())),
                                   chunks_per_rank=len(chunk_plan(local[next(iter(local))].shape[0], args.chunk)))
        if mode == "strong":
            # the gathered frame IS the single-rank frame: rank 0 renders the whole frame alone (outside the timed region, no
            # collective) and compares every gathered map bit for bit
            with torch.no_grad():
                got = frame_strong()
                if rank == 0:
                    if fake:
                        alone = fake_reduce(render_dict(rays_strong))
                    else:
                        alone = {k: v[0] for k, v in rend.render(bdict(rays_strong[None])).items()}
                        alone = reduce_fn(alone) if reduce_fn is not None else {k: alone[k] for k in keys}
                    modes[mode]["gather_equals_single_rank"] = bool(all(torch.equal(got[k], alone[k]) for k in got))
    value, ms_per_step, frames_per_step = (modes[args.scaling][k] for k in ("value", "ms_per_step", "frames_per_step"))

    # ---- the process group, machine-checkable: backend, size, one device per rank, the gradient bucket's all-reduce
    n_bucket = 1_300_000 if fake or net is None else sum(p.numel() for p in net.parameters())
    mine = "cpu:rank%d (fake)" % rank if fake else "cuda:%d %s" % (local_rank, torch.cuda.get_device_name(local_rank))
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, mine)
        bucket = torch.ones(n_bucket, device=dev, dtype=torch.float32)
        for _ in range(3):
            dist.all_reduce(bucket)
        sync()
        t0 = time.perf_counter()
        for _ in range(10):
            dist.all_reduce(bucket)
        sync()
        ar_ms = (time.perf_counter() - t0) / 10 * 1e3
        ok = bool(abs(float(bucket[0]) - float(world) ** 13) <= 1e-3 * float(world) ** 13)
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices": devices,
                "allreduce_ms": round(ar_ms, 4), "allreduce_bytes": 4 * n_bucket, "allreduce_sum_ok": ok}
    else:
        rccl = {"backend": None, "world_size": 1, "devices": [mine], "allreduce_ms": None, "allreduce_bytes": 4 * n_bucket,
                "note": "single rank: no process group was created"}

    roofline = None
    extra = {}
    rays = rays_weak
    if rank == 0 and not args.no_roofline and not fake:
        with torch.no_grad():
            # dominant kernel: the fused top-level MLP on one renderer chunk
            rc = rays[: args.chunk].contiguous()
            Rc = rc.shape[0]
            z = ops.stratified(rc, N_TOP)
            desc, img = net.packed(top, dev)
            # the image the fused pass consumes (its own chunk order), and the descriptor flag of its compositing mode
            fdesc, fimg = net.packed(top, dev, fused=ops.fused_image(rend.sem_mode))
            fdesc = ops.desc_for_mode(fdesc, rend.sem_mode)
            ch = 4 + N_SEM + N_INST
            raw = ops.alloc_raw(ch, Rc * N_TOP, dev)   # as Renderer allocates it
            ops.mlp_forward(desc, img, rc, z, out=raw)                       # fills raw for the compositing measurements below
            # the launch the step actually runs: with the fused compositing epilogue where the renderer uses it
            fused = bool(getattr(rend, "fuse", False)) and ops.fused_supported(fdesc, N_TOP, rend.sem_mode, None)
            if fused:
                benchlib.time_mlp_forward_tiles(fdesc, fimg, rc, z, 1)
                ms, kernel_mhz = benchlib.time_mlp_forward_tiles(fdesc, fimg, rc, z, 5)
            else:
                ms, kernel_mhz = benchlib.time_mlp_forward(desc, img, rc, z, raw, 5)
            # what the matrix pipe of THIS device sustains (register-only MFMA loop): with constant operands, and with
            # random operands that change from MFMA to MFMA (the toggle rate of real data: the chip lowers its clock)
            pk_const, mhz_const = benchlib.probe_mfma_peak(False, 12000, dev)
            pk_rand, mhz_rand = benchlib.probe_mfma_peak(True, 12000, dev)
            S = Rc * N_TOP
            skip = c["skips"][0] if c["skips"] else -1
            flops = S * mlp_flops_per_sample(c["D"], c["W"], skip, 63, 27, N_SEM, N_INST)
            ach = flops / (ms * 1e-3) / 1e12
            peak = MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16" else MFMA_F32_PEAK_TFLOPS
            kname = ("k_mlp_tt<two-tile assembly, fused %scompositing epilogue, plan 2>" % ("softmax " if rend.sem_mode == 1 else "") if fused and fdesc.plan == 2 else
                     "k_mlp_pp<fused %scompositing epilogue, plan %d>" % ("softmax " if rend.sem_mode == 1 else "", fdesc.plan) if fused else
                     "k_mlp_pp" if (ops.default_schedule() != 1 and args.precision == "bf16") else "k_mlp_fused")
            tkey = ("k_mlp_tt_fused" if fdesc.plan == 2 else "k_mlp_pp_fused") if fused else "k_mlp_pp"
            roofline = {"kernel": "%s (%s level, %d rays x %d samples, %dx%d MLP)" % (kname, "fine" if top else "coarse", Rc, N_TOP, c["D"], c["W"]),
                        "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": traffic(tkey, Rc, args.config),
                        "traffic_source": "profiles/latest_traffic.json: rocprofv3 PMC passes of this command, recorded (not measured in this run)",
                        "traffic_note": None,
                        "ms_per_launch": round(ms, 4), "flop_per_launch": flops,
                        "shader_mhz_during_kernel": round(kernel_mhz, 0),
                        "mfma_sustained": {"note": "register-only bf16 MFMA loop on every SIMD of this device, measured in this run",
                                           "constant_operands_tflops": round(pk_const, 1), "constant_operands_mhz": round(mhz_const, 0),
                                           "random_operands_tflops": round(pk_rand, 1), "random_operands_mhz": round(mhz_rand, 0)},
                        "frac_of_sustained_random_operand_peak": round(ach / pk_rand, 4) if (pk_rand > 0 and args.precision == "bf16") else None}
            if fused and roofline["traffic"]:
                # algorithmic HBM bytes of the fused launch: per sample the (lw, r, g, b) quadruple written + z read, per 32-sample tile
                # one record (Q + the C + K logit sums, padded to a multiple of 4 floats) written
                rec_b = 4 * (((1 + N_SEM + N_INST) + 3) // 4 * 4)
                alg = S * (16 + 4) + (S // 32) * rec_b
                roofline["traffic_algorithmic"] = alg
                ratio = roofline["traffic"] / alg
                if fdesc.plan == 2:
                    roofline["traffic_note"] = ("%.2fx the algorithmic %.2f GB: k_mlp_tt requests its weight pieces with the DEFAULT cache policy and the 69 GB "
                                                "L2 -> LDS weight stream stays in the L2.  With `nt` (what k_mlp_pp uses) 20 %% of that stream missed the L2 -- "
                                                "13.8 GB of fabric traffic per launch, 37x the algorithmic bytes -- and the power-limited part answered with a lower "
                                                "clock: same-box A/Bs profiles/r05/r05p, r05q: 10.6-10.7 ms at 1818-1877 MHz without nt, 11.3-11.5 ms at 1773-1798 MHz "
                                                "with it" % (ratio, alg / 1e9))
                else:
                    roofline["traffic_note"] = ("%.1fx the algorithmic %.2f GB: k_mlp_pp requests its weight pieces with the `nt` policy, so ~3 %% of the 69 GB "
                                                "L2 -> LDS weight stream misses the L2 and is re-fetched over the fabric (0.2 TB/s).  Same-box A/Bs: nt "
                                                "10.97-11.08 ms against 11.17 ms with the default policy in round 4 (profiles/r04/r04j), a wash in round 5 "
                                                "(profiles/r05/r05q: 1087-1089 against 1083-1094 Msamples/s); with the default policy the launch moves 1.07x its "
                                                "algorithmic bytes (profiles/r03/r03d)" % (ratio, alg / 1e9))
            if fused or (ops.default_schedule() != 1 and args.precision == "bf16"):
                # the weight stream of the same launch: every workgroup (256 samples: 8 waves x one 32-sample tile in registers)
                # streams the whole packed image L2 -> LDS by LDS-DMA once per group.  NOT a ceiling: reported next to what the
                # path delivers alone (tools/probe/stream_probe.hip: 8 waves per CU streaming an L2-resident image, 83.6 GB/s per CU)
                ibytes = int((fimg if fused else img).numel() * (fimg if fused else img).element_size())
                groups = (S + 255) // 256
                rate = groups * ibytes / (ms * 1e-3) / 1e12
                roofline["weight_stream"] = {"note": "L2 -> LDS weight stream of the same launch (LDS-DMA): packed image bytes x 256-sample groups / launch time; "
                                                     "not a bound of this kernel -- its cost is issue time beside the partner's MFMAs (DESIGN.md 4)",
                                             "image_bytes": ibytes, "groups": groups, "achieved": round(rate, 3), "unit": "TB/s",
                                             "path_alone_tbs": 21.4, "frac_of_path_alone": round(rate / 21.4, 4)}

            # secondary: compositing scan (HBM-bound), algorithmic bytes per SURVEY.md 8d, at the top and the coarse level
            def comp_roofline(N, want_w, tag):
                zz = ops.stratified(rc, N)
                rw = raw if N == N_TOP else ops.alloc_raw(ch, Rc * N, dev)
                if N != N_TOP:
                    rw.copy_(raw[:, : Rc * N])
                ls = li = None
                if c["bbox"]:      # the scene's own per-sample labels (most samples are outside every box: -1)
                    hits = ops.bbox_hits(rc, box, cfg.max_hits if hasattr(cfg, "max_hits") else 8)
                    ls, li = ops.sample_labels(zz, hits[0], hits[1], hits[2], ids)
                    ls, li = (ls if N_SEM else None), (li if N_INST else None)
                cms = event_ms(lambda: ops.composite(rw, zz, rc, N_SEM, N_INST, True, None, ls, li, 0, False, want_w), 20, 3)
                bytes_ray = composite_bytes_per_ray(N, N_SEM, N_INST, c["bbox"] and (N_SEM or N_INST), want_w)
                gbs = Rc * bytes_ray / (cms * 1e-3) / 1e9
                read_gbs = benchlib.probe_raw_read(rw, Rc, N, 5)     # same image, same order, no arithmetic
                out = {"kernel": "k_composite<channel-major> (%s, N=%d, %d channels)" % (tag, N, ch), "bound": "hbm",
                       "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                       "traffic": traffic("k_composite", Rc, args.config) if N == 192 else None,
                       "ms_per_launch": round(cms, 4), "bytes_per_ray": bytes_ray, "timing": "hipEvents around 20 back-to-back launches",
                       "note": "training / two-kernel path only: the fused inference step does not launch it"}
                if args.config == 5 and Rc == 65536:      # what the profile's bench run launched: ANOTHER box's numbers, kept out of
                    rec = rocprof_recorded("k_composite" if N == 192 else "k_composite_coarse", Rc * bytes_ray)   # this run's roofline object
                    if rec:
                        extra.setdefault("recorded_on_builder_box", {"note": "rocprofv3 kernel-trace averages of the committed profile of this bench command, measured on "
                                                                             "the builder's box (another MI355X, another power / clock state) -- NOT this run; this run's own "
                                                                             "numbers are roofline_composite[_coarse].{achieved, frac}"})["k_composite" if N == 192 else "k_composite_coarse"] = rec
                # the pure-read probe walks the image in the mapping pnr_composite uses at this N (4 samples per lane; 8 lanes x 8
                # samples for 32 < N <= 64, round 5): what HBM delivers for the pattern with no arithmetic
                out.update(pure_read_same_pattern_gbs=round(read_gbs, 1), frac_of_pure_read=round(gbs / read_gbs, 4))
                return out

            # weights are written where the renderer needs them: the coarse level of a coarse+fine render (sample_pdf input)
            extra["roofline_composite"] = comp_roofline(N_TOP, not N_F, "top level")
            if N_F:
                extra["roofline_composite_coarse"] = comp_roofline(N_C, True, "coarse level, weights written")

    if rank == 0 and not fake and not args.no_roofline:
        # on the SERIAL frame (every launch on one stream, cfg.overlap_levels = False): per-launch event differences only add up there;
        # the timed frames run the fine level of a chunk beside the coarse level of the next (Renderer._render_overlapped)
        overlapped = bool(getattr(rend, "overlap_levels", False))
        try:
            rend.overlap_levels = False
            extra["frame_accounting"] = frame_accounting(frame_weak, N_C if N_F else -1)
            extra["frame_accounting"]["serial_frame"] = ("cfg.overlap_levels = False for this one frame; the timed frames overlap the levels of "
                                                         "neighbouring chunks" if overlapped else "the timed frames are serial too")
        except Exception as e:      # noqa: BLE001
            extra["frame_accounting"] = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            rend.overlap_levels = overlapped

    # secondary measurement (never the headline value): one training step on a ray batch per rank --
    # render with autograd, the loss wrapper (RGB / depth / 2D CE on learned and fixed fields / 3D CE; fused HIP),
    # backward through the HIP kernels (compositing, dgrad, wgrad), flat-bucket
    # gradient all-reduce over RCCL, Adam.  Guarded: a failure here must not lose the headline line.
    train_info = None
    if args.train_steps > 0 and not fake:
        try:
            from panopticnerf_amd import NetworkWrapper, train as pnr_train
            tnet = make_network(cfg).to(dev).train()
            synthetic.trained_like_(tnet)
            wrap = NetworkWrapper(tnet, cfg)       # the trainer's loss wrapper: render + fused losses
            # fused=True: torch's single-kernel Adam (the foreach default is ~10 small kernels = 0.15 ms of this 8 ms step)
            opt = torch.optim.Adam(tnet.parameters(), lr=5e-4, fused=True)
            tb = make_train_batch(cfg, rays, box, ids, args.train_rays, N_SEM, N_INST, dev, rank)

            reducer = pnr_train.GradReducer(tnet, world)      # world 1: a no-op

            def step(reduce="overlapped"):
                opt.zero_grad(set_to_none=True)
                _, loss, _, _ = wrap(tb)
                loss.backward()
                if reduce == "overlapped":      # the fine NeRF's bucket goes out from its last gradient hook, beside the coarse backward
                    reducer.finish()
                elif reduce == "flat":          # ONE bucket after the whole backward
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
            ar_ms, tdt_flat = None, None
            if world > 1:
                # the same steps with the one-bucket form (hooks removed), so that the line says what the overlap is worth
                reducer.remove()
                step("flat")
                sync()
                t0 = time.perf_counter()
                for _ in range(args.train_steps):
                    step("flat")
                sync()
                tt = torch.tensor([tdt, (time.perf_counter() - t0) / args.train_steps], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tdt, tdt_flat = float(tt[0].item()), float(tt[1].item())
                # the collective alone: ONE flat bucket of every gradient, RCCL over xGMI
                ar_ms = event_ms(lambda: pnr_train.allreduce_grads(tnet, world), 10, 3)
            graph_ms = None
            if world == 1:
                # the same step captured into ONE HIP graph (fresh process: torch's capture wants a network whose autograd
                # nodes were created under the capture-side stream, and a crash there must not lose the headline line)
                import subprocess
                try:
                    cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--graph-step-child", "--train-rays", str(args.train_rays),
                                         "--train-steps", str(max(args.train_steps, 5)), "--precision", args.precision,
                                         "--config", str(args.config)], capture_output=True, text=True, timeout=180)
                    line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
                    graph_ms = json.loads(line[-1])["graph_ms"] if line else "failed: rc %d" % cp.returncode
                except Exception as e:      # noqa: BLE001
                    graph_ms = "failed: %s" % type(e).__name__
            S_step = args.train_rays * per_ray
            skip = c["skips"][0] if c["skips"] else -1
            fwd_flops = mlp_flops_per_sample(c["D"], c["W"], skip, 63, 27, N_SEM, N_INST)
            n_par = sum(p.numel() for p in tnet.parameters())
            try:
                ktable = train_kernel_table(tnet, c, tb["rays"][0].contiguous(), [(0, N_C)] + ([(1, N_TOP)] if N_F else []), dev)
            except Exception as e:      # noqa: BLE001
                ktable = {"error": "%s: %s" % (type(e).__name__, e)}
            bps = train_bytes_per_sample(c)
            tbs = bps * S_step / tdt / 1e12
            train_info = {"ms_per_step": round(tdt * 1e3, 3), "ms_per_step_as_one_hip_graph": graph_ms,
                          "forms": "ms_per_step: the eager loop (wrapper(batch); loss.backward(); optimizer.step()); ms_per_step_as_one_hip_graph: "
                                   "the same step through panopticnerf_amd.train.GraphedStep (bit-identical, tests/test_gpu_backward.py)", "rays_per_rank": args.train_rays,
                          "Msamples_per_s_fwd_bwd": round(S_step * world / tdt / 1e6, 2),
                          "loss_first": round(l0, 5), "loss_last": round(ll.item(), 5),
                          "grad_allreduce": ("one bucket per NeRF (%d fp32 in all), the fine level's launched from its last gradient hook beside the coarse "
                                             "backward, RCCL (nccl): train.GradReducer" % n_par) if world > 1
                                            else "flat bucket of %d fp32, single rank: skipped" % n_par,
                          "ms_per_step_flat_bucket": None if tdt_flat is None else round(tdt_flat * 1e3, 3),
                          "allreduce_ms": None if ar_ms is None else round(ar_ms, 4),
                          "losses": "NetworkWrapper: rgb, depth, semantic/instance 2D CE on learned + fixed fields, 3D CE",
                          "optimizer": "torch.optim.Adam(lr=5e-4, fused=True)",
                          # the three MLP kernels of the step alone (hipEvents, both levels summed; algorithmic bytes; HBM peak 8 TB/s)
                          "kernels": ktable,
                          # forward + data-gradient + weight-gradient GEMMs = 3x the forward's algorithmic FLOPs; the step is
                          # nearer the HBM roof than the MFMA roof, so both fractions are reported
                          "roofline": {"bound": "hbm", "flop_per_sample_fwd_bwd": 3 * fwd_flops,
                                       "achieved_tflops": round(3 * fwd_flops * S_step / tdt / 1e12, 1), "peak_tflops": MFMA_BF16_PEAK_TFLOPS,
                                       "frac_mfma": round(3 * fwd_flops * S_step / tdt / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                       "bytes_per_sample": bps, "achieved": round(tbs * 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(tbs * 1e3 / HBM_PEAK_GBS, 4),
                                       "note": "whole step per rank incl. losses, Adam and host launch gaps; bytes_per_sample = saved "
                                               "activations + gate bits + raw written by the forward, gate bits + d_raw read and dY "
                                               "written by the data-gradient pass, both sets read back by the weight-gradient kernel "
                                               "(DESIGN.md 7)"}}
        except Exception as e:      # noqa: BLE001
            train_info = {"error": "%s: %s" % (type(e).__name__, e)}

    cpu_baseline = None
    # the CPU legs run at N = 1 only (rank 0 would otherwise keep N - 1 idle ranks waiting, on host cores the other ranks share)
    if rank == 0 and world == 1 and not fake and (args.cpu_seconds > 0 or args.cpu1_seconds > 0):
        rays_c = rays.cpu()
        box_c, ids_c = (None, None) if box is None else (box.cpu(), ids.cpu())
        if args.cpu_seconds > 0:
            params = {"coarse": {k: v.detach().cpu() for k, v in net.nerf_0.state_dict().items()},
                      "fine": {k: v.detach().cpu() for k, v in (net.nerf_1 or net.nerf_0).state_dict().items()}}
            try:
                cpu_baseline, first = cpu_leg(args.config, params, rays_c, box_c, ids_c, args.cpu_seconds, True)
                if first is not None:
                    k0 = first["k"]
                    with torch.no_grad():
                        out = rend.render(bdict(rays[k0 * 8192:(k0 + 1) * 8192][None].contiguous()))
                    mse = torch.mean((out[f"rgb_{top}"][0].cpu() - first["rgb"]) ** 2).item()
                    extra["psnr_db_hip_vs_oracle_fp32_render"] = round(-10.0 * torch.log10(torch.tensor(max(mse, 1e-20))).item(), 2)
            except Exception as e:      # noqa: BLE001
                cpu_baseline = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.cpu1_seconds > 0 and args.config != 1:
            # BASELINE configs[0], the reference's own CPU-runnable case (BASELINE.md's CPU-baseline plan): its own small network
            from oracle import torch_oracle as to
            c1 = synthetic.BASELINE_CONFIGS[1]
            p1 = to.init_params(to.mlp_config(D=c1["D"], W=c1["W"], skips=tuple(c1["skips"])), seed=0, sigma_bias=0.03)
            try:
                extra["cpu_baseline_config1"], _ = cpu_leg(1, {"coarse": p1, "fine": p1}, rays_c, None, None, args.cpu1_seconds, False)
            except Exception as e:      # noqa: BLE001
                extra["cpu_baseline_config1"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        if args.scaling == "strong":
            par = "ONE frame, rays interleaved over %d rank(s); per-ray label maps + rgb + depth all-gathered (RCCL) inside the timed region" % world
        else:
            par = "one frame per rank, %d rank(s), no data-path collective" % world
        line = {"metric": "Msamples/sec (coarse+fine), KITTI-360 1408x376", "value": value,
                "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
                "vs_baseline": None, "dtype": args.precision,
                "data": "fake (CPU stub in place of the renderer: launch / collective plumbing test, not a measurement)" if fake else "synthetic",
                "config": {"workload": "BASELINE %s; %dx%d frame, %d%s samples/ray, %dx%d MLP%s, semantic %d / instance %d heads, bbox prior %s"
                                       % (c["name"], W_IMG, H_IMG, N_C, "+%d" % N_F if N_F else "", c["D"], c["W"], "s" if N_F else "",
                                          N_SEM, N_INST, "on" if c["bbox"] else "off"),
                           "baseline_config": args.config, "rays_per_frame": n_rays, "frames_per_step": frames_per_step,
                           "mlp_samples_per_ray": per_ray, "chunk_rays": args.chunk, "keep_weights": KEEP_W,
                           "small_print": "timed frames: rays are pre-generated and resident (pnr_gen_rays, SURVEY 8f-2, is outside the frame); the "
                                          "fine level's per-sample weights are not written (keep_weights false: nothing downstream reads them, ~0.4 GB "
                                          "per frame); inference is deterministic (perturb = 0, no sigma noise) -- in TRAINING the renderer draws "
                                          "perturb / raw_noise_std uniforms with torch.rand / torch.randn (renderer.py), there is no in-kernel RNG; a frame's chunks "
                                          "alternate between two streams so that the fine level of chunk c (3/4 of the compute units) runs beside the coarse "
                                          "level of chunk c + 1 (1/4) -- same kernels, same bits (cfg.overlap_levels, PNR_OVERLAP=0 switches it off)",
                           "semantic_activation": args.semantic_activation, "parallelism": par},
                "scaling_modes": modes, "rccl": rccl,
                "roofline": roofline, "cpu_baseline": cpu_baseline}
        line.update(extra)
        if train_info is not None:
            line["train_step"] = train_info
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
