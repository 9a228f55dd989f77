"""Training-side glue of the render path: the autograd Function that
puts the HIP forward / backward kernels behind `Renderer.render`, and the flat-bucket gradient
all-reduce.

What runs where: forward (pnr_mlp_forward_train + pnr_composite), compositing backward
(pnr_composite_backward), the MLP data-gradient pass (pnr_mlp_backward) and the weight gradients
(pnr_mlp_wgrad: dW_l = dY_l^T X_l and the bias sums, MFMA over LDS transpose reads) are hand-written
HIP.  (The same weight gradients computed with library GEMMs on the kernels' buffers live in
tests/_wgrad_ref.py as the cross-check the GPU tests compare pnr_mlp_wgrad against.)
"""
import torch
import torch.distributed as dist

from . import ops


class LevelFn(torch.autograd.Function):
    """One level (coarse or fine) of render_rays: NeRF MLP on every sample + raw2outputs.
    Differentiable w.r.t. the NeRF's parameters only (z comes from the detached sampler).  Besides the maps it
    returns, when the bbox-prior labels are present, the fixed-field maps (differentiable through the weights) and the
    per-sample 3D cross-entropies of the learned fields against those labels."""

    OUT = ("rgb", "depth", "acc", "weights", "semantic", "instance", "fix_semantic", "fix_instance",
           "ce3d_semantic", "ce3d_instance", "ce3d_semantic_n", "ce3d_instance_n")

    @staticmethod
    def forward(ctx, rend, lv, rays, z, ls, li, noise, names, *params):
        net = rend.net
        nerf = net.nerf(lv)
        dev = rays.device
        C, K = nerf.n_sem, nerf.n_inst
        ctx.fp32 = net.precision == "fp32"
        if ctx.fp32:
            # parity mode (pnr_mlp_forward_train_fp32): the live fp32 parameters, nothing packed, plain fp32 arithmetic
            raw, acts = ops.mlp_forward_train_fp32(nerf.desc("fp32"), {n: p.detach() for n, p in nerf.named_parameters()}, rays, z)
        else:
            desc, img = net.packed(lv, dev, "bf16")
            raw, acts = ops.mlp_forward_train(desc, img, rays, z)
        out = ops.composite(raw, z, rays, C, K, True, noise, ls, li, rend.sem_mode, rend.white_bkgd, True)
        empty = torch.zeros(0, device=dev)
        ce_s = ops.ce3d(raw, 4, C, ls) if (C and ls is not None) else None
        ce_i = ops.ce3d(raw, 4 + C, K, li) if (K and li is not None) else None
        ctx.rend, ctx.lv, ctx.names = rend, lv, names
        ctx.set_materialize_grads(False)            # unused outputs arrive as None, not as zero tensors
        ctx.save_for_backward(raw, acts, z, rays, noise if noise is not None else empty,
                              ls if ls is not None else empty, li if li is not None else empty,
                              ce_s if ce_s is not None else empty, ce_i if ce_i is not None else empty)
        ctx.has_noise, ctx.has_ls, ctx.has_li = noise is not None, ls is not None, li is not None
        # views of the (mean, count) pairs pnr_ce3d wrote -- nothing modifies them in place, so no copies (eight small kernels per step)
        out["ce3d_semantic"] = ce_s[0] if ce_s is not None else None
        out["ce3d_instance"] = ce_i[0] if ce_i is not None else None
        out["ce3d_semantic_n"] = ce_s[1] if ce_s is not None else None        # labelled-sample counts: the
        out["ce3d_instance_n"] = ce_i[1] if ce_i is not None else None        # weights of the per-chunk means
        res = [out.get(k) for k in LevelFn.OUT]
        res = tuple(r if r is not None else empty for r in res)
        ctx.mark_non_differentiable(res[10], res[11])
        return res

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_acc, g_w, g_sem, g_inst, g_fs, g_fi, g_ces, g_cei, _g_ns=None, _g_ni=None):
        raw, acts, z, rays, noise, ls, li, ce_s, ce_i = ctx.saved_tensors
        rend, lv = ctx.rend, ctx.lv
        net = rend.net
        nerf = net.nerf(lv)
        C, K = nerf.n_sem, nerf.n_inst
        R, N = z.shape
        grads = {"rgb": g_rgb, "depth": g_depth, "acc": g_acc, "weights": g_w,
                 "semantic": g_sem if C else None, "instance": g_inst if K else None,
                 "fix_semantic": g_fs if (C and ctx.has_ls) else None, "fix_instance": g_fi if (K and ctx.has_li) else None}
        if rend.white_bkgd and g_rgb is not None:       # rgb += 1 - acc
            grads["acc"] = (g_acc if g_acc is not None else 0) - g_rgb.sum(-1)
        grads = {k: v for k, v in grads.items() if v is not None and v.numel()}
        # d(mean CE)/d logits = (softmax - onehot) / count, times the upstream gradient of the scalar
        sc_s = (g_ces / ce_s[1].clamp(min=1.0)) if (g_ces is not None and ce_s.numel()) else None
        sc_i = (g_cei / ce_i[1].clamp(min=1.0)) if (g_cei is not None and ce_i.numel()) else None
        d_raw = ops.composite_backward(raw, z, rays, C, K, grads, noise if ctx.has_noise else None,
                                       ls if ctx.has_ls else None, li if ctx.has_li else None, sc_s, sc_i, rend.sem_mode)
        if ctx.fp32:
            wg = ops.mlp_backward_fp32(nerf.desc("fp32"), {n: p.detach() for n, p in nerf.named_parameters()}, d_raw.contiguous(),
                                       acts, R, N)
        else:
            desc, img_b = net.packed_bwd(lv, rays.device)
            dys = ops.mlp_backward(desc, img_b, d_raw, acts, R, N)
            shapes = {n: p.shape for n, p in nerf.named_parameters()}
            wg = ops.mlp_wgrad(desc, acts, dys, R * N, shapes)      # pnr_mlp_wgrad (cross-check: tests/_wgrad_ref.py)
        return (None,) * 8 + tuple(wg[n].to(p_dtype) for n, p_dtype in ctx.names)


def level_train(rend, lv, rays, z, ls, li, noise):
    """Differentiable level: returns the same dict as ops.composite()."""
    nerf = rend.net.nerf(lv)
    named = list(nerf.named_parameters())
    names = tuple((n, p.dtype) for n, p in named)
    res = LevelFn.apply(rend, lv, rays, z, ls, li, noise, names, *[p for _, p in named])
    return {k: v for k, v in zip(LevelFn.OUT, res) if v.numel()}


def allreduce_grads(module, world=None, group=None):
    """Average the gradients of `module` across ranks with ONE flat bucket (~1.3 M fp32
    = 5 MB per step is latency-bound on xGMI, so one all-reduce beats DDP's many buckets).
    RCCL when the process group is 'nccl' (ROCm), gloo in the CPU tests."""
    world = world or dist.get_world_size(group)
    if world == 1:
        return
    ps = [p for p in module.parameters() if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world)
    o = 0
    for p in ps:
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n


class GradReducer:
    """Gradient averaging overlapped with the backward ("overlap the all-reduce of one level's network with the
    other level's backward").  The two levels of render_rays are independent autograd subgraphs (z of the fine level comes from
    the detached sampler), and the engine runs the fine level's LevelFn.backward first: when the last gradient of the fine NeRF
    has been accumulated its flat bucket goes out as an ASYNCHRONOUS all-reduce (RCCL's own stream under 'nccl'), beside the
    ~2 ms of coarse-level k_mlp_bwd / k_wgrad; only the coarse bucket's collective is exposed.  Two latency-bound collectives of
    ~2.5 MB instead of one of 5 MB.

        reducer = GradReducer(net)           # once; world 1: a no-op
        loss.backward(); reducer.finish(); optimizer.step()

    Several backward() calls before one finish() (gradient accumulation) are reduced correctly but without overlap: a bucket
    that left after the first backward is dropped and everything is reduced in finish().

    Every rank launches the same buckets in the same order: a bucket whose parameters all received a gradient goes out from
    the hook that completes it (fine, then coarse); anything left -- a module with an unused parameter, parameters outside the
    NeRFs -- goes out in finish(), in module order.  For trainers that wrap the network in DistributedDataParallel none of this is needed (DDP's reducer
    does the same job, tests/test_integration.py); allreduce_grads() is the one-bucket form without hooks.

    With GraphedStep (`reduce=reducer.finish`): the hooks also fire while the step is being CAPTURED into a HIP graph, and a
    collective inside the captured region would tie the graph to one communicator state and be replayed beside the eager one --
    so a hook that runs under stream capture only counts (`_capturing`), nothing is launched, and `finish()` -- which GraphedStep
    calls eagerly between its two graphs, where no hook has run -- sends every bucket.  Replayed steps therefore get the
    two-bucket form without the overlap.  `always=True` keeps the hooks and collectives at world size 1 (tests)."""

    def __init__(self, net, world=None, group=None, always=False):
        self.group = group
        self.world = world or (dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1)
        self.buckets, self.works, self.handles = [], [], []
        self.active = self.world > 1 or bool(always)
        if not self.active:
            return
        seen = set()
        levels = [1, 0] if getattr(net, "N_importance", 0) > 0 else [0]
        for lv in levels:                      # fine first: the order the backward completes them in
            m = net.nerf(lv)
            if id(m) in seen:              # cfg.share_coarse_fine: one NeRF, one bucket (autograd sums both levels' contributions
                continue                   # before the parameter's AccumulateGrad -- and with it the hook -- runs once)
            seen.add(id(m))
            ps = [p for p in m.parameters() if p.requires_grad]
            self.buckets.append({"params": ps, "hits": 0, "sent": False})
        inside = {id(p) for b in self.buckets for p in b["params"]}
        rest = [p for p in net.parameters() if p.requires_grad and id(p) not in inside]
        if rest:
            self.buckets.append({"params": rest, "hits": 0, "sent": False})
        for b in self.buckets:
            for p in b["params"]:
                self.handles.append(p.register_post_accumulate_grad_hook(self._hook(b)))

    @staticmethod
    def _capturing():
        """True while the current stream is being captured into a HIP graph (GraphedStep's capture of forward + backward)."""
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def _hook(self, b):
        def fn(_p):
            b["hits"] += 1
            if b["sent"]:
                b["stale"] = True          # a second backward() before finish() (gradient accumulation): what went out is not the sum
            elif b["hits"] == len(b["params"]) and self._in_order(b) and not self._capturing():
                self._send(b)              # (under capture: nothing may be launched from here -- finish() sends, eagerly)
        return fn

    def _in_order(self, b):
        # a bucket may go out from a hook only if every bucket in front of it has: the launch order is the same on every rank
        for o in self.buckets:
            if o is b:
                return True
            if not o["sent"]:
                return False
        return True

    def _send(self, b):
        ps = [p for p in b["params"] if p.grad is not None]
        b["sent"] = True
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        self.works.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat, ps))

    def finish(self):
        """After backward(): launch what no hook launched, wait for every collective, write the means back."""
        if not self.active:
            return
        if self._capturing():
            raise RuntimeError("GradReducer.finish() inside a HIP-graph capture: pass it to GraphedStep as `reduce` (it then runs "
                               "eagerly between the two graphs), do not call it in the captured region")
        if any(b.get("stale") for b in self.buckets):
            # gradients kept accumulating after a bucket had left: drop what is in flight and reduce everything now, in bucket
            # order (every rank takes this branch together: the hooks fire the same way everywhere)
            for work, _flat, _ps in self.works:
                work.wait()
            self.works = []
            for b in self.buckets:
                b["sent"], b["stale"] = False, False
        for b in self.buckets:
            if not b["sent"]:
                self._send(b)
        for work, flat, ps in self.works:
            work.wait()
            flat.div_(self.world)
            o = 0
            for p in ps:
                n = p.grad.numel()
                p.grad.copy_(flat[o:o + n].view_as(p.grad))
                o += n
        self.works = []
        for b in self.buckets:
            b["hits"], b["sent"], b["stale"] = 0, False, False

    def remove(self):
        for h in self.handles:
            h.remove()
        self.handles = []


class GraphedStep:
    """One training step -- `wrapper(batch)`, `loss.backward()`, `optimizer.step()` -- captured ONCE into a HIP graph and
    replayed: the step's ~75 kernel launches (forward, fused losses, compositing backward, k_mlp_bwd, k_wgrad, the in-place
    re-pack of both weight images, the optimiser) become one graph launch, so neither the host nor the gaps between dependent
    launches are in the way (4096 rays x (64 + 192) samples on one MI355X: 7.55 ms replayed vs 8.06 ms eager, bench.py
    `train_step`).  Replays are bit-identical to eager steps (tests/test_gpu_backward.py).

        step = GraphedStep(wrapper, optimizer, example_batch)      # optimizer built with capturable=True (fused=True advised)
        for batch in loader:                                       # same keys, shapes and dtypes as example_batch
            ret, loss, stats = step(batch)                         # static tensors: read (.item(), .clone()) before the next call

    What a HIP graph fixes: shapes (N_rays per step), the set of batch keys, the loss weights, the learning-rate TENSOR (a
    capturable optimiser keeps lr on the device when it is given as a tensor: update it in place for a schedule).  The
    constructor warms up with two real steps on a side stream (first-call allocations, optimiser state) and then restores the
    parameters and the optimiser state IN PLACE (the captured addresses must survive), so constructing a GraphedStep trains
    nothing.  `reduce`: None on one GPU; with several ranks a callable (e.g. `lambda: allreduce_grads(net)` or a
    GradReducer's finish: its hooks launch nothing while a stream is being captured, so every bucket leaves from finish()) that
    runs EAGERLY between two graphs -- forward + backward | collective | optimiser -- since a collective inside a captured
    region ties the graph to one communicator state.  After training through replays call
    `net.eval()` (or `net.invalidate_packed()`) before rendering: tensor versions do not see what a replay did to the weights."""

    def __init__(self, wrapper, optimizer, example_batch, reduce=None, warmup=2):
        for grp in optimizer.param_groups:
            if "capturable" in grp and not grp["capturable"]:
                raise ValueError("GraphedStep: build the optimizer with capturable=True (its step counters must live on the GPU)")
        self.wrapper, self.optimizer, self.reduce = wrapper, optimizer, reduce
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        if not any(torch.is_tensor(v) and v.is_cuda for v in self.static.values()):
            raise RuntimeError("GraphedStep: the batch must be on the GPU")
        params = [p for grp in optimizer.param_groups for p in grp["params"]]
        keep_p = [p.detach().clone() for p in params]
        # optimiser state that exists already (a resumed run) is restored; state the warm-up creates is reset to its initial zeros
        keep_s = {id(v): v.detach().clone() for st in optimizer.state.values() for v in st.values() if torch.is_tensor(v)}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):
                self._fwd_bwd()
                if reduce is not None:
                    reduce()
                optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():      # undo the warm-up steps in place: parameters, moments, step counters
            for p, k in zip(params, keep_p):
                p.copy_(k)
            for st in optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        if id(v) in keep_s:
                            v.copy_(keep_s[id(v)])
                        else:
                            v.zero_()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.graph_opt = None
        # capture_error_mode "thread_local": the default ("global") makes EVERY thread's stream-unsafe call an error while this
        # thread captures -- and a process group's watchdog thread polls the events of the warm-up's collectives on its own
        # schedule: a poll that lands inside the capture kills the process ("operation not permitted when stream is capturing",
        # seen once in round 6 with a one-rank nccl group).  Everything captured here is issued by this thread.
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = self._fwd_bwd()
            if reduce is None:
                optimizer.step()
        if reduce is not None:
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool(), capture_error_mode="thread_local"):
                optimizer.step()

    def _fwd_bwd(self):
        self.optimizer.zero_grad(set_to_none=False)      # the gradient buffers are part of the graph: zeroed, never freed
        ret, loss, stats, _ = self.wrapper(self.static)
        loss.backward()
        return ret, loss, stats

    def __call__(self, batch):
        for k, v in self.static.items():
            if torch.is_tensor(v):
                src = batch[k]
                if src.shape != v.shape or src.dtype != v.dtype:
                    raise ValueError("GraphedStep: batch[%r] is %s %s, the captured step takes %s %s" %
                                     (k, tuple(src.shape), src.dtype, tuple(v.shape), v.dtype))
                v.copy_(src, non_blocking=True)
        self.graph.replay()
        if self.graph_opt is not None:
            self.reduce()
            self.graph_opt.replay()
        return self.out
