"""Renderer / make_renderer -- host-side mirror of lib/networks/renderer (render,
render_rays; the reference source is not in the mount, so
key names follow SURVEY.md 8b and are documented in DESIGN.md).

`Renderer.render(batch)` orchestrates the HIP kernels per ray chunk:
    pnr_stratified -> [pnr_bbox_hits, pnr_sample_labels] -> pnr_mlp_forward(coarse)
    -> pnr_composite -> pnr_sample_pdf -> [labels] -> pnr_mlp_forward(fine) -> pnr_composite
with every intermediate (z, raw, weights) resident in HBM, raw in the channel-major layout
the compositing kernel streams.  No torch math is on the path.

batch keys:  "rays" (B,N_rays,8) = o,d,near,far  (required)
             "bbox" (M,15), "bbox_ids" (M,2) int32  (optional: 3D bbox prior)
             "t_rand" (B,N_rays,N_samples), "u" (B,N_rays,N_importance)  (optional explicit uniforms;
             otherwise drawn with torch.rand on the device when cfg.perturb > 0)
output keys, level l in {0 (coarse), 1 (fine)}:
             rgb_l (B,N_rays,3) depth_l acc_l (B,N_rays) weights_l z_vals_l (B,N_rays,N_l)
             semantic_l / fix_semantic_l (B,N_rays,C), instance_l / fix_instance_l (B,N_rays,K)
"""
import os

import torch

from . import ops


def _get(cfg, name, default):
    return getattr(cfg, name, default) if cfg is not None else default


CHUNK_QUANTUM = 1024     # rays: at 64 and at 192 samples per ray a multiple of 1024 rays is a whole number of rounds of the
                         # MLP kernels' persistent grid (256 workgroups x 256 samples)


PREAMBLE_EARLY = os.environ.get("PNR_OVERLAP_PRE", "1") != "0"     # (A/B) overlapped frames: a chunk's ray_setup in front of the fine MLP two chunks before
CHUNK_TAIL = 8           # a tail of at most chunk_size / CHUNK_TAIL rays joins the other chunks instead of becoming a chunk


def chunk_plan(n_rays, chunk_size):
    """[(start, end)] of the ray chunks of one render() call: BALANCED chunks instead of `chunk_size` pieces plus a tail.
    cfg.chunk_size bounds the working set (raw, acts, weights and the frame maps scale with it): n = ceil(R / chunk_size)
    chunks, except that a tail of at most chunk_size / 8 rays joins the others -- so a chunk never holds more than
    chunk_size * (1 + 1/8) rays plus one rounding quantum -- and every chunk but the last is rounded up to CHUNK_QUANTUM rays
    so that its MLP launches are whole rounds of the persistent grid.  A rank's 66,176-ray share of a 1408 x 376 frame over 8
    ranks is then ONE chunk (not 65,536 + a 640-ray chunk with its own ~14 launches), the full frame 7 x 66,560 + 63,488
    rays (every launch whole rounds) instead of 8 x 65,536 + 5,120."""
    n_rays, chunk_size = int(n_rays), max(1, int(chunk_size))
    if n_rays <= 0:
        return []
    n = -(-n_rays // chunk_size)
    if n > 1 and n_rays - (n - 1) * chunk_size <= chunk_size // CHUNK_TAIL:
        n -= 1
    size = -(-n_rays // n)
    if n > 1 and chunk_size >= CHUNK_QUANTUM * CHUNK_TAIL:
        size = -(-size // CHUNK_QUANTUM) * CHUNK_QUANTUM
    out, s = [], 0
    while s < n_rays:
        e = min(n_rays, s + size)
        out.append((s, e))
        s = e
    return out


class Renderer:
    def __init__(self, net, cfg=None):
        self.net = net
        self.cfg = cfg
        self.N_samples = _get(cfg, "N_samples", 64)
        self.N_importance = _get(cfg, "N_importance", _get(cfg, "cascade_samples", 0))
        self.chunk_size = _get(cfg, "chunk_size", 65536)
        self.perturb = _get(cfg, "perturb", 0.0)
        self.raw_noise_std = _get(cfg, "raw_noise_std", 0.0)
        self.white_bkgd = _get(cfg, "white_bkgd", False)
        self.lindisp = _get(cfg, "lindisp", False)
        self.max_hits = _get(cfg, "max_hits", 8)
        self.sem_mode = {"none": 0, "logits": 0, "softmax": 1}[_get(cfg, "semantic_activation", "none")]
        self.keep_weights = _get(cfg, "keep_weights", True)
        # inference levels run the fused MLP + compositing pass where it applies (bf16, logits compositing, N % 32 == 0);
        # cfg.fuse_composite = False (or PNR_FUSE=0) keeps the two-kernel path with the raw image in HBM
        self.fuse = bool(_get(cfg, "fuse_composite", os.environ.get("PNR_FUSE", "1") != "0"))
        self.strict_hits = bool(_get(cfg, "strict_hits", False))
        # "none": stratified over [near, far] (canonical NeRF); "hull": rays that hit boxes are sampled over the hull of their
        # hit intervals (SURVEY.md 9 item 2 -- which of the two the reference does is unverifiable here: a switch)
        self.bbox_sampling = _get(cfg, "bbox_sampling", "none")
        if self.bbox_sampling not in ("none", "hull"):
            raise ValueError("cfg.bbox_sampling must be 'none' or 'hull', not %r" % (self.bbox_sampling,))
        self._overflow = None
        # inference frames of several chunks are written into frame-sized maps chunk by chunk (no concatenation at the end);
        # cfg.frame_outputs = False (or PNR_FRAME_OUT=0) restores per-chunk maps + torch.cat
        self.frame_outputs = bool(_get(cfg, "frame_outputs", os.environ.get("PNR_FRAME_OUT", "1") != "0"))
        # inference frames of several chunks: the fine level of chunk c runs BESIDE the coarse level of chunk c + 1 (two streams, the
        # two-tile MLP launches capped to 3/4 and 1/4 of the compute units = the levels' 192 : 64 samples), so the small per-ray
        # kernels of one chunk overlap with those of the other instead of standing between the MLP launches (_render_overlapped);
        # cfg.overlap_levels = False (or PNR_OVERLAP=0) keeps every launch of a frame on the caller's stream
        self.overlap_levels = bool(_get(cfg, "overlap_levels", os.environ.get("PNR_OVERLAP", "1") != "0"))
        self._side_streams = {}
        if self.N_importance > 0 and getattr(net, "nerf_1", None) is None and not getattr(net, "share_coarse_fine", False):
            raise ValueError("make_renderer: cfg asks for a fine pass (N_importance / cascade_samples = %d) but the network "
                             "was built without a fine NeRF -- build it with make_network(cfg) from the SAME cfg, or set "
                             "cfg.share_coarse_fine = True to evaluate one NeRF at both levels on purpose" % self.N_importance)

    def _preamble(self, rays, box, box_ids, t_rand, z_out):
        """a chunk's per-ray preamble: (hits or None, z of the coarse level, its labels or None)"""
        hits = lab0 = None
        hull = self.bbox_sampling == "hull"
        if box is not None and self.max_hits <= ops.RAY_SETUP_MAX_HITS:
            # rows a8 + a3 in one launch: hit lists, z and the coarse labels (pnr_ray_setup; bit for bit the separate kernels)
            hits, z, ls0, li0 = ops.ray_setup(rays, box, box_ids, self.N_samples, self.max_hits, self.lindisp, t_rand, hull, out=z_out)
            lab0 = (ls0, li0)
        else:
            if box is not None:
                hits = ops.bbox_hits(rays, box, self.max_hits)
            rays_s = ops.restrict_rays(rays, hits[0], hits[2]) if (hits is not None and hull) else rays
            z = ops.stratified(rays_s, self.N_samples, self.lindisp, t_rand, out=z_out)
        return hits, z, lab0

    # --- one chunk of rays: the reference's render_rays (row a2)
    def render_rays(self, rays, box=None, box_ids=None, t_rand=None, u=None, train=False, grad=False, out=None, sched=None):
        """out: optional {output key: caller-owned tensor of this chunk's shape} (inference only) -- render() passes row slices
        of the frame-sized maps, so the chunks of a frame are never concatenated.  sched: _render_overlapped's hooks for this
        chunk -- {"wait": event the coarse MLP launch waits for, "caps": (coarse, fine) workgroup caps, "pdf_done": filled with the
        event recorded behind sample_pdf}."""
        net, Nc, Nf = self.net, self.N_samples, self.N_importance
        n0 = net.nerf(0)
        C, K = n0.n_sem, n0.n_inst
        dev = rays.device
        ret = {}
        hits = lab0 = None
        if t_rand is None and self.perturb > 0 and train:
            t_rand = torch.rand((rays.shape[0], Nc), device=dev)
        own = (lambda key: out.get(key)) if (out and not grad) else (lambda key: None)
        if sched is not None and sched.get("pre") is not None:
            hits, z, lab0 = sched["pre"]                    # this chunk's preamble was issued earlier on this stream (_render_overlapped)
        else:
            hits, z, lab0 = self._preamble(rays, box, box_ids, t_rand, own("z_vals_0"))
        if hits is not None and self.strict_hits:          # accumulated on the device; checked ONCE at the end of render() (a single sync)
            over = torch.stack([(hits[2] > self.max_hits).sum(), hits[2].max()])
            self._overflow = over if self._overflow is None else torch.stack([self._overflow[0] + over[0],
                                                                              torch.maximum(self._overflow[1], over[1])])

        def level(lv, zz, labels=None):
            ls = li = None
            if labels is not None:
                ls, li = labels
            elif hits is not None:
                ls, li = ops.sample_labels(zz, hits[0], hits[1], hits[2], box_ids)
            noise = None
            if self.raw_noise_std > 0 and train:
                noise = torch.randn(zz.shape, device=dev) * self.raw_noise_std
            if grad:
                from . import train as _train       # autograd path
                res = _train.level_train(self, lv, rays, zz, ls, li, noise)
            else:
                need_w = self.keep_weights or (lv == 0 and Nf > 0)
                mine = {k: own(f"{k}_{lv}") for k in ("rgb", "depth", "acc", "weights", "semantic", "instance", "fix_semantic", "fix_instance")}
                mine = {k: v for k, v in mine.items() if v is not None}
                if self.fuse and ops.fused_supported(net.nerf(lv).desc(net.precision), zz.shape[1], self.sem_mode, noise):
                    # rows a5 + a6 in one pass: no raw image round trip (pnr_mlp_forward_composite, its own chunk order)
                    desc, img = net.packed(lv, dev, fused=ops.fused_image(self.sem_mode))
                    if sched is not None and lv == 0 and sched.get("wait") is not None:
                        torch.cuda.current_stream(dev).wait_event(sched["wait"])      # start beside the other chunk's fine level
                    res = ops.mlp_forward_composite(desc, img, rays, zz, ls, li, self.white_bkgd, need_w, out=mine, sem_mode=self.sem_mode,
                                                    wg_cap=0 if sched is None else sched["caps"][lv])
                else:
                    desc, img = net.packed(lv, dev)
                    raw = ops.mlp_forward(desc, img, rays, zz, channel_major=True)
                    res = ops.composite(raw, zz, rays, C, K, True, noise, ls, li, self.sem_mode, self.white_bkgd, need_w, out=mine)
            for k, v in res.items():
                ret[f"{k}_{lv}"] = v
            ret[f"z_vals_{lv}"] = zz
            return res

        o0 = level(0, z, lab0)
        if Nf > 0:
            if u is None and self.perturb > 0 and train:
                u = torch.rand((rays.shape[0], Nf), device=dev)
            w0 = o0["weights"].detach().contiguous()
            if hits is not None:        # rows a7 + a8 in one launch: the wave that merged a ray's samples labels them
                z_fine, ls1, li1 = ops.sample_pdf_labels(z, w0, Nf, hits, box_ids, u, out=own("z_vals_1"))
                lab1 = (ls1, li1)
            else:
                z_fine, _, _ = ops.sample_pdf(z, w0, Nf, u, want_samples=False, out=own("z_vals_1"))
                lab1 = None
            if sched is not None:
                sched["pdf_done"] = torch.cuda.Event()
                sched["pdf_done"].record(torch.cuda.current_stream(dev))
                if sched.get("before_fine") is not None:
                    sched["before_fine"]()                  # the preamble of the chunk after next, in front of this chunk's fine MLP
            level(1, z_fine, lab1)
        return ret

    def _empty_outputs(self, lead, has_box, dev):
        """render() of zero rays: every output key of a non-empty call, with zero rows and no launch (the edge case a caller
        hits with an empty shard: n_rays < world, or a mask that selects nothing)."""
        n0 = self.net.nerf(0)
        C, K = n0.n_sem, n0.n_inst
        ret = {}
        for lv, N in ((0, self.N_samples),) + (((1, self.N_samples + self.N_importance),) if self.N_importance > 0 else ()):
            need_w = self.keep_weights or (lv == 0 and self.N_importance > 0)
            m = ops._maps(None, 0, N, C, K, True if has_box else None, True if has_box else None, need_w, dev)
            m["z_vals"] = torch.empty((0, N), device=dev, dtype=torch.float32)
            for k, v in m.items():
                ret[f"{k}_{lv}"] = v.reshape(*lead, *v.shape[1:])
        return ret

    # --- inference frames of several chunks: the levels of neighbouring chunks side by side
    def _overlap_caps(self, dev, plan, grad, train, has_box, t_rand, u):
        """(coarse, fine) workgroup caps of the overlapped frame, or None where it does not apply: inference, >= 2 chunks, both levels
        on the two-tile kernel (the only launch that takes PNR_MLP_WG_CAP), frame-sized outputs, and a device whose compute units
        split into two multiples of 8 (one per XCD: the dispatcher deals workgroups round-robin over the 8 XCDs, and a ninth
        workgroup on a 32-CU XCD waits for a whole launch -- 196 + 60 took 25 ms where 192 + 64 takes 14.2, tools/overlap_probe.py) in
        the ratio of the levels' samples within 5 %."""
        if (not self.overlap_levels or grad or train or len(plan) < 2 or not self.fuse or not self.frame_outputs or self.N_importance <= 0
                or self.strict_hits or t_rand is not None or u is not None or self.raw_noise_std > 0 and train
                or torch.cuda.is_current_stream_capturing()):
            return None
        net, Nc, Nt = self.net, self.N_samples, self.N_samples + self.N_importance
        for lv, N in ((0, Nc), (1, Nt)):
            d = net.nerf(lv).desc(net.precision)
            if not ops.fused_supported(d, N, self.sem_mode, None) or net.packed(lv, dev, fused=ops.fused_image(self.sem_mode))[0].plan != 2:
                return None
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        cap_c = int(round(cus * Nc / float(Nc + Nt) / 8.0)) * 8
        cap_f = cus - cap_c
        if cus % 8 or cap_c < 8 or cap_f < 8 or abs((Nc / float(cap_c)) / (Nt / float(cap_f)) - 1.0) > 0.05:
            return None
        return cap_c, cap_f

    def _frame_maps(self, R, has_box, dev):
        """frame-sized output maps, keys and shapes as render_rays returns them"""
        n0 = self.net.nerf(0)
        C, K = n0.n_sem, n0.n_inst
        frame = {}
        for lv, N in ((0, self.N_samples), (1, self.N_samples + self.N_importance)):
            need_w = self.keep_weights or lv == 0
            m = ops._maps(None, R, N, C, K, True if has_box else None, True if has_box else None, need_w, dev)
            m["z_vals"] = torch.empty((R, N), device=dev, dtype=torch.float32)
            for k, v in m.items():
                frame[f"{k}_{lv}"] = v
        return frame

    def _render_overlapped(self, rays, box, box_ids, plan, caps, lead):
        """Chunks alternate between two side streams; a chunk's whole chain (ray_setup, coarse MLP, combine, sample_pdf + labels, fine
        MLP, combine) stays on its stream, and ONE cross-stream rule aligns the pipeline: the coarse MLP of chunk c + 1 (on cap_c
        workgroups) waits for chunk c's sample_pdf, i.e. starts together with chunk c's fine MLP (on cap_f workgroups).  The first
        coarse and the last fine launch have the device to themselves.  Same kernels on the same inputs as the serial frame: every
        map is the same bits (tests/test_gpu_configs.py::test_full_frame_in_one_launch_equals_the_chunked_frame)."""
        dev = rays.device
        R = rays.shape[0]
        cap_c, cap_f = caps
        frame = self._frame_maps(R, box is not None, dev)
        main = torch.cuda.current_stream(dev)
        if dev not in self._side_streams:
            self._side_streams[dev] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        side = self._side_streams[dev]
        start = torch.cuda.Event()
        start.record(main)
        for st in side:
            st.wait_event(start)           # rays, boxes, the frame maps: everything the side streams read or write exists
        prev_pdf = None
        pre = {}

        def preamble(cj):               # issued on chunk cj's stream two chunks early: in front of chunk cj - 2's fine MLP
            sj, ej = plan[cj]
            pre[cj] = self._preamble(rays[sj:ej], box, box_ids, None, frame["z_vals_0"][sj:ej])
        for ci, (s, e) in enumerate(plan):
            sched = {"wait": prev_pdf, "caps": (0 if ci == 0 else cap_c, 0 if ci == len(plan) - 1 else cap_f), "pdf_done": None,
                     "pre": pre.pop(ci, None),
                     "before_fine": (lambda cj=ci + 2: preamble(cj)) if (PREAMBLE_EARLY and ci + 2 < len(plan)) else None}
            with torch.cuda.stream(side[ci & 1]):
                o = self.render_rays(rays[s:e], box, box_ids, None, None, False, False, out={k: v[s:e] for k, v in frame.items()}, sched=sched)
                for k, v in o.items():          # anything a chunk did not write in place (an output without an out= route)
                    if v.data_ptr() != frame[k][s:e].data_ptr():
                        frame[k][s:e].copy_(v)
            prev_pdf = sched["pdf_done"]
        for st in side:
            done = torch.cuda.Event()
            done.record(st)
            main.wait_event(done)
        return {k: v.reshape(*lead, *v.shape[1:]) for k, v in frame.items()}

    # --- the plugin entry point (row a1)
    def render(self, batch):
        rays = batch["rays"]
        if not rays.is_cuda:
            raise RuntimeError("Renderer.render: batch['rays'] must be on the GPU (no CPU fallback)")
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.net.parameters())
        if grad and self.net.precision not in ("bf16", "fp32"):
            raise ValueError("Renderer.render: precision must be 'bf16' (the training path) or 'fp32' (its parity mode)")
        lead = rays.shape[:-1]
        rays = rays.reshape(-1, 8).float().contiguous()
        R = rays.shape[0]
        box = batch.get("bbox")
        box_ids = batch.get("bbox_ids")
        if box is not None:
            box = box.reshape(-1, 15).float().contiguous()
            box_ids = box_ids.reshape(-1, 2).int().contiguous()
        t_rand, u = batch.get("t_rand"), batch.get("u")
        if t_rand is not None:
            t_rand = t_rand.reshape(R, -1).float().contiguous()
        if u is not None:
            u = u.reshape(R, -1).float().contiguous()
        train = self.net.training
        self._overflow = None
        if R == 0:
            return self._empty_outputs(lead, box is not None, rays.device)
        plan = chunk_plan(R, self.chunk_size)
        caps = self._overlap_caps(rays.device, plan, grad, train, box is not None, t_rand, u)
        if caps is not None:
            return self._render_overlapped(rays, box, box_ids, plan, caps, lead)
        outs = []
        frame = None        # inference frames of several chunks: frame-sized maps, every later chunk writes its own rows
        for s, e in plan:
            o = self.render_rays(rays[s:e], box, box_ids,
                                 None if t_rand is None else t_rand[s:e],
                                 None if u is None else u[s:e], train, grad,
                                 out=None if frame is None else {k: v[s:e] for k, v in frame.items()})
            if s == 0 and e < R and not grad and self.frame_outputs and all(v.dim() >= 1 and v.shape[0] == e for v in o.values()):
                frame = {k: torch.empty((R,) + tuple(v.shape[1:]), device=v.device, dtype=v.dtype) for k, v in o.items()}
                for k, v in o.items():
                    frame[k][:e].copy_(v)
            elif frame is not None:
                for k, v in o.items():      # anything a chunk did not write in place (an output without an out= route)
                    if v.data_ptr() != frame[k][s:e].data_ptr():
                        frame[k][s:e].copy_(v)
            outs.append(o)
        if self._overflow is not None:
            n_over, worst = (int(v) for v in self._overflow.tolist())         # the one device sync of strict_hits
            self._overflow = None
            if n_over > 0:
                raise RuntimeError("Renderer.render: %d ray(s) cross more than max_hits = %d boxes (up to %d): the farthest "
                                   "intervals were dropped -- raise cfg.max_hits" % (n_over, self.max_hits, worst))
        if frame is not None:
            return {k: v.reshape(*lead, *v.shape[1:]) for k, v in frame.items()}
        ret = {}
        for k in outs[0]:
            if outs[0][k].dim() == 0:
                # per-level scalars of the training path: ce3d_<field>_<lv> is a mean over the chunk's LABELLED samples,
                # ce3d_<field>_n_<lv> their number -> the frame's mean weights every chunk by its count
                if len(outs) == 1:
                    ret[k] = outs[0][k]
                elif k.rsplit("_", 1)[0].endswith("_n"):
                    ret[k] = torch.stack([o[k] for o in outs]).sum()
                else:
                    head, lv = k.rsplit("_", 1)
                    n = torch.stack([o[f"{head}_n_{lv}"] for o in outs])
                    ret[k] = (torch.stack([o[k] for o in outs]) * n).sum() / n.sum().clamp(min=1.0)
                continue
            v = outs[0][k] if len(outs) == 1 else torch.cat([o[k] for o in outs], 0)
            ret[k] = v.reshape(*lead, *v.shape[1:])
        return ret


def make_renderer(cfg, network):
    """Reference plugin surface: make_renderer(cfg, network) -> obj with .render(batch)."""
    return Renderer(network, cfg)
