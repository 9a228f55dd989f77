"""torch-tensor front ends of the C-ABI kernels (include/pnr.h).

PyTorch is used only for device memory and the current HIP stream: every op takes CUDA
(ROCm) float32 / int32 tensors, checks them, and passes raw pointers + sizes through ctypes.
All ops fail loudly off-GPU; nothing here computes on the CPU.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import MlpDesc, MlpParamsHost


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _on_device(fn):
    """Run `fn` with the device of its first GPU-tensor argument current, so that `_stream()` is THAT device's current
    stream (a tensor on cuda:1 while cuda:0 is current would otherwise be launched on the wrong device's stream)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        cand = []
        for a in list(args) + list(kwargs.values()):
            cand.extend(a.values() if isinstance(a, dict) else (a,))
        for a in cand:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index == torch.cuda.current_device():
                    break
                with torch.cuda.device(a.device):
                    return fn(*args, **kwargs)
        return fn(*args, **kwargs)

    return wrapper


def gen_rays(intr, c2w, width, height, near, far, pix=None, device=None):
    """Pinhole ray generation on the GPU (pnr_gen_rays).  intr: fx, fy, cx, cy; c2w: 3x4 camera-to-world
    (host values); pix: int32 GPU tensor of linear pixel indices or None (whole frame).  Returns rays (R,8)."""
    intr_h = (ctypes.c_float * 4)(*[float(v) for v in torch.as_tensor(intr, dtype=torch.float32).reshape(4).tolist()])
    c2w_h = (ctypes.c_float * 12)(*[float(v) for v in torch.as_tensor(c2w, dtype=torch.float32).reshape(12).tolist()])
    pix = _chk(pix, "pix", torch.int32)
    dev = pix.device if pix is not None else torch.device(device if device is not None else "cuda")
    if dev.type != "cuda":
        raise RuntimeError("gen_rays: expected a GPU device (the HIP path has no CPU fallback)")
    R = pix.numel() if pix is not None else int(width) * int(height)
    rays = torch.empty((R, 8), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().pnr_gen_rays(intr_h, c2w_h, int(width), int(height), float(near), float(far), _p(pix), R,
                                            _p(rays), _stream()), "pnr_gen_rays")
    return rays


def _own(out, shape, dtype, dev, what):
    """`out` (a caller-owned tensor: checked) or a fresh tensor."""
    if out is None:
        return torch.empty(shape, device=dev, dtype=dtype)
    if tuple(out.shape) != tuple(shape) or out.dtype != dtype or out.device != dev or not out.is_contiguous():
        raise ValueError("%s: out must be a contiguous %s tensor of shape %s on %s" % (what, dtype, tuple(shape), dev))
    return out


@_on_device
def stratified(rays, n_samples, lindisp=False, t_rand=None, out=None):
    """rays (R,8) -> z (R,N).  SURVEY 8a row a3."""
    rays = _chk(rays, "rays")
    t_rand = _chk(t_rand, "t_rand")
    R = rays.shape[0]
    assert rays.shape[-1] == 8
    if t_rand is not None:
        assert tuple(t_rand.shape) == (R, n_samples)
    z = _own(out, (R, n_samples), torch.float32, rays.device, "stratified")
    lib = _lib.load()
    _lib.check(lib.pnr_stratified(_p(rays), R, n_samples, int(bool(lindisp)), _p(t_rand), _p(z), _stream()),
               "pnr_stratified")
    return z


@_on_device
def points(rays, z):
    rays, z = _chk(rays, "rays"), _chk(z, "z")
    R, N = z.shape
    pts = torch.empty((R, N, 3), device=z.device, dtype=torch.float32)
    _lib.check(_lib.load().pnr_points(_p(rays), _p(z), R, N, _p(pts), _stream()), "pnr_points")
    return pts


@_on_device
def embed(x, L):
    """x (n,3) -> (n, 3+6L).  SURVEY 8a row a4."""
    x = _chk(x, "x")
    n = x.shape[0]
    out = torch.empty((n, 3 + 6 * L), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().pnr_embed(_p(x), n, L, _p(out), _stream()), "pnr_embed")
    return out


def default_schedule():
    """pnr_mlp_desc.schedule of new descriptors: 0 (ping-pong inference / lock-step training forward) unless the A/B tools'
    environment variable PNR_MLP_VARIANT asks otherwise (its historical values: 0 = lock-step everywhere -> schedule 1,
    1 = the default -> 0, 2 = ping-pong everywhere -> 2).  Read here, on the Python side: libpnr.so keeps no process-global."""
    import os
    return {"0": 1, "1": 0, "2": 2}.get(os.environ.get("PNR_MLP_VARIANT", "1"), 0)


def make_desc(D=8, W=256, skip=4, xyz_L=10, dir_L=4, n_sem=0, n_inst=0, head_W=None, precision="bf16", head_tap="trunk",
              head_depth=2, schedule=None):
    """head_tap: what the semantic / instance heads read -- 'trunk' (the trunk output h) or 'feature' (the feature_linear
    output); head_depth: 2 (W -> head_W -> n) or 1 (one Linear W -> n).  SURVEY.md 9 item 4 as switches.
    schedule: pnr_mlp_desc.schedule (tests / A/B tools: 1 = lock-step kernels everywhere, 2 = ping-pong everywhere)."""
    d = MlpDesc()
    d.D, d.W, d.skip, d.xyz_L, d.dir_L = D, W, skip, xyz_L, dir_L
    d.n_sem, d.n_inst = n_sem, n_inst
    d.head_W = W // 2 if head_W is None else head_W
    d.precision = {"bf16": _lib.PREC_BF16, "fp32": _lib.PREC_FP32}[precision]
    d.head_tap = {"trunk": 0, "feature": 1}[head_tap]
    d.head_depth = {1: 1, 2: 2}[int(head_depth)]
    d.schedule = default_schedule() if schedule is None else int(schedule)
    return d


def fused_plan(desc, limit=None):
    """desc.plan to use for an image that only mlp_forward_composite will consume (pnr_mlp_fused_plan): 2 where the geometry has
    the two-tile assembly kernel, 1 where it has the fused-inference chunk order of the 8-wave kernel, else 0.  limit (or the
    environment's PNR_FUSED_PLAN, for A/B runs) caps it."""
    best = int(_lib.load().pnr_mlp_fused_plan(ctypes.byref(desc)))
    cap = limit if limit is not None else os.environ.get("PNR_FUSED_PLAN")
    if cap is None or int(cap) >= best:
        return best
    # below the best plan: plan 1 where the library takes it (a semantic head of depth 2: the merged logit chunk), else the classic order
    if int(cap) >= 1:
        d1 = _lib.MlpDesc()
        ctypes.memmove(ctypes.byref(d1), ctypes.byref(desc), ctypes.sizeof(d1))
        d1.plan = 1
        if int(_lib.load().pnr_mlp_packed_bytes(ctypes.byref(d1))) > 0:
            return 1
    return 0


def _param_struct(desc, params, device):
    """pnr_mlp_params_host filled with pointers to `params` (dict name -> tensor) moved/kept on `device`.
    Returns (struct, keep-alive list)."""
    keep = []

    def fp(name):
        if name not in params:
            return None
        t = params[name].detach()
        if t.device != torch.device(device) or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(device, torch.float32).contiguous()
        keep.append(t)
        return ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_float))

    P = MlpParamsHost()
    D = desc.D
    pw = (ctypes.POINTER(ctypes.c_float) * D)(*[fp(f"pts_linears.{i}.weight") for i in range(D)])
    pb = (ctypes.POINTER(ctypes.c_float) * D)(*[fp(f"pts_linears.{i}.bias") for i in range(D)])
    keep += [pw, pb]
    P.pts_w, P.pts_b = pw, pb
    P.alpha_w, P.alpha_b = fp("alpha_linear.weight"), fp("alpha_linear.bias")
    P.feature_w, P.feature_b = fp("feature_linear.weight"), fp("feature_linear.bias")
    P.views_w, P.views_b = fp("views_linears.0.weight"), fp("views_linears.0.bias")
    P.rgb_w, P.rgb_b = fp("rgb_linear.weight"), fp("rgb_linear.bias")
    last = 0 if desc.head_depth == 1 else 1            # head_depth 1: the head is its single Linear (the struct's *1 slot)
    if desc.n_sem:
        if last:
            P.sem0_w, P.sem0_b = fp("semantic_linears.0.weight"), fp("semantic_linears.0.bias")
        P.sem1_w, P.sem1_b = fp(f"semantic_linears.{last}.weight"), fp(f"semantic_linears.{last}.bias")
    if desc.n_inst:
        if last:
            P.inst0_w, P.inst0_b = fp("instance_linears.0.weight"), fp("instance_linears.0.bias")
        P.inst1_w, P.inst1_b = fp(f"instance_linears.{last}.weight"), fp(f"instance_linears.{last}.bias")
    return P, keep


@_on_device
def pack_mlp_device(desc, params, backward=False, out=None, workspace=None, repack=False):
    """Pack on the GPU straight from the (CUDA, fp32) parameter tensors: pnr_mlp_pack_device.
    Returns (image uint8 CUDA tensor, workspace) -- pass both back in to reuse the buffers.
    repack=True: `out` / `workspace` come from an earlier call with the SAME parameter tensors (same data pointers) whose
    values changed in place: only the packing kernel runs (pnr_mlp_repack_device; no host copies, graph-capture safe)."""
    lib = _lib.load()
    dev = next(iter(params.values())).device
    if dev.type != "cuda":
        raise RuntimeError("pack_mlp_device: parameters must be on the GPU")
    nbytes = (lib.pnr_mlp_bwd_packed_bytes if backward else lib.pnr_mlp_packed_bytes)(ctypes.byref(desc))
    if nbytes < 0:
        _lib.check(int(nbytes), "pnr_mlp_packed_bytes")
    wbytes = lib.pnr_mlp_pack_workspace_bytes(ctypes.byref(desc), int(backward))
    if out is None or out.numel() != nbytes:
        out = torch.zeros(int(nbytes), dtype=torch.uint8, device=dev)     # zeros: the table->data alignment gap
    if workspace is None or workspace.numel() < wbytes:
        workspace = torch.empty(int(wbytes), dtype=torch.uint8, device=dev)
    P, keep = _param_struct(desc, params, dev)
    fn = lib.pnr_mlp_repack_device if repack else lib.pnr_mlp_pack_device
    _lib.check(fn(ctypes.byref(desc), ctypes.byref(P), int(backward), _p(workspace), _p(out), _stream()), "pnr_mlp_pack_device")
    return out, workspace


def pack_mlp(desc, params):
    """params: dict name -> float32 tensor (canonical NeRF names, see network.py).
    Returns the packed image as a CPU uint8 tensor (pure host work, no GPU needed)."""
    lib = _lib.load()
    nbytes = lib.pnr_mlp_packed_bytes(ctypes.byref(desc))
    if nbytes < 0:
        _lib.check(int(nbytes), "pnr_mlp_packed_bytes")
    P, keep = _param_struct(desc, params, "cpu")
    img = torch.empty(int(nbytes), dtype=torch.uint8)
    _lib.check(lib.pnr_mlp_pack(ctypes.byref(desc), ctypes.byref(P), ctypes.c_void_p(img.data_ptr())),
               "pnr_mlp_pack")
    return img


def pack_mlp_bwd(desc, params):
    """Transposed-weight image for pnr_mlp_backward (CPU uint8 tensor).  bf16, n_sem/n_inst <= 64."""
    lib = _lib.load()
    nbytes = lib.pnr_mlp_bwd_packed_bytes(ctypes.byref(desc))
    if nbytes < 0:
        _lib.check(int(nbytes), "pnr_mlp_bwd_packed_bytes")
    P, keep = _param_struct(desc, params, "cpu")
    img = torch.empty(int(nbytes), dtype=torch.uint8)
    _lib.check(lib.pnr_mlp_pack_bwd(ctypes.byref(desc), ctypes.byref(P), ctypes.c_void_p(img.data_ptr())),
               "pnr_mlp_pack_bwd")
    return img


def train_layout(desc, n_samples):
    """(acts_off, dys_off): element offsets (bf16 units) of the saved-activation / dY regions; last = total."""
    a = (ctypes.c_int64 * (desc.D + 7))()
    d = (ctypes.c_int64 * (desc.D + 8))()
    _lib.check(_lib.load().pnr_mlp_train_layout(ctypes.byref(desc), int(n_samples), a, d), "pnr_mlp_train_layout")
    return list(a), list(d)


@_on_device
def mlp_forward_train(desc, packed, rays, z):
    """Forward that also saves activations for the backward.  Returns (raw (ch,S) channel-major, acts bf16)."""
    rays, z = _chk(rays, "rays"), _chk(z, "z")
    packed = _chk(packed, "packed", torch.uint8)
    R, N = z.shape
    S = R * N
    acts_off, _ = train_layout(desc, S)
    raw = torch.empty((n_channels(desc), S), device=z.device, dtype=torch.float32)
    acts = torch.empty((acts_off[-1],), device=z.device, dtype=torch.bfloat16)
    _lib.check(_lib.load().pnr_mlp_forward_train(ctypes.byref(desc), _p(packed), _p(rays), _p(z), R, N, _p(raw), 1, S,
                                                 _p(acts), _stream()), "pnr_mlp_forward_train")
    return raw, acts


@_on_device
def mlp_backward(desc, packed_bwd, d_raw, acts, n_rays, n_samples):
    """Data-gradient pass: d_raw (ch,S) + saved activations -> dys (bf16, every layer's pre-activation gradient)."""
    d_raw = _chk(d_raw, "d_raw")
    packed_bwd = _chk(packed_bwd, "packed_bwd", torch.uint8)
    acts = _chk(acts, "acts", torch.bfloat16)
    _, dys_off = train_layout(desc, n_rays * n_samples)
    dys = torch.empty((dys_off[-1],), device=d_raw.device, dtype=torch.bfloat16)
    _lib.check(_lib.load().pnr_mlp_backward(ctypes.byref(desc), _p(packed_bwd), _p(d_raw), _p(acts), _p(dys), n_rays,
                                            n_samples, _stream()), "pnr_mlp_backward")
    return dys


@_on_device
def mlp_wgrad(desc, acts, dys, n_samples, shapes):
    """Weight gradients of every Linear from the training forward's `acts` and the data-gradient pass's `dys`
    (pnr_mlp_wgrad: hand-written MFMA kernel + deterministic slab reduction).  shapes: dict name -> shape of the
    parameters (state_dict names).  Returns dict name -> fp32 gradient tensor.  SURVEY 8a row a9."""
    acts, dys = _chk(acts, "acts", torch.bfloat16), _chk(dys, "dys", torch.bfloat16)
    lib = _lib.load()
    dev = acts.device
    nbytes = lib.pnr_mlp_wgrad_workspace_bytes(ctypes.byref(desc), int(n_samples))
    if nbytes < 0:
        raise RuntimeError("pnr_mlp_wgrad_workspace_bytes: " + lib.pnr_last_error().decode(errors="replace"))
    # per call, from torch's caching allocator: stream-ordered and graph-pool aware, so a captured step keeps its own
    # block alive and concurrent streams never share scratch (a process-global buffer did neither)
    ws = torch.empty(int(nbytes), device=dev, dtype=torch.uint8)
    grads = {k: torch.empty(tuple(shp), device=dev, dtype=torch.float32) for k, shp in shapes.items()}
    G, keep = _param_struct(desc, grads, dev)
    _lib.check(lib.pnr_mlp_wgrad(ctypes.byref(desc), _p(acts), _p(dys), int(n_samples), ctypes.byref(G), _p(ws), _stream()),
               "pnr_mlp_wgrad")
    return grads


@_on_device
def mlp_forward_train_fp32(desc, params, rays, z):
    """fp32 parity mode of the training forward (pnr_mlp_forward_train_fp32): params = dict name -> fp32 GPU parameter
    tensor (read in place, nothing is packed).  Returns (raw (ch,S) channel-major, acts fp32)."""
    rays, z = _chk(rays, "rays"), _chk(z, "z")
    R, N = z.shape
    S = R * N
    lib = _lib.load()
    n = lib.pnr_mlp_fp32_acts_floats(ctypes.byref(desc), S)
    if n < 0:
        raise RuntimeError("pnr_mlp_fp32_acts_floats: " + lib.pnr_last_error().decode(errors="replace"))
    raw = torch.empty((n_channels(desc), S), device=z.device, dtype=torch.float32)
    acts = torch.empty((int(n),), device=z.device, dtype=torch.float32)
    P, keep = _param_struct(desc, params, z.device)
    _lib.check(lib.pnr_mlp_forward_train_fp32(ctypes.byref(desc), ctypes.byref(P), _p(rays), _p(z), R, N, _p(raw), 1, S, _p(acts),
                                              _stream()), "pnr_mlp_forward_train_fp32")
    return raw, acts


@_on_device
def mlp_backward_fp32(desc, params, d_raw, acts, n_rays, n_samples):
    """fp32 parity mode of the data-gradient + weight-gradient passes (pnr_mlp_backward_fp32).  d_raw (ch,S) channel-major
    fp32.  Returns dict name -> fp32 gradient tensor for every parameter in `params`."""
    d_raw, acts = _chk(d_raw, "d_raw"), _chk(acts, "acts")
    S = int(n_rays) * int(n_samples)
    if tuple(d_raw.shape) != (n_channels(desc), S):
        raise ValueError(f"d_raw: expected {(n_channels(desc), S)}, got {tuple(d_raw.shape)}")
    lib = _lib.load()
    dev = acts.device
    nbytes = lib.pnr_mlp_backward_fp32_workspace_bytes(ctypes.byref(desc), S)
    if nbytes < 0:
        raise RuntimeError("pnr_mlp_backward_fp32_workspace_bytes: " + lib.pnr_last_error().decode(errors="replace"))
    ws = torch.empty(int(nbytes), device=dev, dtype=torch.uint8)
    grads = {k: torch.empty_like(v, dtype=torch.float32, memory_format=torch.contiguous_format) for k, v in params.items()}
    P, keep = _param_struct(desc, params, dev)
    G, keep2 = _param_struct(desc, grads, dev)
    _lib.check(lib.pnr_mlp_backward_fp32(ctypes.byref(desc), ctypes.byref(P), _p(d_raw), d_raw.stride(0), _p(acts), int(n_rays),
                                         int(n_samples), ctypes.byref(G), _p(ws), _stream()), "pnr_mlp_backward_fp32")
    return grads


def n_channels(desc):
    return 4 + desc.n_sem + desc.n_inst


RAW_PAD = 64      # floats between the channel rows of a channel-major raw image (see alloc_raw)


def alloc_raw(ch, S, device, pad=RAW_PAD):
    """Channel-major raw image (ch, S) whose channel rows are S + pad floats apart.  With the dense stride the 81
    rows of a ray sit exactly S*4 B apart (48 MiB at the fine level) and the 8 row loads of a compositing batch hit
    the same HBM channel; a 256 B skew per row is worth +3 % compositing bandwidth (tools/stride_probe.py)."""
    buf = torch.empty(ch * (S + pad), device=device, dtype=torch.float32)
    return buf.as_strided((ch, S), (S + pad, 1))


def _chk_raw(raw, ch, S):
    """channel-major raw: (ch, S) fp32 on the GPU, unit sample stride, any channel stride >= S"""
    if not raw.is_cuda or raw.dtype != torch.float32:
        raise TypeError("raw: expected a float32 GPU tensor")
    if tuple(raw.shape) != (ch, S) or raw.stride(1) != 1 or raw.stride(0) < S:
        raise ValueError(f"raw: expected shape {(ch, S)} with unit sample stride, got {tuple(raw.shape)} strides {raw.stride()}")
    return raw.stride(0)


@_on_device
def mlp_forward(desc, packed, rays, z, channel_major=True, out=None):
    """Fused gamma() + NeRF MLP + heads on every sample.  SURVEY 8a rows a4+a5.
    Returns raw as (ch, R*N) when channel_major (fast layout; channel stride padded, see alloc_raw) else (R, N, ch)."""
    rays, z = _chk(rays, "rays"), _chk(z, "z")
    packed = _chk(packed, "packed", torch.uint8)
    R, N = z.shape
    ch = n_channels(desc)
    S = R * N
    if channel_major:
        raw = out if out is not None else alloc_raw(ch, S, z.device)
        ss, sc = 1, _chk_raw(raw, ch, S)
    else:
        raw = out if out is not None else torch.empty((R, N, ch), device=z.device, dtype=torch.float32)
        ss, sc = ch, 1
        _chk(raw, "raw")
    _lib.check(_lib.load().pnr_mlp_forward(ctypes.byref(desc), _p(packed), _p(rays), _p(z), R, N, _p(raw), ss, sc,
                                           _stream()), "pnr_mlp_forward")
    return raw


@_on_device
def composite(raw, z, rays, n_sem=0, n_inst=0, channel_major=True, noise=None, label_sem=None, label_inst=None,
              sem_mode=0, white_bkgd=False, want_weights=True, out=None):
    """raw2outputs.  SURVEY 8a row a6.  Returns dict of maps (out: optional caller-owned tensors, see _maps)."""
    z, rays = _chk(z, "z"), _chk(rays, "rays")
    noise = _chk(noise, "noise")
    label_sem = _chk(label_sem, "label_sem", torch.int32)
    label_inst = _chk(label_inst, "label_inst", torch.int32)
    R, N = z.shape
    ch = 4 + n_sem + n_inst
    if channel_major:
        ss, sc = 1, _chk_raw(raw, ch, R * N)
    else:
        _chk(raw, "raw")
        assert tuple(raw.shape) == (R, N, ch)
        ss, sc = ch, 1
    dev = z.device
    out = _maps(out, R, N, n_sem, n_inst, label_sem, label_inst, want_weights, dev)
    g = out.get
    _lib.check(_lib.load().pnr_composite(_p(raw), ss, sc, _p(z), _p(rays), _p(noise), _p(label_sem), _p(label_inst),
                                         R, N, n_sem, n_inst, int(sem_mode), int(bool(white_bkgd)),
                                         _p(out["rgb"]), _p(out["depth"]), _p(out["acc"]), _p(g("weights")),
                                         _p(g("semantic")), _p(g("instance")), _p(g("fix_semantic")),
                                         _p(g("fix_instance")), _stream()), "pnr_composite")
    return out


def _maps(out, R, N, n_sem, n_inst, label_sem, label_inst, want_weights, dev):
    """The per-ray output tensors of a compositing call.  `out`: optional dict of caller-owned tensors (e.g. row slices of
    frame-sized maps: Renderer.render hands every chunk its slice, so a frame is never concatenated); anything it lacks is
    allocated.  Caller-owned tensors must be contiguous fp32 of the exact shape on the right device."""
    f32 = dict(device=dev, dtype=torch.float32)
    shapes = {"rgb": (R, 3), "depth": (R,), "acc": (R,)}
    if want_weights:
        shapes["weights"] = (R, N)
    if n_sem:
        shapes["semantic"] = (R, n_sem)
        if label_sem is not None:
            shapes["fix_semantic"] = (R, n_sem)
    if n_inst:
        shapes["instance"] = (R, n_inst)
        if label_inst is not None:
            shapes["fix_instance"] = (R, n_inst)
    res = {}
    for k, shp in shapes.items():
        t = out.get(k) if out else None
        if t is None:
            t = torch.empty(shp, **f32)
        elif tuple(t.shape) != shp or t.dtype != torch.float32 or t.device != dev or not t.is_contiguous():
            raise ValueError("out[%r] must be a contiguous fp32 tensor of shape %s on %s" % (k, shp, dev))
        res[k] = t
    return res


def fused_supported(desc, n_samples, sem_mode=0, noise=None):
    """Can pnr_mlp_forward_composite take this level?  bf16, no sigma noise, N a multiple of 32; softmax compositing
    (sem_mode 1) where the geometry has a softmax kernel (a head's logit blocks in registers together: plan 2 or 1), on that image:
    net.packed(level, device, fused=fused_image(sem_mode))."""
    ok = (desc.precision == _lib.PREC_BF16 and int(sem_mode) in (0, 1) and noise is None and n_samples % 32 == 0
          and 32 <= n_samples <= 256 and desc.n_sem + desc.n_inst <= 128)
    if ok and int(sem_mode) == 1 and desc.n_sem + desc.n_inst > 0:
        ok = int(_lib.load().pnr_mlp_fused_plan(ctypes.byref(desc_for_mode(desc, 1)))) >= 1     # (flag-aware: a plan with a softmax kernel)
    return ok


def desc_for_mode(desc, sem_mode):
    """desc, or a copy of it, whose PNR_MLP_SOFTMAX flag says sem_mode (what pnr_mlp_forward_composite / _tiles composite)."""
    want = _lib.MLP_SOFTMAX if int(sem_mode) == 1 else 0
    if (desc.flags & _lib.MLP_SOFTMAX) == want:
        return desc
    d2 = _lib.MlpDesc()
    ctypes.memmove(ctypes.byref(d2), ctypes.byref(desc), ctypes.sizeof(d2))
    d2.flags = (desc.flags & ~_lib.MLP_SOFTMAX) | want
    return d2


def fused_image(sem_mode=0):
    """The `fused` argument of PanopticNetwork.packed for a compositing mode: the best plan the geometry has for logits (True), the
    best plan that has a SOFTMAX kernel for softmax ("softmax": plan 2 = k_mlp_tt_sm_* for heads of depth 2, else plan 1)."""
    return "softmax" if int(sem_mode) == 1 else True


@_on_device
def mlp_forward_composite(desc, packed, rays, z, label_sem=None, label_inst=None, white_bkgd=False, want_weights=True, out=None,
                          sem_mode=0, wg_cap=0):
    """Rows a5 + a6 in one pass (inference): the fused MLP reduces every 32-sample tile to one compositing record in its
    epilogue and k_composite_combine finishes the rays -- the raw image (324 B per sample at 45 / 32 heads) is never
    written.  Same dict as composite().  Sums are associated per tile, so results equal mlp_forward + composite to fp32
    rounding (not bit for bit).  sem_mode as in composite(): 1 composites softmax(logits) of each learned field
    (PNR_MLP_SOFTMAX; an image of a plan with a softmax kernel, see fused_supported / fused_image).  wg_cap: plan 2 only, the MLP launch
    on at most that many workgroups (PNR_MLP_WG_CAP) -- a share of the device, for a launch that runs beside another one."""
    rays, z, packed = _chk(rays, "rays"), _chk(z, "z"), _chk(packed, "packed", torch.uint8)
    label_sem = _chk(label_sem, "label_sem", torch.int32)
    label_inst = _chk(label_inst, "label_inst", torch.int32)
    R, N = z.shape
    n_sem, n_inst = desc.n_sem, desc.n_inst
    dev = z.device
    lib = _lib.load()
    desc = desc_for_mode(desc, sem_mode)
    if wg_cap:          # PNR_MLP_WG_CAP: the plan-2 launch on a share of the compute units (Renderer's overlapped levels)
        d2 = _lib.MlpDesc()
        ctypes.memmove(ctypes.byref(d2), ctypes.byref(desc), ctypes.sizeof(d2))
        d2.flags = (desc.flags & 0xFFFF) | ((int(wg_cap) & 0x1FF) << 16)
        desc = d2
    nbytes = lib.pnr_mlp_forward_composite_workspace_bytes(ctypes.byref(desc), R, N, int(bool(want_weights)))
    if nbytes < 0:
        raise RuntimeError("pnr_mlp_forward_composite: unsupported geometry (n_samples=%d must be a multiple of 32)" % N)
    ws = torch.empty(int(nbytes), device=dev, dtype=torch.uint8)
    out = _maps(out, R, N, n_sem, n_inst, label_sem, label_inst, want_weights, dev)
    g = out.get
    _lib.check(lib.pnr_mlp_forward_composite(ctypes.byref(desc), _p(packed), _p(rays), _p(z), R, N, _p(label_sem), _p(label_inst),
                                             int(bool(white_bkgd)), _p(out["rgb"]), _p(out["depth"]), _p(out["acc"]),
                                             _p(g("weights")), _p(g("semantic")), _p(g("instance")), _p(g("fix_semantic")),
                                             _p(g("fix_instance")), _p(ws), _stream()), "pnr_mlp_forward_composite")
    return out


@_on_device
def composite_backward(raw, z, rays, n_sem, n_inst, grads, noise=None, label_sem=None, label_inst=None,
                       ce_sem=None, ce_inst=None, sem_mode=0):
    """Backward of composite() for channel-major raw.  grads: dict with any of rgb, depth, acc, semantic,
    instance, weights, fix_semantic, fix_instance (upstream gradients, contiguous fp32); the fixed-field
    gradients need the per-sample labels.  ce_sem / ce_inst: 1-element device tensors, the scale of the per-sample
    3D cross-entropy gradient (see ce3d).  sem_mode as in composite().  Returns d_raw (ch, R*N).  SURVEY 8a row a9, 8f-1."""
    raw, z, rays = _chk(raw, "raw"), _chk(z, "z"), _chk(rays, "rays")
    noise = _chk(noise, "noise")
    label_sem = _chk(label_sem, "label_sem", torch.int32)
    label_inst = _chk(label_inst, "label_inst", torch.int32)
    R, N = z.shape
    g = {k: _chk(v.contiguous().float(), "g_" + k) for k, v in grads.items() if v is not None}
    ce_sem = None if ce_sem is None else _chk(ce_sem.reshape(1).float().contiguous(), "ce_sem")
    ce_inst = None if ce_inst is None else _chk(ce_inst.reshape(1).float().contiguous(), "ce_inst")
    d_raw = torch.empty_like(raw)
    _lib.check(_lib.load().pnr_composite_backward3(_p(raw), R * N, _p(z), _p(rays), _p(noise), R, N, n_sem, n_inst, int(sem_mode),
                                                   _p(g.get("rgb")), _p(g.get("depth")), _p(g.get("acc")),
                                                   _p(g.get("semantic")), _p(g.get("instance")), _p(g.get("weights")),
                                                   _p(label_sem), _p(label_inst), _p(g.get("fix_semantic")),
                                                   _p(g.get("fix_instance")), _p(ce_sem), _p(ce_inst),
                                                   _p(d_raw), _stream()), "pnr_composite_backward")
    return d_raw


_LOSS_KEYS = ("rgb", "depth", "semantic", "fix_semantic", "instance", "fix_instance")


@_on_device
def losses(weights, maps, targets, n_sem=0, n_inst=0, depth_l2=False, fix_eps=1e-5, want_grads=True, maps_are_prob=False):
    """The trainer's per-ray loss terms of one level and the gradient of their weighted total w.r.t. every map
    (pnr_losses).  weights: dict over rgb/depth/semantic/fix_semantic/instance/fix_instance;
    maps: dict of (R,·) fp32 GPU tensors (any subset); targets: rgb (R,3), depth (R), semantic (R) int32,
    instance (R) int32.  Returns (losses (8,) device tensor: the six means, total, 0; dict of gradients)."""
    lib = _lib.load()
    m = {k: _chk(v.detach().contiguous(), k) for k, v in maps.items() if k in _LOSS_KEYS and v is not None and v.numel()}
    any_map = next(iter(m.values()))
    dev, R = any_map.device, any_map.shape[0]
    t_rgb, t_depth = _chk(targets.get("rgb"), "rgb_gt"), _chk(targets.get("depth"), "depth_gt")
    t_sem = _chk(targets.get("semantic"), "sem_gt", torch.int32)
    t_inst = _chk(targets.get("instance"), "inst_gt", torch.int32)
    cfg = _lib.LossCfg(*(float(weights.get(k, 0.0)) for k in _LOSS_KEYS), int(bool(depth_l2)), float(fix_eps), int(bool(maps_are_prob)))
    use = {"rgb": t_rgb is not None, "depth": t_depth is not None, "semantic": t_sem is not None, "fix_semantic": t_sem is not None,
           "instance": t_inst is not None, "fix_instance": t_inst is not None}
    m = {k: v for k, v in m.items() if use[k]}
    grads = {k: torch.empty_like(v) for k, v in m.items()} if want_grads else {}
    out = torch.empty(8, device=dev, dtype=torch.float32)
    ws = torch.empty(int(lib.pnr_losses_workspace_bytes(R)), device=dev, dtype=torch.uint8)
    g = grads.get
    _lib.check(lib.pnr_losses(ctypes.byref(cfg), R, int(n_sem), int(n_inst), _p(m.get("rgb")), _p(m.get("depth")),
                              _p(m.get("semantic")), _p(m.get("fix_semantic")), _p(m.get("instance")), _p(m.get("fix_instance")),
                              _p(t_rgb), _p(t_depth), _p(t_sem), _p(t_inst), _p(out), _p(g("rgb")), _p(g("depth")),
                              _p(g("semantic")), _p(g("fix_semantic")), _p(g("instance")), _p(g("fix_instance")), _p(ws),
                              _stream()), "pnr_losses")
    return out, grads


@_on_device
def ce3d(raw, first_channel, n_classes, label):
    """Per-sample 3D cross-entropy of the learned logits raw[first_channel:+n_classes] (channel-major (ch,S)) against
    label (S or (R,N)) int32, -1 = unlabelled.  Returns a 2-element device tensor (mean CE, labelled count)."""
    lib = _lib.load()
    label = _chk(label, "label", torch.int32)
    S = label.numel()
    sc = _chk_raw(raw, raw.shape[0], S)
    out = torch.empty(2, device=raw.device, dtype=torch.float32)
    ws = torch.empty(int(lib.pnr_ce3d_workspace_bytes(S)), device=raw.device, dtype=torch.uint8)
    _lib.check(lib.pnr_ce3d(_p(raw), sc, int(first_channel), int(n_classes), _p(label), S, _p(out), _p(ws), _stream()), "pnr_ce3d")
    return out


@_on_device
def sample_pdf(z, weights, n_importance, u=None, want_samples=True, out=None):
    """Coarse z, weights (R,Nc) -> z_fine (R,Nc+Nf) sorted [, z_samples (R,Nf), inds (R,Nf)].
    SURVEY 8a row a7."""
    z, weights, u = _chk(z, "z"), _chk(weights, "weights"), _chk(u, "u")
    R, Nc = z.shape
    dev = z.device
    z_fine = _own(out, (R, Nc + n_importance), torch.float32, dev, "sample_pdf")
    zs = inds = None
    if want_samples:
        zs = torch.empty((R, n_importance), device=dev, dtype=torch.float32)
        inds = torch.empty((R, n_importance), device=dev, dtype=torch.int32)
    _lib.check(_lib.load().pnr_sample_pdf(_p(z), _p(weights), _p(u), R, Nc, n_importance, _p(zs), _p(inds),
                                          _p(z_fine), _stream()), "pnr_sample_pdf")
    return z_fine, zs, inds


@_on_device
def bbox_hits(rays, box, max_hits=8):
    """rays (R,8), box (M,15) -> hit_t (R,mh,2), hit_box (R,mh) int32, hit_count (R) int32.  Row a8."""
    rays, box = _chk(rays, "rays"), _chk(box, "box")
    R, M = rays.shape[0], box.shape[0]
    dev = rays.device
    hit_t = torch.empty((R, max_hits, 2), device=dev, dtype=torch.float32)
    hit_box = torch.empty((R, max_hits), device=dev, dtype=torch.int32)
    hit_count = torch.empty((R,), device=dev, dtype=torch.int32)
    _lib.check(_lib.load().pnr_bbox_hits(_p(rays), R, _p(box), M, max_hits, _p(hit_t), _p(hit_box), _p(hit_count),
                                         _stream()), "pnr_bbox_hits")
    return hit_t, hit_box, hit_count


RAY_SETUP_MAX_HITS = 8      # pnr_ray_setup keeps the hit lists of 256 rays in LDS


@_on_device
def ray_setup(rays, box, box_ids=None, n_samples=64, max_hits=8, lindisp=False, t_rand=None, hull=False, out=None):
    """The coarse level's per-ray preamble in ONE launch (pnr_ray_setup): (hit_t, hit_box, hit_count) as bbox_hits, z (R,N) as
    stratified -- over the hull of the kept intervals with hull=True, as restrict_rays + stratified --, and (label_sem,
    label_inst) as sample_labels when box_ids is given (else None, None).  Bit for bit the separate ops; max_hits <= 8."""
    rays, box, t_rand = _chk(rays, "rays"), _chk(box, "box"), _chk(t_rand, "t_rand")
    box_ids = _chk(box_ids, "box_ids", torch.int32)
    R, M, N = rays.shape[0], box.shape[0], int(n_samples)
    dev = rays.device
    if t_rand is not None:
        assert tuple(t_rand.shape) == (R, N)
    hit_t = torch.empty((R, max_hits, 2), device=dev, dtype=torch.float32)
    hit_box = torch.empty((R, max_hits), device=dev, dtype=torch.int32)
    hit_count = torch.empty((R,), device=dev, dtype=torch.int32)
    z = _own(out, (R, N), torch.float32, dev, "ray_setup")
    ls = li = None
    if box_ids is not None:
        ls = torch.empty((R, N), device=dev, dtype=torch.int32)
        li = torch.empty((R, N), device=dev, dtype=torch.int32)
    _lib.check(_lib.load().pnr_ray_setup(_p(rays), R, _p(box), M, int(max_hits), _p(box_ids), N, int(bool(lindisp)), _p(t_rand),
                                         int(bool(hull)), _p(hit_t), _p(hit_box), _p(hit_count), _p(z), _p(ls), _p(li), _stream()),
               "pnr_ray_setup")
    return (hit_t, hit_box, hit_count), z, ls, li


@_on_device
def sample_pdf_labels(z, weights, n_importance, hits, box_ids, u=None, out=None):
    """sample_pdf + sample_labels of its result in ONE launch (pnr_sample_pdf_labels): (z_fine (R,Nc+Nf), label_sem, label_inst).
    hits = (hit_t, hit_box, hit_count) of bbox_hits / ray_setup.  Bit for bit the separate ops."""
    z, weights, u = _chk(z, "z"), _chk(weights, "weights"), _chk(u, "u")
    hit_t, hit_box, hit_count = _chk(hits[0], "hit_t"), _chk(hits[1], "hit_box", torch.int32), _chk(hits[2], "hit_count", torch.int32)
    box_ids = _chk(box_ids, "box_ids", torch.int32)
    R, Nc = z.shape
    dev = z.device
    Nt = Nc + int(n_importance)
    z_fine = _own(out, (R, Nt), torch.float32, dev, "sample_pdf_labels")
    ls = torch.empty((R, Nt), device=dev, dtype=torch.int32)
    li = torch.empty((R, Nt), device=dev, dtype=torch.int32)
    _lib.check(_lib.load().pnr_sample_pdf_labels(_p(z), _p(weights), _p(u), R, Nc, int(n_importance), _p(z_fine), _p(hit_t), _p(hit_box),
                                                 _p(hit_count), hit_box.shape[1], _p(box_ids), _p(ls), _p(li), _stream()),
               "pnr_sample_pdf_labels")
    return z_fine, ls, li


@_on_device
def restrict_rays(rays, hit_t, hit_count):
    """rays with near / far replaced by the hull of each ray's kept bbox intervals (pnr_restrict_rays); no hit: unchanged."""
    rays = _chk(rays, "rays")
    out = torch.empty_like(rays)
    _lib.check(_lib.load().pnr_restrict_rays(_p(rays), rays.shape[0], _p(_chk(hit_t, "hit_t")), _p(_chk(hit_count, "hit_count", torch.int32)),
                                             hit_t.shape[1], _p(out), _stream()), "pnr_restrict_rays")
    return out


@_on_device
def sample_labels(z, hit_t, hit_box, hit_count, box_ids):
    z, hit_t = _chk(z, "z"), _chk(hit_t, "hit_t")
    hit_box, hit_count = _chk(hit_box, "hit_box", torch.int32), _chk(hit_count, "hit_count", torch.int32)
    box_ids = _chk(box_ids, "box_ids", torch.int32)
    R, N = z.shape
    ls = torch.empty((R, N), device=z.device, dtype=torch.int32)
    li = torch.empty((R, N), device=z.device, dtype=torch.int32)
    _lib.check(_lib.load().pnr_sample_labels(_p(z), R, N, _p(hit_t), _p(hit_box), _p(hit_count), hit_box.shape[1],
                                             _p(box_ids), _p(ls), _p(li), _stream()), "pnr_sample_labels")
    return ls, li


@_on_device
def panoptic_labels(sem, inst=None, is_thing=None):
    """Composited maps -> (semantic label, instance label, panoptic id), each (R) int32 (pnr_panoptic_labels)."""
    sem, inst = _chk(sem, "sem"), _chk(inst, "inst")
    is_thing = _chk(is_thing, "is_thing", torch.int32)
    R, C = sem.shape
    K = inst.shape[1] if inst is not None else 0
    out = [torch.empty(R, device=sem.device, dtype=torch.int32) for _ in range(3)]
    _lib.check(_lib.load().pnr_panoptic_labels(_p(sem), _p(inst), _p(is_thing), R, C, K, _p(out[0]), _p(out[1]), _p(out[2]),
                                               _stream()), "pnr_panoptic_labels")
    return tuple(out)


@_on_device
def confusion(pred, gt, n_classes, conf=None):
    """Accumulate the (n_classes, n_classes) int64 confusion matrix conf[gt, pred] (pnr_confusion).  gt < 0 = ignore."""
    pred, gt = _chk(pred, "pred", torch.int32), _chk(gt, "gt", torch.int32)
    if conf is None:
        conf = torch.zeros((n_classes, n_classes), device=pred.device, dtype=torch.int64)
    _chk(conf, "conf", torch.int64)
    _lib.check(_lib.load().pnr_confusion(_p(pred), _p(gt), pred.numel(), int(n_classes), _p(conf), _stream()), "pnr_confusion")
    return conf
