"""Ray sharding across ranks.  Rays are independent, so the render path has
NO data-path collective: each rank renders an interleaved subset (ray_id % world == rank, which
balances bbox-hit density across ranks better than image tiles) and, when a whole frame is
wanted on every rank, the per-ray maps are all-gathered afterwards.

`render_fn` is whatever renders a (n,8) ray tensor into a dict of per-ray tensors --
Renderer.render on the GPU; the CPU gloo tests pass the oracle in its place.
"""
import torch
import torch.distributed as dist


def shard_indices(n_rays, rank, world):
    return torch.arange(rank, n_rays, world)


def shard_rays(rays, rank, world):
    """rays (R,8) -> this rank's interleaved share."""
    return rays[rank::world].contiguous()


def gather_maps(local, n_rays, rank, world, group=None):
    """Inverse of shard_rays for a dict of per-ray tensors (first dim = local ray count): ONE collective per frame.  Every
    map is flattened to 4-byte columns (int32 maps bit-cast, other widths through an exact carrier: _to_carrier) and packed side by side into one (n_max, F) buffer; one
    all_gather_into_tensor brings the (world, n_max, F) block to every rank, and because the sharding is interleaved
    (ray = i * world + r) a permute to (n_max, world, F) IS the frame order -- no per-rank scatter.  On xGMI a collective is
    latency-bound at these sizes (a few MB), so one flat bucket beats one all_gather per map."""
    if world == 1:
        return local
    n_max = (n_rays + world - 1) // world
    keys = list(local)
    cols, parts = [], []
    for k in keys:
        v = local[k]
        width = 1
        for d in v.shape[1:]:
            width *= int(d)
        # explicit width: reshape(n, -1) is ambiguous for an EMPTY shard (a frame with fewer rays than ranks)
        flat = _to_carrier(v.reshape(v.shape[0], width).contiguous(), k)
        cols.append(flat.shape[1])
        parts.append(flat)
    first = local[keys[0]]
    buf = torch.zeros((n_max, sum(cols)), dtype=torch.float32, device=first.device)
    buf[: first.shape[0]] = torch.cat(parts, 1) if len(parts) > 1 else parts[0]
    out_all = torch.empty((world * n_max, buf.shape[1]), dtype=torch.float32, device=first.device)    # rank-major concatenation
    try:
        dist.all_gather_into_tensor(out_all, buf, group=group)
    except (RuntimeError, NotImplementedError, AttributeError):
        # a backend without the flat form (older gloo builds): the list form into views of the same block -- still one collective
        dist.all_gather(list(out_all.view(world, n_max, buf.shape[1]).unbind(0)), buf, group=group)
    full = out_all.view(world, n_max, buf.shape[1]).permute(1, 0, 2).reshape(world * n_max, buf.shape[1])[:n_rays]   # (i, r) -> ray i * world + r
    out, c0 = {}, 0
    for k, c in zip(keys, cols):
        v = local[k]
        # clone(), not contiguous(): a one-row slice counts as contiguous where it lies, and its odd storage offset cannot be
        # viewed as an 8-byte type
        out[k] = _from_carrier(full[:, c0:c0 + c].clone(memory_format=torch.contiguous_format), v.dtype).reshape((n_rays,) + tuple(v.shape[1:]))
        c0 += c
    return out


def _to_carrier(flat, name):
    """(n, c) map of any dtype -> (n, c') float32 columns that carry it EXACTLY: 4-byte types are bit-cast, 8-byte types are
    bit-cast into two columns per element, narrower types are widened (bool / 8- / 16-bit integers -> int32, bf16 / fp16 ->
    float32: both exact) -- so render_sharded(keys=None) gathers whatever maps a renderer returns."""
    es = flat.element_size()
    if es == 4 or es == 8:
        return flat.view(torch.float32)
    if flat.dtype in (torch.bfloat16, torch.float16):
        return flat.float()
    if flat.dtype in (torch.bool, torch.uint8, torch.int8, torch.int16):
        return flat.to(torch.int32).view(torch.float32)
    raise TypeError("gather_maps: %s has dtype %s, which has no exact 4-byte carrier" % (name, flat.dtype))


def _from_carrier(cols, dtype):
    es = torch.empty((), dtype=dtype).element_size()
    if es == 4 or es == 8:
        return cols.view(dtype)
    if dtype in (torch.bfloat16, torch.float16):
        return cols.to(dtype)
    return cols.view(torch.int32).to(dtype)


def render_sharded(render_fn, rays, rank, world, gather=True, group=None, keys=None, reduce_fn=None):
    """Render this rank's share of `rays`; optionally reassemble the full maps on every rank.

    keys: gather only these maps.  reduce_fn(local) -> dict: applied before the gather, e.g. `label_maps` below,
    so that the (rays/world, C+K) logit maps shrink to three int32 label maps first (~2 MB instead
    of ~100 MB per frame at C+K = 77)."""
    local = render_fn(shard_rays(rays, rank, world))
    if reduce_fn is not None:
        local = reduce_fn(local)
    if keys is not None:
        local = {k: local[k] for k in keys}
    return gather_maps(local, rays.shape[0], rank, world, group) if gather else local


def label_maps(level=1, is_thing=None, keep=("rgb", "depth")):
    """reduce_fn for render_sharded: replaces the semantic / instance logit maps of `level` by the semantic, instance
    and panoptic label maps (pnr_panoptic_labels on the GPU) and keeps only `keep` of the other maps."""
    from . import ops

    def fn(local):
        out = {f"{k}_{level}": local[f"{k}_{level}"] for k in keep if f"{k}_{level}" in local}
        sem = local[f"semantic_{level}"]
        inst = local.get(f"instance_{level}")
        th = None if is_thing is None else torch.as_tensor(is_thing, dtype=torch.int32, device=sem.device)
        sl, il, pan = ops.panoptic_labels(sem.reshape(-1, sem.shape[-1]).contiguous(),
                                          None if inst is None else inst.reshape(-1, inst.shape[-1]).contiguous(), th)
        out.update({"semantic_label": sl, "instance_label": il, "panoptic_id": pan})
        return out

    return fn
