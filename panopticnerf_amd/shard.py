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
    """Inverse of shard_rays for a dict of per-ray tensors (first dim = local ray count)."""
    if world == 1:
        return local
    out = {}
    n_max = (n_rays + world - 1) // world
    for k, v in local.items():
        pad = torch.zeros((n_max,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        pad[: v.shape[0]] = v
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
        full = torch.empty((n_rays,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        for r in range(world):
            idx = shard_indices(n_rays, r, world)
            full[idx] = parts[r][: idx.numel()]
        out[k] = full
    return out


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
