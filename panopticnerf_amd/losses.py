"""NetworkWrapper -- host-side mirror of the trainer's loss wrapper (the reference's
lib/train/trainers NetworkWrapper is not in the mount, SURVEY.md 0, so names follow SURVEY.md section 2 row 8 and
are documented in DESIGN.md 9).

`NetworkWrapper(net, cfg)(batch)` renders the batch (Renderer.render, autograd on) and evaluates, per level,
    w_rgb * MSE(rgb, batch['rgb'])  +  w_depth * L1|L2(depth, batch['depth'] where > 0)
  + w_sem * CE(semantic logits, batch['pseudo_label'])  +  w_fix_sem * NLL(fixed semantic map, batch['pseudo_label'])
  + the same two terms for the instance field (batch['instance_label'])
  + w_sem3d / w_inst3d * per-sample CE of the learned logits against the bbox labels (when batch has 'bbox')
with ONE fused kernel per level for the per-ray terms and their gradients (pnr_losses); the 3D term's value comes from
pnr_ce3d and its gradient is fused into the compositing backward.  Returns (ret, loss, scalar_stats, image_stats) as
the reference trainer expects from a wrapper.  No torch math on the path; the loss scalars stay on the GPU.
"""
import torch
import torch.nn as nn

from . import ops
from .renderer import make_renderer

_TERMS = ("rgb", "depth", "semantic", "fix_semantic", "instance", "fix_instance")


def _get(cfg, name, default):
    return getattr(cfg, name, default) if cfg is not None else default


class PanopticLossFn(torch.autograd.Function):
    """(total, stats) of one level; differentiable w.r.t. the maps through the gradients pnr_losses wrote."""

    @staticmethod
    def forward(ctx, weights, n_sem, n_inst, depth_l2, fix_eps, prob, targets, keys, *maps):
        m = dict(zip(keys, maps))
        out, grads = ops.losses(weights, m, targets, n_sem, n_inst, depth_l2, fix_eps, True, prob)
        ctx.keys = keys
        ctx.save_for_backward(*[grads.get(k, torch.zeros(0, device=out.device)) for k in keys])
        ctx.mark_non_differentiable(out)
        return out[6], out          # the total as a view of the eight loss scalars (read-only downstream: no copy)

    @staticmethod
    def backward(ctx, g_total, _g_stats):
        gs = ctx.saved_tensors
        # one multi-tensor kernel for the level's maps instead of one small multiply per map (six per level)
        live = [g for g in gs if g.numel()]
        scaled = iter(torch._foreach_mul(live, g_total) if live else ())
        return (None,) * 8 + tuple((next(scaled) if g.numel() else None) for g in gs)


class NetworkWrapper(nn.Module):
    def __init__(self, net, cfg=None):
        super().__init__()
        self.net = net
        self.renderer = make_renderer(cfg, net)
        self.weights = {"rgb": _get(cfg, "w_rgb", 1.0), "depth": _get(cfg, "w_depth", 0.1),
                        "semantic": _get(cfg, "w_sem", 1.0), "fix_semantic": _get(cfg, "w_fix_sem", 1.0),
                        "instance": _get(cfg, "w_inst", 1.0), "fix_instance": _get(cfg, "w_fix_inst", 1.0)}
        self.w_sem3d, self.w_inst3d = _get(cfg, "w_sem3d", 0.1), _get(cfg, "w_inst3d", 0.1)
        self.depth_l2 = bool(_get(cfg, "depth_l2", False))
        self.fix_eps = float(_get(cfg, "fix_eps", 1e-5))
        self._tw = {}              # the loss terms' weight vector on the device (rebuilt when the set of terms changes)

    def forward(self, batch):
        ret = self.renderer.render(batch)
        n0 = self.net.nerf(0)
        C, K = n0.n_sem, n0.n_inst
        dev = batch["rays"].device
        flat = lambda t, dt: None if t is None else t.reshape(-1, *t.shape[2:]).to(dev, dt).contiguous()
        targets = {"rgb": flat(batch.get("rgb"), torch.float32), "depth": flat(batch.get("depth"), torch.float32),
                   "semantic": flat(batch.get("pseudo_label"), torch.int32) if C else None,
                   "instance": flat(batch.get("instance_label"), torch.int32) if K else None}
        terms, tw = [], []          # the scalars the loss is a weighted sum of: per level the fused total (weight 1), the 3D terms
        stats = {}
        for lv in (0, 1):
            if f"rgb_{lv}" not in ret:
                continue
            keys = tuple(k for k in _TERMS if f"{k}_{lv}" in ret)
            maps = [ret[f"{k}_{lv}"].reshape(-1, *ret[f"{k}_{lv}"].shape[2:]) for k in keys]
            total, st = PanopticLossFn.apply(self.weights, C, K, self.depth_l2, self.fix_eps, self.renderer.sem_mode == 1,
                                             targets, keys, *maps)
            terms.append(total)
            tw.append(1.0)
            for i, k in enumerate(_TERMS):
                stats[f"{k}_loss_{lv}"] = st[i]
            for k, w in (("ce3d_semantic", self.w_sem3d), ("ce3d_instance", self.w_inst3d)):
                if f"{k}_{lv}" in ret and w:
                    terms.append(ret[f"{k}_{lv}"].reshape(()))
                    tw.append(float(w))
                    stats[f"{k}_loss_{lv}"] = ret[f"{k}_{lv}"].detach()
        # one gather, one multiply, one sum instead of a multiply and an add per term (ten small kernels forward, four backward, per step)
        loss = (torch.stack(terms) * self._term_weights(tuple(tw), dev)).sum()      # (not torch.dot: that is a rocBLAS call)
        stats["loss"] = loss.detach()
        return ret, loss, stats, {}

    def _term_weights(self, tw, dev):
        key = (tw, str(dev))
        if self._tw.get("key") != key:
            self._tw = {"key": key, "vec": torch.tensor(tw, dtype=torch.float32, device=dev)}
        return self._tw["vec"]
