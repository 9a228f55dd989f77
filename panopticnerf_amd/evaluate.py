"""Evaluator -- host-side mirror of the reference's evaluator (lib/evaluators/*; not in the mount,
SURVEY.md 0): `evaluate(output, batch)` per frame, `summarize()` at the end.

Per frame, on the GPU: semantic / instance / panoptic label maps from the fine-level composited maps
(pnr_panoptic_labels), the semantic confusion matrix against batch['pseudo_label' | 'semantic_gt'] (pnr_confusion,
accumulated over frames in a device int64 matrix) and the colour MSE (pnr_losses' rgb term).  `summarize()` turns those
counters into PSNR (mean over frames), mIoU / per-class IoU and pixel accuracy.  Nothing is copied to the host before
summarize()."""
import math

import torch

from . import ops


class Evaluator:
    def __init__(self, cfg=None, n_classes=None, is_thing=None):
        g = lambda k, d: getattr(cfg, k, d) if cfg is not None else d
        self.n_classes = n_classes if n_classes is not None else g("num_classes", 0)
        self.is_thing = is_thing
        self.level = 1
        self.conf = None
        self.mse = []
        self.n_inst = g("num_instances", 0)
        self.pq = None          # (C, 4) device tensor: sum of matched IoUs, TP, FP, FN per class

    def evaluate(self, output, batch):
        lv = self.level if f"rgb_{self.level}" in output else 0
        rgb = output[f"rgb_{lv}"].reshape(-1, 3).float().contiguous()
        dev = rgb.device
        res = {}
        if batch.get("rgb") is not None:
            gt = batch["rgb"].reshape(-1, 3).to(dev, torch.float32).contiguous()
            out, _ = ops.losses({"rgb": 1.0}, {"rgb": rgb}, {"rgb": gt}, want_grads=False)
            self.mse.append(out[0])
        if self.n_classes and f"semantic_{lv}" in output:
            sem = output[f"semantic_{lv}"].reshape(-1, self.n_classes).float().contiguous()
            inst = output.get(f"instance_{lv}")
            inst = None if inst is None else inst.reshape(-1, inst.shape[-1]).float().contiguous()
            th = None if self.is_thing is None else torch.as_tensor(self.is_thing, dtype=torch.int32, device=dev)
            res["semantic_label"], res["instance_label"], res["panoptic_id"] = ops.panoptic_labels(sem, inst, th)
            gt = batch.get("semantic_gt", batch.get("pseudo_label"))
            if gt is not None:
                self.conf = ops.confusion(res["semantic_label"], gt.reshape(-1).to(dev, torch.int32).contiguous(),
                                          self.n_classes, self.conf)
            pg = batch.get("panoptic_gt")
            if pg is not None:
                self._accumulate_pq(res["panoptic_id"], pg.reshape(-1).to(dev, torch.int32).contiguous())
        return res

    def _accumulate_pq(self, pred_id, gt_id):
        """Per-frame PQ terms (Kirillov et al.): segments match when same class and IoU > 0.5 (then the match is unique)."""
        C, K1 = self.n_classes, max(self.n_inst, 1) + 1
        compact = lambda x: torch.where(x < 0, torch.full_like(x, -1),
                                        torch.where(x >= 1000, (x // 1000) * K1 + (x % 1000) + 1, x * K1)).int().contiguous()
        n_seg = C * K1
        pair = ops.confusion(compact(pred_id), compact(gt_id), n_seg).double()          # [gt segment, pred segment]
        # pixels whose ground truth is ignored do not count against the prediction
        valid = gt_id >= 0
        area_p = torch.bincount(compact(pred_id)[valid].long().clamp(min=0), minlength=n_seg).double()
        area_g = pair.sum(1)
        union = area_g[:, None] + area_p[None, :] - pair
        iou = torch.where(union > 0, pair / union.clamp(min=1), torch.zeros_like(pair))
        cls = torch.arange(n_seg, device=pair.device) // K1
        match = (iou > 0.5) & (cls[:, None] == cls[None, :])
        tp_g, tp_p = match.any(1), match.any(0)
        terms = torch.zeros((C, 4), device=pair.device, dtype=torch.float64)
        terms[:, 0].index_add_(0, cls, (iou * match).sum(1))
        terms[:, 1].index_add_(0, cls, tp_g.double())
        terms[:, 2].index_add_(0, cls, ((area_p > 0) & ~tp_p).double())
        terms[:, 3].index_add_(0, cls, ((area_g > 0) & ~tp_g).double())
        self.pq = terms if self.pq is None else self.pq + terms

    def summarize(self):
        out = {}
        if self.mse:
            mse = torch.stack(self.mse).double().cpu()
            out["psnr"] = float((-10.0 * torch.log10(mse.clamp(min=1e-12))).mean())
            out["mse"] = float(mse.mean())
        if self.conf is not None:
            c = self.conf.double().cpu()
            tp = c.diag()
            union = c.sum(0) + c.sum(1) - tp
            seen = union > 0
            iou = torch.where(seen, tp / union.clamp(min=1), torch.full_like(tp, float("nan")))
            out["iou"] = iou.tolist()
            out["miou"] = float(iou[seen].mean()) if seen.any() else math.nan
            out["pixel_acc"] = float(tp.sum() / c.sum().clamp(min=1))
        if self.pq is not None:
            t = self.pq.cpu()
            denom = t[:, 1] + 0.5 * t[:, 2] + 0.5 * t[:, 3]
            seen = denom > 0
            pq = torch.where(seen, t[:, 0] / denom.clamp(min=1e-12), torch.full_like(denom, float("nan")))
            out["pq_per_class"] = pq.tolist()
            out["pq"] = float(pq[seen].mean()) if seen.any() else math.nan
            out["sq"] = float((t[:, 0][seen] / t[:, 1][seen].clamp(min=1e-12))[t[:, 1][seen] > 0].mean()) if (t[:, 1] > 0).any() else math.nan
            out["rq"] = float((t[:, 1] / denom.clamp(min=1e-12))[seen].mean()) if seen.any() else math.nan
        self.mse, self.conf, self.pq = [], None, None
        return out
