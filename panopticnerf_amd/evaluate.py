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
        # panoptic id = class * 1000 + instance on things, class on stuff (KITTI-360's convention, where id 0 is 'unlabeled'):
        # a THING of class 0 would be encoded as its bare instance index and read back as the stuff class of that number
        if is_thing is not None and len(is_thing) and int(is_thing[0]):
            raise ValueError("Evaluator: class 0 cannot be a 'thing' under the class * 1000 + instance id convention (its "
                             "segments would alias stuff classes); re-index the label set so that class 0 is stuff / unlabeled")
        self.level = 1
        self.conf = None
        self.mse = []
        self.n_inst = g("num_instances", 0)
        self.pq = None          # (C, 4) device tensor: sum of matched IoUs, TP, FP, FN per class
        self._bad_ids = None    # device counter: segments whose class index was >= n_classes (reported by summarize())

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
        """Per-frame PQ terms (Kirillov et al.): segments match when same class and IoU > 0.5 (then the match is unique).
        Panoptic ids: class*1000 + instance for things, class for stuff; < 0 = ignore.  Segment ids are re-mapped PER
        FRAME to 0..n-1 (torch.unique), so arbitrary instance indices (KITTI-360 ground truth uses large ones) can never
        alias into another class's slots.  The pixel-pair counting is pnr_confusion; the small (n_seg x n_seg) IoU /
        matching algebra below is host-side torch bookkeeping on the device (evaluation only, not on the render path)."""
        C = self.n_classes
        dev = pred_id.device

        def segments(ids):
            ok = ids >= 0
            uniq, inv = torch.unique(ids[ok], return_inverse=True)
            comp = torch.full_like(ids, -1)
            comp[ok] = inv.int()
            cls = torch.where(uniq >= 1000, uniq // 1000, uniq).long()
            # out-of-range classes go to a real OVERFLOW row (index C of a C + 1 row table, dropped before it is accumulated), so
            # they never count in a real class's terms; their number is kept on the device and summarize() reports it -- no host
            # round trip per frame
            bad = (cls >= C) | (cls < 0)
            self._bad_ids = bad.sum() if self._bad_ids is None else self._bad_ids + bad.sum()
            return comp.contiguous(), torch.where(bad, torch.full_like(cls, C), cls)

        seg_p, cls_p = segments(pred_id)
        seg_g, cls_g = segments(gt_id)
        n = max(int(cls_p.numel()), int(cls_g.numel()), 1)
        pair = ops.confusion(seg_p, seg_g, n).double()[: max(cls_g.numel(), 1), : max(cls_p.numel(), 1)]   # [gt seg, pred seg]
        if cls_g.numel() == 0 or cls_p.numel() == 0:
            pair = torch.zeros((cls_g.numel(), cls_p.numel()), device=dev, dtype=torch.float64)
        # pixels whose ground truth is ignored do not count against the prediction
        valid = (gt_id >= 0) & (seg_p >= 0)
        area_p = torch.bincount(seg_p[valid].long(), minlength=cls_p.numel()).double()
        area_g = pair.sum(1)
        union = area_g[:, None] + area_p[None, :] - pair
        iou = torch.where(union > 0, pair / union.clamp(min=1), torch.zeros_like(pair))
        match = (iou > 0.5) & (cls_g[:, None] == cls_p[None, :])
        tp_g, tp_p = match.any(1), match.any(0)
        terms = torch.zeros((C + 1, 4), device=dev, dtype=torch.float64)          # row C: the overflow row (dropped)
        terms[:, 0].index_add_(0, cls_g, (iou * match).sum(1))
        terms[:, 1].index_add_(0, cls_g, tp_g.double())
        terms[:, 2].index_add_(0, cls_p, ((area_p > 0) & ~tp_p).double())
        terms[:, 3].index_add_(0, cls_g, ((area_g > 0) & ~tp_g).double())
        terms = terms[:C]
        self.pq = terms if self.pq is None else self.pq + terms

    def summarize(self):
        """The metrics accumulated since the last call; the accumulators are reset in every case (also when it raises)."""
        try:
            return self._summarize()
        finally:
            self.mse, self.conf, self.pq, self._bad_ids = [], None, None, None

    def _summarize(self):
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
            if self._bad_ids is not None and int(self._bad_ids) > 0:
                raise ValueError("Evaluator: %d segment id(s) carried a class index >= n_classes = %d" % (int(self._bad_ids), self.n_classes))
            t = self.pq.cpu()
            denom = t[:, 1] + 0.5 * t[:, 2] + 0.5 * t[:, 3]
            seen = denom > 0
            pq = torch.where(seen, t[:, 0] / denom.clamp(min=1e-12), torch.full_like(denom, float("nan")))
            out["pq_per_class"] = pq.tolist()
            out["pq"] = float(pq[seen].mean()) if seen.any() else math.nan
            out["sq"] = float((t[:, 0][seen] / t[:, 1][seen].clamp(min=1e-12))[t[:, 1][seen] > 0].mean()) if (t[:, 1] > 0).any() else math.nan
            out["rq"] = float((t[:, 1] / denom.clamp(min=1e-12))[seen].mean()) if seen.any() else math.nan
        return out
