"""GPU tests of SURVEY.md 8f rank 4: label post-processing and the evaluator counters, bit-exact with oracle/np_oracle.py."""
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from oracle import np_oracle as no
from panopticnerf_amd import ops
from panopticnerf_amd.evaluate import Evaluator

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,C,K", [(1, 2, 0), (1000, 45, 32), (70001, 19, 7)])
def test_panoptic_labels_bit_exact(dev, R, C, K):
    g = torch.Generator().manual_seed(R)
    sem = torch.randn(R, C, generator=g)
    sem[::7] = sem[::7].round()                       # ties: lowest index wins
    inst = torch.randn(R, K, generator=g).round() if K else None
    # NaN logits never win and never hide a later column of the same lane (columns c and c + 16 share a lane): ADVICE r3
    sem[1::11, 0] = float("nan")
    if C > 16:
        sem[3::13, 2] = float("nan")
        sem[3::13, 18] = 50.0                         # the row maximum sits 16 columns behind a NaN
    sem[5::17] = float("nan")                         # rows without any number: index 0
    if K:
        inst[2::9, 0] = float("nan")
    thing = (torch.arange(C) % 3 == 0).int()
    sl, il, pan = ops.panoptic_labels(sem.to(dev), None if inst is None else inst.to(dev), thing.to(dev))
    rs, ri, rp = no.panoptic_labels(sem.numpy(), None if inst is None else inst.numpy(), thing.numpy())
    assert np.array_equal(sl.cpu().numpy(), rs) and np.array_equal(il.cpu().numpy(), ri) and np.array_equal(pan.cpu().numpy(), rp)
    sl2, il2, _ = ops.panoptic_labels(sem.to(dev), None if inst is None else inst.to(dev), None)      # every class a thing
    assert torch.equal(sl2, sl) and (K == 0 or (il2 >= 0).all())


def test_confusion_exact_and_accumulates(dev):
    g = torch.Generator().manual_seed(1)
    C, n = 45, 529408
    pred = torch.randint(0, C, (n,), generator=g, dtype=torch.int32)
    gt = torch.randint(-1, C + 1, (n,), generator=g, dtype=torch.int32)      # -1 and C are ignored
    conf = ops.confusion(pred.to(dev), gt.to(dev), C)
    ref = no.confusion(pred.numpy(), gt.numpy(), C)
    assert np.array_equal(conf.cpu().numpy(), ref)
    conf = ops.confusion(pred.to(dev), gt.to(dev), C, conf)                  # second frame accumulates
    assert np.array_equal(conf.cpu().numpy(), 2 * ref)
    assert ops.confusion(pred[:0].to(dev), gt[:0].to(dev), C).sum().item() == 0


def test_evaluator_psnr_miou(dev):
    g = torch.Generator().manual_seed(3)
    R, C, K = 5000, 6, 3
    ev = Evaluator(n_classes=C, is_thing=[0, 1, 0, 1, 0, 0])
    ref_conf = np.zeros((C, C), np.int64)
    psnr = []
    for f in range(3):
        rgb, gt_rgb = torch.rand(1, R, 3, generator=g), torch.rand(1, R, 3, generator=g)
        sem, inst = torch.randn(1, R, C, generator=g), torch.randn(1, R, K, generator=g)
        lab = torch.randint(-1, C, (1, R), generator=g)
        out = {"rgb_1": rgb.to(dev), "semantic_1": sem.to(dev), "instance_1": inst.to(dev)}
        res = ev.evaluate(out, {"rgb": gt_rgb.to(dev), "pseudo_label": lab.to(dev)})
        sl, il, pan = no.panoptic_labels(sem[0].numpy(), inst[0].numpy(), np.array([0, 1, 0, 1, 0, 0]))
        assert np.array_equal(res["panoptic_id"].cpu().numpy(), pan)
        ref_conf += no.confusion(sl, lab[0].numpy(), C)
        psnr.append(-10 * np.log10(((rgb - gt_rgb).double() ** 2).mean().item()))
    s = ev.summarize()
    tp = np.diag(ref_conf).astype(float)
    union = ref_conf.sum(0) + ref_conf.sum(1) - tp
    assert abs(s["psnr"] - np.mean(psnr)) < 1e-4
    assert abs(s["miou"] - np.mean(tp / union)) < 1e-12 and abs(s["pixel_acc"] - tp.sum() / ref_conf.sum()) < 1e-12
    assert ev.summarize() == {}                       # counters reset


def test_panoptic_quality_matches_oracle(dev):
    """PQ terms of the evaluator (pnr_confusion over compact segment ids + tensor ops) vs plain segment loops."""
    g = torch.Generator().manual_seed(9)
    C, K, H, W = 5, 4, 40, 60
    thing = [0, 1, 1, 0, 0]        # class 0 is stuff: id = class*1000 + instance cannot tell a class-0 thing from a stuff class
    ev = Evaluator(NS(num_classes=C, num_instances=K), is_thing=thing)
    ref = np.zeros((C, 4))
    for f in range(3):
        # blocky ground truth and a prediction that agrees on most of it
        gt_sem = torch.randint(0, C, (H // 10, W // 10), generator=g).repeat_interleave(10, 0).repeat_interleave(10, 1)
        gt_ins = torch.randint(0, K, (H // 10, W // 10), generator=g).repeat_interleave(10, 0).repeat_interleave(10, 1)
        th = torch.tensor(thing)[gt_sem] != 0
        if f == 2:
            gt_ins = gt_ins * 37 + 500        # arbitrary, large ground-truth instance indices (KITTI-360 style): ids are re-mapped per frame
        gt_pan = torch.where(th, gt_sem * 1000 + gt_ins, gt_sem)
        gt_pan[:3] = -1                                                  # an ignored band
        noise = torch.rand(H, W, generator=g) < 0.25
        sem_logits = torch.nn.functional.one_hot(torch.where(noise, torch.randint(0, C, (H, W), generator=g), gt_sem), C).float() * 5
        ins_logits = torch.nn.functional.one_hot(torch.where(noise, torch.randint(0, K, (H, W), generator=g), gt_ins % 37 % K if f == 2 else gt_ins), K).float() * 5
        out = {"rgb_1": torch.zeros(1, H * W, 3, device=dev), "semantic_1": sem_logits.reshape(1, -1, C).to(dev),
               "instance_1": ins_logits.reshape(1, -1, K).to(dev)}
        res = ev.evaluate(out, {"panoptic_gt": gt_pan.reshape(1, -1).to(dev)})
        ref += no.panoptic_quality_terms(res["panoptic_id"].cpu().numpy(), gt_pan.reshape(-1).numpy(), C)
    got = ev.pq.cpu().numpy()
    assert np.allclose(got, ref, atol=1e-9), (got, ref)
    s = ev.summarize()
    den = ref[:, 1] + 0.5 * ref[:, 2] + 0.5 * ref[:, 3]
    assert abs(s["pq"] - np.mean(ref[den > 0, 0] / den[den > 0])) < 1e-12



def test_out_of_range_panoptic_ids_go_to_the_overflow_row_and_reset(dev):
    """A ground-truth segment whose class index is >= n_classes is counted in NO real class (it lands in an overflow row that
    is dropped), summarize() reports it -- and resets every accumulator before raising, so the next frame set starts clean
    (the ids used to be clamped into class C-1 and the polluted table survived the exception)."""
    C, K, H, W = 4, 3, 20, 30
    ev = Evaluator(NS(num_classes=C, num_instances=K), is_thing=[0, 1, 1, 0])
    sem = torch.nn.functional.one_hot(torch.randint(0, C, (H * W,), generator=torch.Generator().manual_seed(3)), C).float()[None] * 5
    ins = torch.zeros(1, H * W, K)
    out = {"rgb_1": torch.zeros(1, H * W, 3, device=dev), "semantic_1": sem.to(dev), "instance_1": ins.to(dev)}
    res = ev.evaluate(out, {})
    good = res["panoptic_id"].clone().reshape(1, -1)
    ev.summarize()
    # the same frame twice: once with a clean ground truth, once with a band of class-index 7 (>= C) segments
    ev.evaluate(out, {"panoptic_gt": good})
    clean = ev.pq.clone()
    ev.summarize()
    bad = good.clone()
    bad[:, : 5 * W] = 7 * 1000 + 1
    ev.evaluate(out, {"panoptic_gt": bad})
    # exactly the terms of a table with room for class 7, cut back to the real classes: the segment took no real row
    want = no.panoptic_quality_terms(res["panoptic_id"].cpu().numpy(), bad.reshape(-1).cpu().numpy(), 8)[:C]
    assert np.allclose(ev.pq.cpu().numpy(), want, atol=1e-9), (ev.pq, want)
    with pytest.raises(ValueError, match="class index"):
        ev.summarize()
    assert ev.pq is None and ev._bad_ids is None and ev.summarize() == {}      # reset although it raised
    ev.evaluate(out, {"panoptic_gt": good})
    assert torch.equal(ev.pq, clean)                                           # and the next accumulation is unpolluted
