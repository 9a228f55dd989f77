"""One BASELINE config-5 frame rendered in 8 chunks (the default), in 2 and in ONE chunk (529,408 rays x 192 samples = 101.6 M samples in a
single fused launch: the largest launch the frame can ask for -- 32-bit sample indices, the per-ray table's 67 MB, 49,632 groups per
workgroup-stride): every output map must be the same bits.  usage: python tools/chunk_equality.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panopticnerf_amd import make_network, make_renderer, synthetic

dev = torch.device("cuda:0")
ref = None
for chunk in (65536, 300000, 600000):
    cfg = synthetic.baseline_cfg(5, precision="bf16", chunk_size=chunk)
    torch.manual_seed(0)
    net = make_network(cfg).eval()
    synthetic.trained_like_(net)
    net = net.to(dev)
    rays = synthetic.camera_rays().to(dev)
    box, ids = (t.to(dev) for t in synthetic.random_boxes(64, cfg.num_classes, cfg.num_instances))
    with torch.no_grad():
        out = make_renderer(cfg, net).render({"rays": rays[None], "bbox": box, "bbox_ids": ids})
    torch.cuda.synchronize()
    out = {k: v.cpu() for k, v in out.items()}
    if ref is None:
        ref = out
        print("chunk_size %7d: reference (%d maps, %d rays)" % (chunk, len(out), rays.shape[0]))
        continue
    bad = [k for k in ref if not torch.equal(ref[k], out[k])]
    print("chunk_size %7d: %s" % (chunk, "every map bit-identical to the 8-chunk frame" if not bad else "DIFFERENT: %s" % bad), flush=True)
    assert not bad
