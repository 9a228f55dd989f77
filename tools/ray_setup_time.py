"""The per-ray preamble of a chunk, fused against separate (hipEvents, one process):
   python tools/ray_setup_time.py [rays]      (PNR_LIB_PATH selects an A/B build of libpnr.so)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panopticnerf_amd import ops, synthetic

R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
rays = synthetic.camera_rays()[:R].contiguous().to(dev)
box, ids = (t.to(dev) for t in synthetic.random_boxes(64, 45, 32))
w = torch.rand((R, 64), device=dev) ** 4


def ms(fn, n=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


hits = ops.bbox_hits(rays, box, 8)
z = ops.stratified(rays, 64)
print("lib", os.environ.get("PNR_LIB_PATH", "default"))
print("bbox_hits            %7.1f us" % ms(lambda: ops.bbox_hits(rays, box, 8)))
print("stratified           %7.1f us" % ms(lambda: ops.stratified(rays, 64)))
print("sample_labels (64)   %7.1f us" % ms(lambda: ops.sample_labels(z, *hits, ids)))
print("ray_setup            %7.1f us" % ms(lambda: ops.ray_setup(rays, box, ids, 64, 8)))
print("ray_setup, no labels %7.1f us" % ms(lambda: ops.ray_setup(rays, box, None, 64, 8)))
zf = ops.sample_pdf(z, w, 128, want_samples=False)[0]
print("sample_pdf           %7.1f us" % ms(lambda: ops.sample_pdf(z, w, 128, want_samples=False)))
print("sample_labels (192)  %7.1f us" % ms(lambda: ops.sample_labels(zf, *hits, ids)))
print("sample_pdf_labels    %7.1f us" % ms(lambda: ops.sample_pdf_labels(z, w, 128, hits, ids)))
