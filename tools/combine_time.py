"""k_composite_combine alone (the second half of pnr_mlp_forward_composite) at a renderer chunk: 65536 rays x 192 and x 64 samples,
45 / 32 heads, scene labels; hipEvent time of the whole fused call minus nothing -- run under `rocprofv3 --kernel-trace --stats`
to read the kernel's own duration.  Prints an output fingerprint (builds that only move loads must agree bit for bit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, ops, synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
synthetic.trained_like_(net)
net = net.to(dev)
rays = synthetic.camera_rays()[:65536].to(dev)
box, ids = (t.to(dev) for t in synthetic.random_boxes(64, 45, 32))
h = ops.bbox_hits(rays, box, 8)
for N, lv in ((192, 1), (64, 0)):
    z = ops.stratified(rays, N)
    desc, img = net.packed(lv, dev, fused=True)
    ls, li = ops.sample_labels(z, h[0], h[1], h[2], ids)
    for _ in range(6):
        out = ops.mlp_forward_composite(desc, img, rays, z, ls, li, False, True)
    torch.cuda.synchronize()
    fp = "/".join("%016x" % (int(out[k].double().sum().cpu().view(torch.int64)) & 0xffffffffffffffff)
                  for k in ("rgb", "depth", "semantic", "instance", "fix_semantic", "fix_instance", "weights", "acc"))
    print(os.path.basename(os.environ.get("PNR_LIB_PATH", "libpnr.so")), N, fp)
