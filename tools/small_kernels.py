"""The small kernels of an inference frame (everything that is not the MLP launch): one 65,536-ray chunk of configs[4] through
Renderer.render + the evaluator's label post-processing, 4 times -- run under `rocprofv3 --kernel-trace` and summarise with
tools/small_kernels.py --summary <db>.  Prints a fingerprint of the outputs (A/B builds must agree bit for bit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--summary":
    import sqlite3
    cur = sqlite3.connect(sys.argv[2]).cursor()
    tag = sys.argv[3] if len(sys.argv) > 3 else ""
    for r in cur.execute("select name, count(*), avg(end-start), min(end-start) from kernels where name not like '%k_mlp_pp%' "
                         "group by name order by 3*count(*) desc limit 9"):
        print(f"{tag:10s} {r[0][:56]:56s} n={r[1]:3d} avg {r[2] / 1e3:7.1f} us  min {r[3] / 1e3:7.1f}")
    sys.exit(0)
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, make_renderer, ops, synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32, keep_weights=False)
net = make_network(cfg).eval()
synthetic.trained_like_(net)
net = net.to(dev)
rend = make_renderer(cfg, net)
rays = synthetic.camera_rays()[:65536].to(dev)
box, ids = (t.to(dev) for t in synthetic.random_boxes(64, 45, 32))
thing = (torch.arange(45, device=dev) % 3 == 0).int()
with torch.no_grad():
    for _ in range(4):
        out = rend.render({"rays": rays[None], "bbox": box, "bbox_ids": ids})
        lab = ops.panoptic_labels(out["semantic_1"][0], out["instance_1"][0], thing)
torch.cuda.synchronize()
fp = "/".join("%016x" % (int(t.double().sum().cpu().view(torch.int64)) & 0xffffffffffffffff)
              for t in (out["rgb_1"], out["depth_1"], out["semantic_1"], out["fix_semantic_1"], out["z_vals_1"], lab[0].float(), lab[1].float(), lab[2].float()))
print("fingerprint", os.path.basename(os.environ.get("PNR_LIB_PATH", "libpnr.so")), fp)
