import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, ops, synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = make_network(NS(N_importance=128, num_classes=5, num_instances=3)).train().to(dev)
synthetic.trained_like_(net, 0.05)
R, N = 510, 192
rays = synthetic.camera_rays()[:: (1408 * 376) // R][:R].contiguous().to(dev)
z = ops.stratified(rays, N)
desc, img = net.packed(1, dev)
res = []
for rep in range(3):
    raw, acts = ops.mlp_forward_train(desc, img, rays, z)
    torch.cuda.synchronize()
    res.append((raw.clone(), acts.clone()))
torch.save([(r.cpu(), a.cpu()) for r, a in res], "/tmp/pp_%%s.pt" %% os.environ["PNR_MLP_VARIANT"])
''' % ROOT
for v in ("0", "2"):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PNR_MLP_VARIANT=v), timeout=120)
import torch
a, b = torch.load("/tmp/pp_0.pt"), torch.load("/tmp/pp_2.pt")
sys.path.insert(0, ROOT)
from panopticnerf_amd import ops
ref_raw, ref_acts = a[0]
S = 510 * 192
for rep in range(3):
    raw, acts = b[rep]
    bad = (raw != ref_raw)
    idx = bad.any(0).nonzero().flatten()
    print("rep", rep, "bad samples", idx.numel(), "bad channels per bad sample (min/max)", (int(bad[:, idx].sum(0).min()), int(bad[:, idx].sum(0).max())) if idx.numel() else None)
    if idx.numel():
        s = int(idx[0])
        print("  sample", s, "raw pp ", [round(float(x), 4) for x in raw[:8, s]])
        print("  sample", s, "raw ref", [round(float(x), 4) for x in ref_raw[:8, s]])
        print("  which channels bad:", bad[:, s].nonzero().flatten().tolist())
    ab = (acts.view(torch.int16) != ref_acts.view(torch.int16)).nonzero().flatten()
    print("  acts elements differing:", ab.numel(), "first/last", (int(ab[0]), int(ab[-1])) if ab.numel() else None, "of", acts.numel())
