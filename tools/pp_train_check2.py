import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
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
mode = sys.argv[1]
outs = []
for rep in range(4):
    if mode == "train":
        raw, acts = ops.mlp_forward_train(desc, img, rays, z)
    else:
        raw = ops.mlp_forward(desc, img, rays, z).clone()
    torch.cuda.synchronize()
    outs.append(raw.clone())
S = R * N
for i in range(1, 4):
    d = (outs[i] != outs[0]).any(0).nonzero().flatten()
    print(mode, "variant", os.environ.get("PNR_MLP_VARIANT"), "rep", i, "differing samples:", d.numel(), "groups:", sorted(set((d // 256).tolist()))[:12], "of", (S + 255) // 256, "min/max sample", (int(d.min()), int(d.max())) if d.numel() else None, flush=True)
import os
''' % ROOT
CHILD = "import os\n" + CHILD
for mode, v in (("train", "2"), ("infer", "1"), ("infer", "0")):
    subprocess.run([sys.executable, "-c", CHILD, mode], env=dict(os.environ, PNR_MLP_VARIANT=v), timeout=120)
