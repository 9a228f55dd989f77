"""Diagnostic: is the ping-pong TRAIN forward deterministic and equal to the lock-step one (raw and saved activations)?"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch, hashlib
sys.path.insert(0, %r)
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, ops, synthetic
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = make_network(NS(N_importance=128, num_classes=5, num_instances=3)).train().to(dev)
synthetic.trained_like_(net, 0.05)
R = int(sys.argv[1])
rays = synthetic.camera_rays()[:: (1408 * 376) // R][:R].contiguous().to(dev)
for N in (64, 192):
    z = ops.stratified(rays, N)
    desc, img = net.packed(1, dev)
    outs = []
    for rep in range(3):
        raw, acts = ops.mlp_forward_train(desc, img, rays, z)
        torch.cuda.synchronize()
        outs.append((raw.clone(), acts.clone()))
    a_off, _ = ops.train_layout(desc, R * N)
    same_raw = all(torch.equal(outs[0][0], o[0]) for o in outs[1:])
    same_act = all(torch.equal(outs[0][1].view(torch.int16), o[1].view(torch.int16)) for o in outs[1:])
    h = hashlib.sha1(outs[0][0].cpu().numpy().tobytes()).hexdigest()[:12], hashlib.sha1(outs[0][1].view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
    nan = torch.isnan(outs[0][1].float()).sum().item()
    print("variant", sys.argv[2], "R", R, "N", N, "deterministic raw/acts:", same_raw, same_act, "sha raw/acts:", h, "nan in acts:", nan, flush=True)
''' % ROOT
for R in ("510", "512"):
    for v in ("0", "2"):
        subprocess.run([sys.executable, "-c", CHILD, R, v], env=dict(os.environ, PNR_MLP_VARIANT=v), timeout=120)
