"""A/B timing of fused-MLP builds/variants on the bench's dominant launch (fine level, 65536 rays x
192 samples, 8x256 + semantic(45) + instance(32) heads, bf16), each in its own process.
usage: python tools/mlp_variants.py [lib:variant ...]   lib = 'default' or a name under build/ab/"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
import bench
from types import SimpleNamespace as NS
from panopticnerf_amd import benchlib, make_network, ops, synthetic
from oracle import torch_oracle as to
dev = torch.device("cuda:0")
torch.manual_seed(0)
import os
NOSKIP = os.environ.get("PNR_NET_SKIPS") == "none"     # feasibility experiments on a network without the skip layer
net = make_network(NS(N_importance=128, num_classes=45, num_instances=32, **({"skips": []} if NOSKIP else {}))).eval()
synthetic.trained_like_(net)
rays = synthetic.camera_rays()[:65536].to(dev)
z = ops.stratified(rays, 192)
desc, img = net.packed(1, dev)
raw = ops.alloc_raw(81, 65536 * 192, dev)
benchlib.time_mlp_forward(desc, img, rays, z, raw, 2)
ms = min(benchlib.time_mlp_forward(desc, img, rays, z, raw, 5)[0] for _ in range(3))
fl = 65536 * 192 * bench.mlp_flops_per_sample()
# correctness spot check against the bf16-emulating oracle on 24 rays spread over the launch
idx = torch.arange(0, 65536, 2731)
oc = to.mlp_config(n_sem=45, n_inst=32, **({"skips": ()} if NOSKIP else {}))
ref = to.run_network({k: v.detach().cpu() for k, v in net.nerf_1.state_dict().items()}, oc, rays[idx].cpu(), z[idx].cpu(), emulate_bf16=True)
got = raw.unflatten(1, (65536, 192))[:, idx].permute(1, 2, 0).cpu()
err = (got - ref).abs().max().item()
print("%%-14s %%8.3f ms  %%7.1f TFLOP/s  %%6.1f Msamples/s  max|err| vs bf16 oracle %%.2e" %% (
    sys.argv[1], ms, fl / ms / 1e9, 65536 * 192 / ms / 1e3, err), flush=True)
''' % ROOT

for spec in (sys.argv[1:] or ["default:3"]):
    lib, _, v = spec.partition(":")
    env = dict(os.environ, PNR_MLP_VARIANT=v or "3")
    if lib != "default":
        env["PNR_LIB_PATH"] = os.path.join(ROOT, "build", "ab", "libpnr_%s.so" % lib)
    try:
        subprocess.run([sys.executable, "-c", CHILD, spec], env=env, check=False, timeout=90)
    except subprocess.TimeoutExpired:
        print("%-14s TIMEOUT (hang?)" % spec, flush=True)
