"""Round-6 experiment: the TRAINING forward with one wave per SIMD and two (three) 32-sample tiles per wave --
the real kernel, k_mlp_fused<TRAIN> is generic in (tiles, waves) -- against the shipped 8 waves x 1 tile, same process,
interleaved repeats, saved tensors compared bit for bit.
  build:  tools/build_ab.sh tiles:"-DPNR_TRAIN_TILES_EXPERIMENT=1"
  run:    PNR_LIB_PATH=build/ab/libpnr_tiles.so python tools/train_tiles_experiment.py [rays=4096]
schedule 0 = 8 waves x 1 tile (HEAD), 3 = 4 waves x 2 tiles (256 VGPR + 216 AGPR, no scratch), 4 = 4 waves x 3 tiles (384 samples per
weight pass; hipcc spills 295 registers: 656 B of scratch per lane -- timing only, and only to show what the spills cost)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, ops, synthetic

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
net = make_network(NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32, precision="bf16")).to(dev).train()
synthetic.trained_like_(net)
rays = synthetic.camera_rays()[::129][:R].contiguous().to(dev)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for lv, N in ((1, 192), (0, 64)):
    z = ops.stratified(rays, N)
    desc, img = net.packed(lv, dev, "bf16")
    scheds = [0, 3] + ([4] if (R * N) % 384 == 0 else [])
    res, out = {s: [] for s in scheds}, {}
    for rep in range(4):
        for sch in scheds:
            desc.schedule = sch
            res[sch].append(timed(lambda: ops.mlp_forward_train(desc, img, rays, z)))
            raw, acts = ops.mlp_forward_train(desc, img, rays, z)
            out[sch] = (raw.clone(), acts.view(torch.int16).clone())
    names = {0: "8 waves x 1 tile (HEAD)", 3: "4 waves x 2 tiles", 4: "4 waves x 3 tiles (spills; timing only)"}
    for sch in scheds:
        same = torch.equal(out[0][0], out[sch][0]) and torch.equal(out[0][1], out[sch][1])
        print("N=%3d %d samples  schedule %d %-40s " % (N, R * N, sch, names[sch]) + " ".join("%.4f" % t for t in res[sch]) +
              " ms   best %.4f   raw + saved tensors %s" % (min(res[sch]), "identical to HEAD" if same else "DIFFER"))
