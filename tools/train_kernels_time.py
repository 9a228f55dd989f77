"""The three training kernels alone -- the training forward (pnr_mlp_forward_train) at the benched training geometry, lock-step (pnr_mlp_desc.schedule 0 / 1) against
ping-pong (schedule 2), same process, interleaved repeats; checks that raw and the saved tensors are bit-identical.
usage: python tools/train_kernels_time.py [rays=4096]"""
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
    res, out = {0: [], 2: []}, {}
    for rep in range(4):
        for sch in (0, 2):
            desc.schedule = sch
            res[sch].append(timed(lambda: ops.mlp_forward_train(desc, img, rays, z)))
            raw, acts = ops.mlp_forward_train(desc, img, rays, z)
            out[sch] = (raw.clone(), acts.view(torch.int16).clone())
    same = torch.equal(out[0][0], out[2][0]) and torch.equal(out[0][1], out[2][1])
    # the data-gradient pass and the weight-gradient kernel on the same level (what PNR_LIB_PATH builds are compared on)
    desc.schedule = 0
    raw, acts = ops.mlp_forward_train(desc, img, rays, z)
    _, img_b = net.packed_bwd(lv, dev)
    d_raw = torch.randn_like(raw) * 1e-3
    dys = ops.mlp_backward(desc, img_b, d_raw, acts, R, N)
    shapes = {n: p.shape for n, p in net.nerf(lv).named_parameters()}
    tb = [timed(lambda: ops.mlp_backward(desc, img_b, d_raw, acts, R, N)) for _ in range(4)]
    tw = [timed(lambda: ops.mlp_wgrad(desc, acts, dys, R * N, shapes)) for _ in range(4)]
    print("N=%3d mlp_backward: " % N + " ".join("%.4f" % t for t in tb) + " ms   best %.4f | mlp_wgrad: " % min(tb) + " ".join("%.4f" % t for t in tw) +
          " ms   best %.4f" % min(tw))
    for sch in (0, 2):
        print("N=%3d schedule %d (%s): " % (N, sch, "lock-step" if sch == 0 else "ping-pong") + " ".join("%.4f" % t for t in res[sch]) +
              " ms   best %.4f   outputs %s" % (min(res[sch]), "identical" if same else "DIFFER"))
