"""Round-3 verdict item 2a, measured: do the INDEPENDENT halves of a training step's backward -- fine-level k_mlp_bwd + k_wgrad and
coarse-level k_mlp_bwd + k_wgrad -- gain from running on two streams?  Same kernels, same buffers, one process:
  serial      fine bwd, fine wgrad, coarse bwd, coarse wgrad on one stream (what train.LevelFn does)
  two streams fine on stream A, coarse on stream B, no CU masks (the dispatcher interleaves workgroups as CUs free up)
  masked a/b  the same with hipExtStreamCreateWithCUMask: fine on a CUs, coarse on b CUs (contiguous bits, and bits interleaved
              so that every XCD contributes to both partitions)
  pipelined   the fine level cut into two half-batches; wgrad(half 1) on stream B beside bwd(half 2) on stream A
The streams are made here, through ctypes on libamdhip64 (the product library never creates a stream), and handed to torch as
ExternalStreams, so ops._stream() launches on them.
usage: python tools/train_overlap_probe.py [rays=4096]"""
import ctypes
import os
import sys
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
hip = ctypes.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    words = (ctypes.c_uint32 * ((NCU + 31) // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), words)
    assert rc == 0, "hipExtStreamCreateWithCUMask -> %d" % rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def level(lv, N, rr):
    z = ops.stratified(rr, N)
    desc, img = net.packed(lv, dev, "bf16")
    raw, acts = ops.mlp_forward_train(desc, img, rr, z)
    _, img_b = net.packed_bwd(lv, dev)
    d_raw = torch.randn_like(raw) * 1e-3
    shapes = {n: p.shape for n, p in net.nerf(lv).named_parameters()}
    n_rays = rr.shape[0]

    def bwd():
        return ops.mlp_backward(desc, img_b, d_raw, acts, n_rays, N)

    dys = bwd()

    def wgrad(d=None):
        return ops.mlp_wgrad(desc, acts, dys if d is None else d, n_rays * N, shapes)

    return NS(bwd=bwd, wgrad=wgrad)


fine, coarse = level(1, 192, rays), level(0, 64, rays)
h1, h2 = level(1, 192, rays[: R // 2].contiguous()), level(1, 192, rays[R // 2:].contiguous())
main = torch.cuda.current_stream()


def wall(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(n):
        fn()
    e1.record(main); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def serial():
    fine.bwd(); fine.wgrad(); coarse.bwd(); coarse.wgrad()


def forked(sa, sb):
    def run():
        ev = torch.cuda.Event(); ev.record(main)
        for s, lv in ((sa, fine), (sb, coarse)):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                lv.bwd(); lv.wgrad()
            done = torch.cuda.Event(); done.record(s); main.wait_event(done)
    return run


def pipelined(sa, sb):
    def run():
        ev = torch.cuda.Event(); ev.record(main)
        sa.wait_event(ev); sb.wait_event(ev)
        with torch.cuda.stream(sa):
            h1.bwd(); e1 = torch.cuda.Event(); e1.record(sa)
            h2.bwd(); e2 = torch.cuda.Event(); e2.record(sa)
            coarse.bwd(); e3 = torch.cuda.Event(); e3.record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(e1); h1.wgrad()
            sb.wait_event(e2); h2.wgrad()
            sb.wait_event(e3); coarse.wgrad()
        for s in (sa, sb):
            done = torch.cuda.Event(); done.record(s); main.wait_event(done)
    return run


def halves_serial():
    h1.bwd(); h1.wgrad(); h2.bwd(); h2.wgrad(); coarse.bwd(); coarse.wgrad()


print("%d rays, %d CUs" % (R, NCU))
rows = [("serial, one stream", serial)]
rows.append(("two streams, no masks", forked(torch.cuda.Stream(), torch.cuda.Stream())))
for a in (192, 160, 128):
    rows.append(("masked %d / %d contiguous" % (a, NCU - a), forked(masked_stream(range(a)), masked_stream(range(a, NCU)))))
    k = NCU // (NCU - a) if NCU - a else 1
    small = [b for b in range(NCU) if (b // 8) % k == k - 1][: NCU - a]
    big = [b for b in range(NCU) if b not in set(small)]
    rows.append(("masked %d / %d interleaved by 8" % (len(big), len(small)), forked(masked_stream(big), masked_stream(small))))
rows.append(("half batches, serial", halves_serial))
rows.append(("half batches, bwd stream + wgrad stream, no masks", pipelined(torch.cuda.Stream(), torch.cuda.Stream())))
for a in (128, 96):
    rows.append(("half batches, bwd on %d CUs + wgrad on %d CUs" % (a, NCU - a), pipelined(masked_stream(range(a)), masked_stream(range(a, NCU)))))
for rep in range(2):
    for name, fn in rows:
        print("%-58s %.4f ms" % (name, wall(fn)), flush=True)
