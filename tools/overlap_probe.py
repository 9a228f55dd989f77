"""Can the fine level of chunk c and the coarse level of chunk c + 1 run SIDE BY SIDE?  Two k_mlp_tt launches on two streams, each capped to a
share of the compute units (PNR_MLP_WG_CAP: the persistent grid is one workgroup per CU; 192 : 64 = the levels' 192 : 64 samples per ray),
against the same two launches back to back on the whole device.  usage: python tools/overlap_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import _lib, make_network, ops, synthetic

dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
synthetic.trained_like_(net)
net = net.to(dev)
R = 66560
rays = synthetic.camera_rays()[:R].contiguous().to(dev)
lv = {}
for name, level, N in (("coarse", 0, 64), ("fine", 1, 192)):
    z = ops.stratified(rays, N)
    desc, img = net.packed(level, dev, fused=True)
    nbytes = lib.pnr_mlp_forward_composite_workspace_bytes(ctypes.byref(desc), R, N, 0)
    lv[name] = dict(z=z, desc=desc, img=img, N=N, ws=torch.empty(int(nbytes), device=dev, dtype=torch.uint8))
p = lambda t: ctypes.c_void_p(t.data_ptr())


def capped(desc, cap):
    d = _lib.MlpDesc()
    ctypes.memmove(ctypes.byref(d), ctypes.byref(desc), ctypes.sizeof(d))
    d.flags = (desc.flags & 0xFFFF) | ((cap & 0x1FF) << 16)
    return d


def launch(name, cap, stream):
    L = lv[name]
    d = capped(L["desc"], cap)
    rc = lib.pnr_mlp_forward_tiles(ctypes.byref(d), p(L["img"]), p(rays), p(L["z"]), R, L["N"], p(L["ws"]), ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0, lib.pnr_last_error()


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ref = {}
for name in lv:
    launch(name, 0, torch.cuda.current_stream())
    torch.cuda.synchronize()
    ref[name] = lv[name]["ws"].clone()


def timed(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def serial():
    launch("coarse", 0, torch.cuda.current_stream())
    launch("fine", 0, torch.cuda.current_stream())


def side_by_side(cf, cc):
    def fn():
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(cur)
        sa.wait_event(ev); sb.wait_event(ev)
        launch("fine", cf, sa)
        launch("coarse", cc, sb)
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        ea.record(sa); eb.record(sb)
        cur.wait_event(ea); cur.wait_event(eb)
    return fn


print("coarse alone %.3f ms, fine alone %.3f ms" % (timed(lambda: launch("coarse", 0, torch.cuda.current_stream())),
                                                   timed(lambda: launch("fine", 0, torch.cuda.current_stream()))))
for rep in range(2):
    print("back to back on the whole device: %.3f ms" % timed(serial))
    for cf, cc in ((192, 64), (196, 60), (188, 68), (200, 56)):
        t = timed(side_by_side(cf, cc))
        torch.cuda.synchronize()
        same = all(torch.equal(ref[k][: lv[k]["ws"].numel() - 0], lv[k]["ws"]) for k in lv)
        print("side by side, fine on %d + coarse on %d workgroups: %.3f ms   workspaces %s" % (cf, cc, t, "bit-identical" if same else "DIFFER"), flush=True)
