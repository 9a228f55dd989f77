"""Per-unit timeline of k_mlp_tt (trace build: s_memtime at every unit's start, workgroup 0's first wave, its last group).
  python tools/tt_trace.py  ->  cycles per unit next to 32 x MFMAs (what the matrix pipe needs)"""
import ctypes
import os
import sys
from types import SimpleNamespace as NS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from panopticnerf_amd import _lib, make_network, ops, synthetic  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "panopticnerf_amd", "csrc", "asm"))
import gen_mlp_tt as G  # noqa: E402


def main():
    abl = int(sys.argv[1]) if len(sys.argv) > 1 else 0       # timing-only ablation of the trace build (library built with EXTRA_TT=abl)
    dev = torch.device("cuda:0")
    R, N = 65536, 192
    torch.manual_seed(0)
    net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
    synthetic.trained_like_(net)
    rays = synthetic.camera_rays()[:R].to(dev)
    z = ops.stratified(rays, N)
    desc, img = net.packed(1, dev, fused=2)
    desc = _lib.MlpDesc.from_buffer_copy(bytes(desc))
    clk = torch.zeros(64 + 256, dtype=torch.int32, device=dev)
    desc.clk_probe[0] = clk.data_ptr() & 0xffffffff
    desc.clk_probe[1] = clk.data_ptr() >> 32
    desc.flags = 0x7A00 + (abl << 4)        # bits 4..6: the ablation (bit 0 is PNR_MLP_SOFTMAX)
    lib = _lib.load()
    nbytes = lib.pnr_mlp_forward_composite_workspace_bytes(ctypes.byref(desc), R, N, 0)
    ws = torch.empty(int(nbytes), device=dev, dtype=torch.uint8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(3):
        if it == 2:
            e0.record()
        _lib.check(lib.pnr_mlp_forward_tiles(ctypes.byref(desc), ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(rays.data_ptr()),
                                             ctypes.c_void_p(z.data_ptr()), R, N, ctypes.c_void_p(ws.data_ptr()),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pnr_mlp_forward_tiles")
    e1.record()
    torch.cuda.synchronize()
    print("trace build%s: %.3f ms per launch" % (" (ABLATION %d: 1 = no pieces, 2 = no hand-over, 4 = no packs)" % abl if abl else "", e0.elapsed_time(e1)))
    from panopticnerf_amd import benchlib
    d2, i2 = net.packed(1, dev, fused=2)
    benchlib.time_mlp_forward_tiles(d2, i2, rays, z, 2)
    ms, mhz = min((benchlib.time_mlp_forward_tiles(d2, i2, rays, z, 5) for _ in range(3)), key=lambda x: x[0])
    print("production build: %.3f ms at %.0f MHz = %.0f cycles per 256-sample group on average (192 groups per workgroup)" % (ms, mhz, ms * 1e-3 * mhz * 1e6 / 192))
    t = [int(x) & 0xffffffff for x in clk.cpu().tolist()]
    wg = sorted(t[64:64 + 256])
    print("cycles of the 256 workgroups (first wave, start to end): min %d  median %d  max %d  (max / min %.3f); slowest five %s" % (
        wg[0], wg[128], wg[-1], wg[-1] / max(1, wg[0]), wg[-5:]))
    per_xcd = [sorted(t[64 + x: 64 + 256: 8]) for x in range(8)]
    print("median per workgroup-id mod 8 (XCD round-robin):", [p[len(p) // 2] for p in per_xcd])
    g = G.Gen(2, 1, "x")
    names = ["%s %s" % (g.layers[u["layer"]]["name"], u["blocks"]) for u in g.units] + ["side-queue drain", "advance state", "(end)"]
    nm = [2 * len(g.unit_frags(u)) for u in g.units] + [0, 0, 0]
    tot = (t[len(names) - 1] - t[0]) & 0xffffffff
    print("group: %d cycles for %d MFMAs = %.2f cycles per MFMA" % (tot, sum(nm), tot / sum(nm)))
    ends = [(t[56 + i]) for i in range(8)]
    order = sorted(ends)
    print("last eight groups' end-to-end cycles:", [(order[i + 1] - order[i]) & 0xffffffff for i in range(7)])
    print("%-4s %-22s %6s %8s %8s %7s" % ("unit", "what", "MFMAs", "cycles", "32xMFMA", "excess"))
    nu = len(g.units)
    print("tail: local weights of both tiles %d, the FMA chains of both tiles %d, exchange + stores %d cycles" % ((t[nu + 4] - t[nu]) & 0xffffffff, (t[nu + 5] - t[nu + 4]) & 0xffffffff,
                                                                                 (t[nu + 1] - t[nu + 5]) & 0xffffffff))
    for i in range(len(names) - 1):
        d = (t[i + 1] - t[i]) & 0xffffffff
        print("%-4d %-22s %6d %8d %8d %7d" % (i, names[i], nm[i], d, 32 * nm[i], d - 32 * nm[i]))


if __name__ == "__main__":
    main()
