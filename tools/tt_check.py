"""k_mlp_tt (two-tile assembly kernel, plan 2) against k_mlp_pp<fused, plan 1> on the same network, rays and z: the per-tile records
(Q, semantic / instance logit sums) and the per-sample quadruples (lw, r, g, b) must agree BIT FOR BIT.  Prints where they do not.
  python tools/tt_check.py [R] [N] [--time]"""
import ctypes
import os
import sys
from types import SimpleNamespace as NS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from panopticnerf_amd import _lib, benchlib, make_network, ops, synthetic  # noqa: E402


def tiles(desc, img, rays, z):
    lib = _lib.load()
    R, N = z.shape
    nbytes = lib.pnr_mlp_forward_composite_workspace_bytes(ctypes.byref(desc), R, N, 0)
    ws = torch.full((int(nbytes),), 0xAB, device=z.device, dtype=torch.uint8)
    _lib.check(lib.pnr_mlp_forward_tiles(ctypes.byref(desc), ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(rays.data_ptr()),
                                         ctypes.c_void_p(z.data_ptr()), R, N, ctypes.c_void_p(ws.data_ptr()),
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "pnr_mlp_forward_tiles")
    torch.cuda.synchronize()
    return ws


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    R = int(args[0]) if args else 4096
    N = int(args[1]) if len(args) > 1 else 192
    C, K = 45, 32
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = make_network(NS(N_importance=128, num_classes=C, num_instances=K)).eval()
    synthetic.trained_like_(net)
    rays = synthetic.camera_rays()[:: max(1, 529408 // R)][:R].contiguous().to(dev)
    z = ops.stratified(rays, N)
    d1, i1 = net.packed(1, dev, fused=1)
    d2, i2 = net.packed(1, dev, fused=2)
    print("plans:", d1.plan, d2.plan, "image bytes:", i1.numel(), i2.numel(), flush=True)
    assert d2.plan == 2, "the geometry has no two-tile plan"
    w1 = tiles(d1, i1, rays, z)
    w2 = tiles(d2, i2, rays, z)
    S = R * N
    rf = (1 + C + K + 3) & ~3
    ntile_pad = (S + 255) // 256 * 8
    nt = (S + 31) // 32
    rec1 = w1[: ntile_pad * rf * 4].view(torch.float32).reshape(ntile_pad, rf)[:nt]
    rec2 = w2[: ntile_pad * rf * 4].view(torch.float32).reshape(ntile_pad, rf)[:nt]
    ps1 = w1[ntile_pad * rf * 4: ntile_pad * rf * 4 + S * 16].view(torch.float32).reshape(S, 4)
    ps2 = w2[ntile_pad * rf * 4: ntile_pad * rf * 4 + S * 16].view(torch.float32).reshape(S, 4)
    ok = True
    for name, a, b in (("Q", rec1[:, 0], rec2[:, 0]), ("sem logit sums", rec1[:, 1:1 + C], rec2[:, 1:1 + C]),
                       ("inst logit sums", rec1[:, 1 + C:1 + C + K], rec2[:, 1 + C:1 + C + K]), ("lw", ps1[:, 0], ps2[:, 0]),
                       ("rgb", ps1[:, 1:], ps2[:, 1:])):
        same = a.view(torch.int32) == b.view(torch.int32)
        bad = int((~same).sum())
        ok = ok and bad == 0
        msg = "%-16s %9d values, %9d differ" % (name, a.numel(), bad)
        if bad:
            d = (a.double() - b.double()).abs()
            idx = torch.nonzero(~same.reshape(a.shape))[:4].tolist()
            msg += "  max |diff| %.3e (ref scale %.3e) first at %s: %s vs %s" % (float(d.max()), float(a.abs().max()), idx,
                                                                                 [float(a[tuple(i)]) for i in idx][:4], [float(b[tuple(i)]) for i in idx][:4])
        print(msg, flush=True)
    print("BIT-IDENTICAL" if ok else "MISMATCH", flush=True)
    if "--time" in sys.argv:
        for tag, d, im in (("plan 1 (k_mlp_pp)", d1, i1), ("plan 2 (k_mlp_tt)", d2, i2)):
            benchlib.time_mlp_forward_tiles(d, im, rays, z, 2)
            ms, mhz = min((benchlib.time_mlp_forward_tiles(d, im, rays, z, 5) for _ in range(3)), key=lambda t: t[0])
            print("%-20s %8.3f ms  %6.0f MHz  %7.1f Msamples/s" % (tag, ms, mhz, S / ms / 1e3), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
