"""Round-5 verdict item 6, the table it asked for: what a PURE READ of the channel-major raw image delivers under every candidate mapping
of a ray's samples onto lanes (L lanes per ray x N / L samples per lane, 64 / L rays per wave; J channel rows in flight; W waves per
SIMD), at the coarse level's N = 64 (256-byte rows per ray and channel) and, for reference, at N = 192 -- beside the compositing kernels
themselves.  usage: python tools/composite_patterns.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panopticnerf_amd import benchlib, ops, synthetic

dev = torch.device("cuda:0")
R, CH = 65536, 81
rays = synthetic.camera_rays()[:R].to(dev)
for N in (64, 192):
    raw = ops.alloc_raw(CH, R * N, dev)
    raw.normal_()
    z = ops.stratified(rays, N)
    print("N = %d: %d rays x %d channels = %.2f GB" % (N, R, CH, R * N * CH * 4 / 1e9))
    for L in (8, 16, 32, 64):
        if N % (4 * L):
            continue
        for J in (4, 8, 16):
            row = []
            for W in (2, 3, 4, 8):
                g = max(benchlib.probe_raw_read_pattern(raw, R, N, L, J, W, 5) for _ in range(3))
                row.append("%d w/SIMD %5.0f" % (W, g))
            print("  %2d lanes x %2d samples, %d rays per wave (%4d B contiguous per wave-load), %2d rows in flight (GB/s): %s" %
                  (L, N // L, 64 // L, min(64 // L * N * 4, 64 * 16), J, "   ".join(row)), flush=True)
    # the compositing kernel itself (weights written at the coarse level, as the training forward runs it)
    def timed(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    ms = min(timed(lambda: ops.composite(raw, z, rays, 45, 32, True, None, None, None, 0, False, N == 64)) for _ in range(3))
    bytes_ = R * {64: 22428, 192: 65180}[N]         # bench.py's algorithmic bytes per ray (raw rows, z, maps; weights written at N = 64)
    print("  pnr_composite (%s): %.4f ms = %.0f GB/s = %.3f of 8 TB/s" % ("k_composite2<8, 2>, weights written" if N == 64 else "k_composite", ms,
                                                                         bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 8000))
    del raw
