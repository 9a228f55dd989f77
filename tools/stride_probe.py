"""Does the channel stride of the channel-major raw image matter?  raw (ch, R*N) with stride R*N*4 B = 48 MiB at the fine
level puts the 81 channel rows of a ray 3*2^24 B apart.  Times pnr_composite for padded channel strides (direct C-ABI
calls: the stride is an argument), all buffers allocated up front, MLP-written and random contents, interleaved repeats."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, ops, synthetic, _lib
dev = torch.device("cuda:0")
net = make_network(NS(N_importance=128, num_classes=45, num_instances=32)).eval()
synthetic.trained_like_(net)
R, N, C, K = 65536, 192, 45, 32
S, ch = R * N, 81
rays = synthetic.camera_rays()[:R].to(dev)
z = ops.stratified(rays, N)
desc, img = net.packed(1, dev)
lib = _lib.load()
p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
f32 = dict(device=dev, dtype=torch.float32)
o = dict(rgb=torch.empty((R, 3), **f32), depth=torch.empty(R, **f32), acc=torch.empty(R, **f32), sem=torch.empty((R, C), **f32),
         inst=torch.empty((R, K), **f32))
pads = [int(a) for a in (sys.argv[1:] or ["0", "64", "4160", "65600", "327744", "1048640"])]
bufs = {pd: torch.empty(ch * (S + pd), **f32) for pd in pads}
nbytes = R * (4 * N * (4 + C + K + 1) + 4 * (5 + C + K) + 32)

def comp(pd):
    _lib.check(lib.pnr_composite(p(bufs[pd]), 1, S + pd, p(z), p(rays), p(None), p(None), p(None), R, N, C, K, 0, 0, p(o["rgb"]),
                                 p(o["depth"]), p(o["acc"]), p(None), p(o["sem"]), p(o["inst"]), p(None), p(None), st), "composite")

def timed(pd, n=10):
    comp(pd); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): comp(pd)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for kind in ("mlp-written", "random", "mlp-written"):
    ms = ctypes.c_float(0.0)
    for pd in pads:
        if kind == "random": bufs[pd].normal_()
        else: _lib.check(lib.pnr_mlp_forward(ctypes.byref(desc), p(img), p(rays), p(z), R, N, p(bufs[pd]), 1, S + pd, st), "mlp")
    torch.cuda.synchronize()
    res = {pd: [] for pd in pads}
    for rep in range(3):
        for pd in pads: res[pd].append(timed(pd))
    print(f"contents: {kind}")
    for pd in pads:
        print(f"   pad {pd:8d} floats: " + "  ".join(f"{t:6.3f} ms" for t in res[pd]) + f"   best {nbytes / min(res[pd]) / 1e9:5.2f} TB/s", flush=True)
