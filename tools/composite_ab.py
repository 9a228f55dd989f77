"""Same-process A/B of two libpnr builds on pnr_composite: identical buffers (the physical placement of the 4 GB raw
image alone moves the result by ~4 % between processes), interleaved repeats, with and without bbox labels.
usage: python tools/composite_ab.py <libA.so> <libB.so> [N=192]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from panopticnerf_amd import _lib, ops, synthetic
dev = torch.device("cuda:0")
libs = {}
for path in sys.argv[1:3]:
    lib = ctypes.CDLL(os.path.abspath(path))
    res, args = _lib.SIGNATURES["pnr_composite"]
    lib.pnr_composite.restype, lib.pnr_composite.argtypes = res, args
    libs[os.path.basename(path)] = lib
R, N, C, K = 65536, (int(sys.argv[3]) if len(sys.argv) > 3 else 192), 45, 32
S, ch = R * N, 81
rays = synthetic.camera_rays()[:R].to(dev)
z = ops.stratified(rays, N)
raw = ops.alloc_raw(ch, S, dev)
raw.normal_()
p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
f32 = dict(device=dev, dtype=torch.float32)
o = dict(rgb=torch.empty((R, 3), **f32), depth=torch.empty(R, **f32), acc=torch.empty(R, **f32), sem=torch.empty((R, C), **f32),
         inst=torch.empty((R, K), **f32), fs=torch.empty((R, C), **f32), fi=torch.empty((R, K), **f32), w=torch.empty((R, N), **f32))
g = torch.Generator(device=dev).manual_seed(0)
lab_s = torch.where(torch.rand((R, N), device=dev, generator=g) < 0.3, torch.randint(0, C, (R, N), device=dev, generator=g), -1).int()
lab_i = torch.where(lab_s >= 0, torch.randint(0, K, (R, N), device=dev, generator=g), -1).int()

def run(lib, labels):
    rc = lib.pnr_composite(p(raw), 1, raw.stride(0), p(z), p(rays), p(None), p(lab_s if labels else None), p(lab_i if labels else None),
                           R, N, C, K, 0, 0, p(o["rgb"]), p(o["depth"]), p(o["acc"]), p(o["w"] if labels else None), p(o["sem"]), p(o["inst"]),
                           p(o["fs"] if labels else None), p(o["fi"] if labels else None), st)
    assert rc == 0

def timed(lib, labels, n=10):
    run(lib, labels); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run(lib, labels)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for labels in (False, True):
    nbytes = R * (4 * N * (ch + 1) + (3 * 4 * N if labels else 0) + 4 * (5 + (2 if labels else 1) * (C + K)) + 32)
    res = {k: [] for k in libs}
    chk = {}
    for rep in range(4):
        for k, lib in libs.items():
            res[k].append(timed(lib, labels))
            chk[k] = (o["rgb"].double().sum() + o["sem"].double().sum() + o["depth"].double().sum()).item()
    vals = list(chk.values())
    same = abs(vals[0] - vals[-1]) <= 1e-6 * abs(vals[0])
    for k in libs:
        print(f"labels={int(labels)} {k:24s} " + " ".join(f"{t:6.3f}" for t in res[k]) + f" ms   best {nbytes / min(res[k]) / 1e9:5.2f} TB/s   outputs {'agree' if same else 'DIFFER'}")
