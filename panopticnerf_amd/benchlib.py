"""ctypes binding of libpnr_bench.so (include/pnr_bench.h): hipEvent timing of the MLP launches and the device ceilings that
bench.py and tools/ report.  NOT imported by the package: the product path (libpnr.so) never synchronises; everything here does.
The library times the entry points of the libpnr.so that panopticnerf_amd._lib has loaded (their addresses are handed over)."""
import ctypes
import os

import torch

from . import _lib, ops

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNR_BENCH_LIB_PATH") or os.path.join(_HERE, "libpnr_bench.so")
c_f, c_i64, c_int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
_fp = ctypes.POINTER(ctypes.c_float)
SIGNATURES = {
    "pnrb_last_error": (ctypes.c_char_p, []),
    "pnrb_bind": (c_int, [c_f, c_f]),
    "pnrb_time_mlp_forward": (c_int, [ctypes.POINTER(_lib.MlpDesc), c_f, c_f, c_f, c_i64, c_int, c_f, c_i64, c_i64, c_int, c_f, _fp, _fp, c_f]),
    "pnrb_time_mlp_forward_tiles": (c_int, [ctypes.POINTER(_lib.MlpDesc), c_f, c_f, c_f, c_i64, c_int, c_f, c_int, c_f, _fp, _fp, c_f]),
    "pnrb_probe_mfma_peak": (c_int, [c_int, c_int, c_f, _fp, _fp, c_f]),
    "pnrb_probe_mfma_order": (c_int, [c_int, c_int, c_f, _fp, _fp, c_f]),
    "pnrb_probe_raw_read": (c_int, [c_f, c_i64, c_i64, c_int, c_int, c_int, c_f, _fp, c_f]),
    "pnrb_probe_raw_read_pattern": (c_int, [c_f, c_i64, c_i64, c_int, c_int, c_int, c_int, c_int, c_int, c_f, _fp, c_f]),
}
_blib = None


def load():
    global _blib
    if _blib is not None:
        return _blib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(or `make -C panopticnerf_amd/csrc`)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    prod = _lib.load()
    addr = lambda f: ctypes.cast(f, ctypes.c_void_p)
    rc = lib.pnrb_bind(addr(prod.pnr_mlp_forward), addr(prod.pnr_mlp_forward_tiles))
    if rc != 0:
        raise RuntimeError("pnrb_bind failed: " + lib.pnrb_last_error().decode())
    _blib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (code %d): %s | %s" % (what, rc, load().pnrb_last_error().decode(errors="replace"),
                                                             _lib.load().pnr_last_error().decode(errors="replace")))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def time_mlp_forward(desc, packed, rays, z, raw, iters):
    """(mean ms per pnr_mlp_forward launch, mean shader MHz during the last one): hipEvents on the launch stream."""
    R, N = z.shape
    ms, mhz = ctypes.c_float(0.0), ctypes.c_float(0.0)
    with torch.cuda.device(z.device):
        scratch = torch.zeros(4, device=z.device, dtype=torch.int64)
        sc = ops._chk_raw(raw, ops.n_channels(desc), R * N)
        _check(load().pnrb_time_mlp_forward(ctypes.byref(desc), _p(packed), _p(rays), _p(z), R, N, _p(raw), 1, sc, int(iters),
                                            _p(scratch), ctypes.byref(ms), ctypes.byref(mhz), _stream()), "pnrb_time_mlp_forward")
    return float(ms.value), float(mhz.value)


def time_mlp_forward_tiles(desc, packed, rays, z, iters=5):
    """(mean ms per FUSED inference MLP launch -- pnr_mlp_forward_tiles, i.e. pnr_mlp_forward_composite without its combine
    kernel --, mean shader MHz during the last one)."""
    R, N = z.shape
    ms, mhz = ctypes.c_float(0.0), ctypes.c_float(0.0)
    with torch.cuda.device(z.device):
        nbytes = _lib.load().pnr_mlp_forward_composite_workspace_bytes(ctypes.byref(desc), R, N, 0)
        if nbytes < 0:
            raise RuntimeError("time_mlp_forward_tiles: unsupported geometry")
        ws = torch.empty(int(nbytes), device=z.device, dtype=torch.uint8)
        scratch = torch.zeros(4, device=z.device, dtype=torch.int64)
        _check(load().pnrb_time_mlp_forward_tiles(ctypes.byref(desc), _p(packed), _p(rays), _p(z), R, N, _p(ws), int(iters), _p(scratch),
                                                  ctypes.byref(ms), ctypes.byref(mhz), _stream()), "pnrb_time_mlp_forward_tiles")
    return float(ms.value), float(mhz.value)


def probe_mfma_order(pattern, iters=12000, device=None):
    """(TFLOP/s, shader MHz) of the register-only bf16 MFMA loop on random operands with operand-change order `pattern`
    (pnrb_probe_mfma_order: 0 both change every MFMA, 1 the two-tile order, 2 the snake, 3 only A changes, 4 only B)."""
    dev = torch.device(device if device is not None else "cuda")
    tf, mhz = ctypes.c_float(0.0), ctypes.c_float(0.0)
    with torch.cuda.device(dev):
        scratch = torch.zeros(4, device=dev, dtype=torch.int64)
        _check(load().pnrb_probe_mfma_order(int(pattern), int(iters), _p(scratch), ctypes.byref(tf), ctypes.byref(mhz), _stream()),
               "pnrb_probe_mfma_order")
    return float(tf.value), float(mhz.value)


def probe_mfma_peak(random_operands, iters=20000, device=None):
    """(TFLOP/s, shader MHz) a register-only bf16 MFMA loop sustains on this device."""
    dev = torch.device(device if device is not None else "cuda")
    tf, mhz = ctypes.c_float(0.0), ctypes.c_float(0.0)
    with torch.cuda.device(dev):
        scratch = torch.zeros(4, device=dev, dtype=torch.int64)
        _check(load().pnrb_probe_mfma_peak(int(bool(random_operands)), int(iters), _p(scratch), ctypes.byref(tf), ctypes.byref(mhz),
                                           _stream()), "pnrb_probe_mfma_peak")
    return float(tf.value), float(mhz.value)


def probe_raw_read_pattern(raw, n_rays, n_samples, lanes_per_ray, rows_in_flight=8, waves_per_simd=8, iters=5):
    """GB/s of a pure read of the channel-major raw image with `lanes_per_ray` lanes x n_samples / lanes_per_ray samples per ray."""
    gbs = ctypes.c_float(0.0)
    with torch.cuda.device(raw.device):
        scratch = torch.zeros(256, device=raw.device, dtype=torch.float32)
        _check(load().pnrb_probe_raw_read_pattern(_p(raw), ops._chk_raw(raw, raw.shape[0], n_rays * n_samples), int(n_rays), int(n_samples),
                                                  int(raw.shape[0]), int(lanes_per_ray), int(rows_in_flight), int(waves_per_simd), int(iters),
                                                  _p(scratch), ctypes.byref(gbs), _stream()), "pnrb_probe_raw_read_pattern")
    return float(gbs.value)


def probe_raw_read(raw, n_rays, n_samples, iters=5):
    """GB/s of a pure read of the channel-major raw image in k_composite's access order."""
    gbs = ctypes.c_float(0.0)
    with torch.cuda.device(raw.device):
        scratch = torch.zeros(256, device=raw.device, dtype=torch.float32)
        _check(load().pnrb_probe_raw_read(_p(raw), ops._chk_raw(raw, raw.shape[0], n_rays * n_samples), int(n_rays), int(n_samples),
                                          int(raw.shape[0]), int(iters), _p(scratch), ctypes.byref(gbs), _stream()), "pnrb_probe_raw_read")
    return float(gbs.value)
