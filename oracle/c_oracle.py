"""ctypes binding of oracle/libpnr_oracle.so (the strict-order C restatement).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/pnr_oracle.c.  PARITY UNPINNED: the
reference mount holds no source; the arithmetic follows SURVEY.md 8a.
Everything here takes and returns numpy arrays (C-contiguous, float32 / int32).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpnr_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile the C oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "pnr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpnr_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.pnro_version.restype = ctypes.c_int
    return _lib


def _fp(a):
    return None if a is None else a.ctypes.data_as(_f)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_i)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def gen_rays(intr, c2w, width, height, near, far, pix=None):
    """intr (4) fx fy cx cy, c2w (3,4) -> rays (R,8); pix: int32 linear pixel indices or None (whole frame)."""
    intr = np.ascontiguousarray(intr, np.float32).reshape(4)
    c2w = np.ascontiguousarray(c2w, np.float32).reshape(12)
    if pix is not None:
        pix = np.ascontiguousarray(pix, np.int32)
    R = int(pix.shape[0]) if pix is not None else int(width) * int(height)
    rays = np.empty((R, 8), np.float32)
    lib().pnro_gen_rays(_fp(intr), _fp(c2w), int(width), int(height), ctypes.c_float(near), ctypes.c_float(far),
                        pix.ctypes.data_as(ctypes.c_void_p) if pix is not None else None, ctypes.c_int64(R), _fp(rays))
    return rays


def stratified(rays, n_samples, lindisp=False, t_rand=None):
    rays = _f32(rays).reshape(-1, 8)
    t_rand = _f32(t_rand)
    R = rays.shape[0]
    z = np.empty((R, n_samples), np.float32)
    lib().pnro_stratified(_fp(rays), ctypes.c_int64(R), int(n_samples), int(bool(lindisp)),
                          _fp(t_rand), _fp(z))
    return z


def points(rays, z):
    rays = _f32(rays).reshape(-1, 8)
    z = _f32(z)
    R, N = z.shape
    pts = np.empty((R, N, 3), np.float32)
    lib().pnro_points(_fp(rays), _fp(z), ctypes.c_int64(R), int(N), _fp(pts))
    return pts


def embed(x, L):
    x = _f32(x).reshape(-1, 3)
    out = np.empty((x.shape[0], 3 + 6 * L), np.float32)
    lib().pnro_embed(_fp(x), ctypes.c_int64(x.shape[0]), int(L), _fp(out))
    return out


def composite(raw, z, rays, C, K, channel_major=False, noise=None, label_sem=None,
              label_inst=None, sem_mode=0, white_bkgd=False):
    """raw: (R,N,4+C+K) sample-major, or (4+C+K, R*N) when channel_major."""
    raw = _f32(raw)
    z = _f32(z)
    rays = _f32(rays).reshape(-1, 8)
    R, N = z.shape
    ch = 4 + C + K
    if channel_major:
        assert raw.shape == (ch, R * N)
        ss, sc = 1, R * N
    else:
        assert raw.reshape(R, N, ch).shape == (R, N, ch)
        ss, sc = ch, 1
    noise = _f32(noise)
    label_sem = _i32(label_sem)
    label_inst = _i32(label_inst)
    out = dict(rgb=np.empty((R, 3), np.float32), depth=np.empty(R, np.float32),
               acc=np.empty(R, np.float32), weights=np.empty((R, N), np.float32),
               semantic=np.zeros((R, C), np.float32), instance=np.zeros((R, K), np.float32),
               fix_semantic=np.zeros((R, C), np.float32), fix_instance=np.zeros((R, K), np.float32))
    lib().pnro_composite(_fp(raw), ctypes.c_int64(ss), ctypes.c_int64(sc), _fp(z), _fp(rays),
                         _fp(noise), _ip(label_sem), _ip(label_inst), ctypes.c_int64(R), int(N),
                         int(C), int(K), int(sem_mode), int(bool(white_bkgd)),
                         _fp(out["rgb"]), _fp(out["depth"]), _fp(out["acc"]), _fp(out["weights"]),
                         _fp(out["semantic"]) if C else None, _fp(out["instance"]) if K else None,
                         _fp(out["fix_semantic"]) if (C and label_sem is not None) else None,
                         _fp(out["fix_instance"]) if (K and label_inst is not None) else None)
    return out


def sample_pdf(z, weights, n_importance, u=None):
    z = _f32(z)
    weights = _f32(weights)
    u = _f32(u)
    R, Nc = z.shape
    zs = np.empty((R, n_importance), np.float32)
    inds = np.empty((R, n_importance), np.int32)
    lib().pnro_sample_pdf(_fp(z), _fp(weights), _fp(u), ctypes.c_int64(R), int(Nc),
                          int(n_importance), _fp(zs), _ip(inds))
    return zs, inds


def merge_sorted(z, zs):
    z = _f32(z)
    zs = _f32(zs)
    R, Nc = z.shape
    Nf = zs.shape[1]
    out = np.empty((R, Nc + Nf), np.float32)
    lib().pnro_merge_sorted(_fp(z), _fp(zs), ctypes.c_int64(R), int(Nc), int(Nf), _fp(out))
    return out


def bbox_hits(rays, box, max_hits):
    rays = _f32(rays).reshape(-1, 8)
    box = _f32(box).reshape(-1, 15)
    R, M = rays.shape[0], box.shape[0]
    hit_t = np.empty((R, max_hits, 2), np.float32)
    hit_box = np.empty((R, max_hits), np.int32)
    hit_count = np.empty(R, np.int32)
    lib().pnro_bbox_hits(_fp(rays), ctypes.c_int64(R), _fp(box), int(M), int(max_hits),
                         _fp(hit_t), _ip(hit_box), _ip(hit_count))
    return hit_t, hit_box, hit_count


def restrict_rays(rays, hit_t, hit_count):
    rays = _f32(rays).reshape(-1, 8)
    hit_t = _f32(hit_t)
    hit_count = _i32(hit_count)
    out = np.empty_like(rays)
    lib().pnro_restrict_rays(_fp(rays), ctypes.c_int64(rays.shape[0]), _fp(hit_t), _ip(hit_count), int(hit_t.shape[1]), _fp(out))
    return out


def sample_labels(z, hit_t, hit_box, hit_count, box_ids):
    z = _f32(z)
    hit_t = _f32(hit_t)
    hit_box = _i32(hit_box)
    hit_count = _i32(hit_count)
    box_ids = _i32(box_ids).reshape(-1, 2)
    R, N = z.shape
    mh = hit_box.shape[1]
    ls = np.empty((R, N), np.int32)
    li = np.empty((R, N), np.int32)
    lib().pnro_sample_labels(_fp(z), ctypes.c_int64(R), int(N), _fp(hit_t), _ip(hit_box),
                             _ip(hit_count), int(mh), _ip(box_ids), _ip(ls), _ip(li))
    return ls, li


def bf16_round(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().pnro_bf16_round(_fp(x), ctypes.c_int64(x.size), _fp(out))
    return out


def linear(x, W, b, relu=False, emulate_bf16=False):
    x = _f32(x)
    W = _f32(W)
    b = _f32(b)
    n, k = x.shape
    o = W.shape[0]
    y = np.empty((n, o), np.float32)
    lib().pnro_linear(_fp(x), ctypes.c_int64(n), int(k), _fp(W), _fp(b), int(o), int(bool(relu)),
                      int(bool(emulate_bf16)), _fp(y))
    return y
