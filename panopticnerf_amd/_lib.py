"""ctypes binding of libpnr.so (include/pnr.h).  There is NO fallback: if the library is
missing or a call fails, a RuntimeError is raised -- the product path never routes through
the oracle or any CPU implementation."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNR_LIB_PATH") or os.path.join(_HERE, "libpnr.so")   # override: A/B builds only

c_f = ctypes.c_void_p       # device pointers travel as integers
c_i64 = ctypes.c_int64
c_int = ctypes.c_int

PREC_BF16, PREC_FP32 = 0, 1
MLP_SOFTMAX, MLP_TRACE = 1, 0x7A00        # pnr_mlp_desc.flags (include/pnr.h PNR_MLP_*)


class LossCfg(ctypes.Structure):
    """pnr_loss_cfg (include/pnr.h)."""
    _fields_ = [("w_rgb", ctypes.c_float), ("w_depth", ctypes.c_float), ("w_sem", ctypes.c_float),
                ("w_fix_sem", ctypes.c_float), ("w_inst", ctypes.c_float), ("w_fix_inst", ctypes.c_float),
                ("depth_l2", ctypes.c_int32), ("fix_eps", ctypes.c_float), ("maps_are_prob", ctypes.c_int32)]


class MlpDesc(ctypes.Structure):
    """pnr_mlp_desc (include/pnr.h)."""
    _fields_ = [("D", ctypes.c_int32), ("W", ctypes.c_int32), ("skip", ctypes.c_int32),
                ("xyz_L", ctypes.c_int32), ("dir_L", ctypes.c_int32),
                ("n_sem", ctypes.c_int32), ("n_inst", ctypes.c_int32), ("head_W", ctypes.c_int32),
                ("precision", ctypes.c_int32), ("plan", ctypes.c_int32), ("head_tap", ctypes.c_int32),
                ("head_depth", ctypes.c_int32), ("schedule", ctypes.c_int32), ("clk_probe", ctypes.c_int32 * 2), ("flags", ctypes.c_int32)]


_fp = ctypes.POINTER(ctypes.c_float)
_fpp = ctypes.POINTER(_fp)


class MlpParamsHost(ctypes.Structure):
    """pnr_mlp_params_host (include/pnr.h)."""
    _fields_ = [("pts_w", _fpp), ("pts_b", _fpp),
                ("alpha_w", _fp), ("alpha_b", _fp), ("feature_w", _fp), ("feature_b", _fp),
                ("views_w", _fp), ("views_b", _fp), ("rgb_w", _fp), ("rgb_b", _fp),
                ("sem0_w", _fp), ("sem0_b", _fp), ("sem1_w", _fp), ("sem1_b", _fp),
                ("inst0_w", _fp), ("inst0_b", _fp), ("inst1_w", _fp), ("inst1_b", _fp)]


# name -> (restype, argtypes); every symbol include/pnr.h declares
SIGNATURES = {
    "pnr_version": (c_int, []),
    "pnr_last_error": (ctypes.c_char_p, []),
    "pnr_device_check": (c_int, [c_int, ctypes.c_char_p, c_int]),
    "pnr_stratified": (c_int, [c_f, c_i64, c_int, c_int, c_f, c_f, c_f]),
    "pnr_points": (c_int, [c_f, c_f, c_i64, c_int, c_f, c_f]),
    "pnr_embed": (c_int, [c_f, c_i64, c_int, c_f, c_f]),
    "pnr_mlp_packed_bytes": (c_i64, [ctypes.POINTER(MlpDesc)]),
    "pnr_mlp_pack": (c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(MlpParamsHost), ctypes.c_void_p]),
    "pnr_mlp_pack_workspace_bytes": (c_i64, [ctypes.POINTER(MlpDesc), c_int]),
    "pnr_mlp_pack_device": (c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(MlpParamsHost), c_int, c_f, c_f, c_f]),
    "pnr_mlp_repack_device": (c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(MlpParamsHost), c_int, c_f, c_f, c_f]),
    "pnr_mlp_forward": (c_int, [ctypes.POINTER(MlpDesc), c_f, c_f, c_f, c_i64, c_int, c_f, c_i64, c_i64, c_f]),
    "pnr_mlp_train_layout": (c_int, [ctypes.POINTER(MlpDesc), c_i64, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "pnr_mlp_forward_train": (c_int, [ctypes.POINTER(MlpDesc), c_f, c_f, c_f, c_i64, c_int, c_f, c_i64, c_i64, c_f, c_f]),
    "pnr_mlp_bwd_packed_bytes": (c_i64, [ctypes.POINTER(MlpDesc)]),
    "pnr_mlp_pack_bwd": (c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(MlpParamsHost), ctypes.c_void_p]),
    "pnr_mlp_backward": (c_int, [ctypes.POINTER(MlpDesc), c_f, c_f, c_f, c_f, c_i64, c_int, c_f]),
    "pnr_mlp_wgrad_workspace_bytes": (c_i64, [ctypes.POINTER(MlpDesc), c_i64]),
    "pnr_mlp_wgrad": (c_int, [ctypes.POINTER(MlpDesc), c_f, c_f, c_i64, ctypes.POINTER(MlpParamsHost), c_f, c_f]),
    "pnr_mlp_fp32_acts_floats": (c_i64, [ctypes.POINTER(MlpDesc), c_i64]),
    "pnr_mlp_backward_fp32_workspace_bytes": (c_i64, [ctypes.POINTER(MlpDesc), c_i64]),
    "pnr_mlp_forward_train_fp32": (c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(MlpParamsHost), c_f, c_f, c_i64, c_int, c_f, c_i64,
                                           c_i64, c_f, c_f]),
    "pnr_mlp_backward_fp32": (c_int, [ctypes.POINTER(MlpDesc), ctypes.POINTER(MlpParamsHost), c_f, c_i64, c_f, c_i64, c_int,
                                      ctypes.POINTER(MlpParamsHost), c_f, c_f]),
    "pnr_mlp_fused_plan": (c_int, [ctypes.POINTER(MlpDesc)]),
    "pnr_mlp_forward_composite_workspace_bytes": (c_i64, [ctypes.POINTER(MlpDesc), c_i64, c_int, c_int]),
    "pnr_mlp_forward_composite": (c_int, [ctypes.POINTER(MlpDesc), c_f, c_f, c_f, c_i64, c_int, c_f, c_f, c_int,
                                          c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "pnr_composite": (c_int, [c_f, c_i64, c_i64, c_f, c_f, c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_int,
                              c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "pnr_composite_backward": (c_int, [c_f, c_i64, c_f, c_f, c_f, c_i64, c_int, c_int, c_int,
                                       c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "pnr_composite_backward2": (c_int, [c_f, c_i64, c_f, c_f, c_f, c_i64, c_int, c_int, c_int,
                                        c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "pnr_composite_backward3": (c_int, [c_f, c_i64, c_f, c_f, c_f, c_i64, c_int, c_int, c_int, c_int,
                                        c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "pnr_losses_workspace_bytes": (c_i64, [c_i64]),
    "pnr_losses": (c_int, [ctypes.POINTER(LossCfg), c_i64, c_int, c_int] + [c_f] * 19),
    "pnr_ce3d_workspace_bytes": (c_i64, [c_i64]),
    "pnr_ce3d": (c_int, [c_f, c_i64, c_int, c_int, c_f, c_i64, c_f, c_f, c_f]),
    "pnr_gen_rays": (c_int, [ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), c_int, c_int, ctypes.c_float,
                             ctypes.c_float, c_f, c_i64, c_f, c_f]),
    "pnr_panoptic_labels": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_int, c_f, c_f, c_f, c_f]),
    "pnr_confusion": (c_int, [c_f, c_f, c_i64, c_int, c_f, c_f]),
    "pnr_sample_pdf": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_int, c_f, c_f, c_f, c_f]),
    "pnr_bbox_hits": (c_int, [c_f, c_i64, c_f, c_int, c_int, c_f, c_f, c_f, c_f]),
    "pnr_restrict_rays": (c_int, [c_f, c_i64, c_f, c_f, c_int, c_f, c_f]),
    "pnr_sample_labels": (c_int, [c_f, c_i64, c_int, c_f, c_f, c_f, c_int, c_f, c_f, c_f, c_f]),
    "pnr_ray_setup": (c_int, [c_f, c_i64, c_f, c_int, c_int, c_f, c_int, c_int, c_f, c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    "pnr_sample_pdf_labels": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_int, c_f, c_f, c_f, c_f, c_int, c_f, c_f, c_f, c_f]),
    "pnr_mlp_forward_tiles": (c_int, [ctypes.POINTER(MlpDesc), c_f, c_f, c_f, c_i64, c_int, c_f, c_f]),
    "pnr_composite_combine": (c_int, [ctypes.POINTER(MlpDesc), c_f, c_f, c_i64, c_int, c_f, c_f, c_int,
                                      c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
}

_lib = None


def load():
    """Load libpnr.so (built by __graft_entry__.build() / panopticnerf_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C panopticnerf_amd/csrc`). "
            "There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pnr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
