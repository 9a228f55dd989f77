"""CPU tests of the drop-in boundary: libpnr.so loads without a GPU, exports every symbol
include/pnr.h declares, and rejects bad arguments with PNR_EINVAL + a message BEFORE touching
the device (no compute is launched here)."""
import ctypes
import os
import re

import pytest
import torch

from panopticnerf_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "pnr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pnr_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    names = _declared()
    assert len(names) >= 14
    assert sorted(_lib.SIGNATURES) == names


def test_library_loads_and_exports_every_symbol():
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(raw, name), f"libpnr.so does not export {name}"
    assert lib.pnr_version() == 1


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_lib.MlpDesc) == 64          # 13 named ints + clk_probe[2] + flags
    assert ctypes.sizeof(_lib.MlpParamsHost) == 18 * ctypes.sizeof(ctypes.c_void_p)
    # pnr_loss_cfg: the binding's fields are, in order and type, the header's (a silent mismatch would scramble the weights)
    import re
    hdr = open(os.path.join(ROOT, "include", "pnr.h")).read()
    body = re.search(r"typedef struct pnr_loss_cfg \{(.*?)\} pnr_loss_cfg;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            ty, names = decl.split(None, 1)
            fields += [(n.strip(), ty) for n in names.split(",")]
    ct = {"float": ctypes.c_float, "int32_t": ctypes.c_int32}
    assert [(n, ct[t]) for n, t in fields] == list(_lib.LossCfg._fields_)
    assert ctypes.sizeof(_lib.LossCfg) == 4 * len(fields)


def test_invalid_arguments_are_rejected_before_any_launch():
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)     # non-null, never dereferenced: validation fails first
    assert lib.pnr_stratified(null, 4, 8, 0, null, null, null) == -1
    assert b"null" in lib.pnr_last_error()
    assert lib.pnr_composite(one, 1, 4, one, one, null, null, null, 4, 6, 0, 0, 0, 0,
                             null, null, null, null, null, null, null, null, null) == -1
    assert b"multiple of 4" in lib.pnr_last_error()
    assert lib.pnr_sample_pdf(one, one, null, 4, 2, 8, null, null, null, null) == -1
    assert lib.pnr_sample_pdf(one, one, null, 4, 64, 1000, null, null, null, null) == -1
    assert lib.pnr_bbox_hits(one, 4, one, 3, 0, one, one, one, null) == -1
    d = ops.make_desc(W=200)
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(d)) == -1
    assert b"W=200" in lib.pnr_last_error()
    d = ops.make_desc(skip=7)
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(d)) == -1
    d = ops.make_desc(n_sem=3, head_W=100)
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(d)) == -1
    d = ops.make_desc()
    assert lib.pnr_mlp_forward(ctypes.byref(d), null, null, null, 4, 4, null, 1, 16, null) == -1
    # entry points added for training and the SURVEY 8f rows
    assert lib.pnr_mlp_wgrad(ctypes.byref(d), null, null, 64, None, null, null) == -1
    assert lib.pnr_mlp_wgrad_workspace_bytes(ctypes.byref(ops.make_desc(precision="fp32")), 64) == -1
    assert lib.pnr_mlp_wgrad_workspace_bytes(ctypes.byref(d), 786432) > 0
    assert lib.pnr_losses(None, 4, 3, 0, *([null] * 19)) == -1
    assert lib.pnr_losses_workspace_bytes(529408) >= 64 + 2068 * 32
    assert lib.pnr_ce3d(null, 4, 4, 3, null, 4, null, null, null) == -1
    assert lib.pnr_confusion(one, one, 4, 10000, one, null) == -1 and b"8192" in lib.pnr_last_error()
    assert lib.pnr_panoptic_labels(null, null, null, 4, 3, 0, null, null, null, null) == -1
    intr = (ctypes.c_float * 4)(0.0, 1.0, 0.0, 0.0)
    c2w = (ctypes.c_float * 12)()
    assert lib.pnr_gen_rays(intr, c2w, 8, 8, 0.5, 10.0, null, 64, one, null) == -1 and b"focal" in lib.pnr_last_error()
    assert lib.pnr_gen_rays(intr, c2w, 8, 8, 0.5, 10.0, null, 0, null, null) == 0            # empty input: a no-op
    # zero rays is a no-op, not an error (empty input edge case)
    assert lib.pnr_stratified(one, 0, 8, 0, null, one, null) == 0
    assert lib.pnr_embed(one, 0, 10, one, null) == 0


def test_descriptor_diagnostic_words_are_validated():
    """pnr_mlp_desc.clk_probe is a device address the forward kernels store to and flags select kernels, so a caller
    that did not zero-initialise the descriptor is rejected where that can be recognised: undefined flag bits, a misaligned
    clk_probe, ablation bits without PNR_MLP_TRACE, PNR_MLP_TRACE without a clock buffer.  The trace ablation lives in bits 4..6
    (bit 0 stays PNR_MLP_SOFTMAX)."""
    lib = _lib.load()
    ok = ops.make_desc(8, 256, 4, 10, 4, 45, 32, 128, "bf16")
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(ok)) > 0
    for flags in (2, 0x80, 0x2000000, 0x10, 0x7B00, -1):
        d = ops.make_desc(8, 256, 4, 10, 4, 45, 32, 128, "bf16")
        d.flags = flags
        assert lib.pnr_mlp_packed_bytes(ctypes.byref(d)) == -1, hex(flags)
        assert b"flags" in lib.pnr_last_error()
    for cap in (8, 64, 192, 511):          # PNR_MLP_WG_CAP(n): bits 16..24, a share of the compute units for a plan-2 launch
        d = ops.make_desc(8, 256, 4, 10, 4, 45, 32, 128, "bf16")
        d.flags = cap << 16
        assert lib.pnr_mlp_packed_bytes(ctypes.byref(d)) > 0, cap
    d = ops.make_desc(8, 256, 4, 10, 4, 45, 32, 128, "bf16")
    d.clk_probe[0] = 0x1008
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(d)) == -1 and b"clk_probe" in lib.pnr_last_error()
    d.clk_probe[0] = 0x1000
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(d)) > 0
    d.flags = _lib.MLP_SOFTMAX
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(d)) > 0
    d.flags = _lib.MLP_TRACE + (3 << 4)
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(d)) > 0
    d.clk_probe[0] = 0
    assert lib.pnr_mlp_packed_bytes(ctypes.byref(d)) == -1 and b"PNR_MLP_TRACE" in lib.pnr_last_error()


def test_ops_refuse_cpu_tensors():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.stratified(torch.zeros(4, 8), 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.composite(torch.zeros(4, 32), torch.zeros(4, 8), torch.zeros(4, 8))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libpnr.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


@pytest.mark.parametrize("S", [1, 255, 256, 1000, 786432])
def test_train_layout_is_padded_line_aligned_and_ordered(S):
    """pnr_mlp_train_layout (host arithmetic only): every region of the training buffers starts on a 128-byte line, holds
    S_pad = ceil(S / 256) * 256 rows of its width, regions do not overlap, and the acts total leaves room for the gate bits
    (one bit per element of X_1..X_D, G, SH_sem, SH_inst) behind the last bf16 region -- what include/pnr.h documents."""
    desc = ops.make_desc(8, 256, 4, 10, 4, 45, 32, 128, "bf16")
    ao, do = ops.train_layout(desc, S)
    D, W, H = 8, 256, 128
    Sp = (S + 255) // 256 * 256
    aw = [64, 32] + [W] * D + [W, H, H, H]
    assert len(ao) == D + 7 and len(do) == D + 8
    for offs, widths in ((ao, aw), (do, [H, W, H, H] + [W] * D + [32, 64, 64])):
        for i, w in enumerate(widths):
            assert offs[i] % 64 == 0, (i, offs[i])                  # bf16 elements: 64 = one 128-byte line
            assert offs[i + 1] - offs[i] >= Sp * w if i + 1 < len(widths) else offs[-1] - offs[i] >= Sp * w
    gate_bits = Sp * (D * W + 3 * H)
    assert ao[-1] - (ao[5 + D] + Sp * H) >= gate_bits // 16         # bits -> bf16 units
    assert do[-1] == do[D + 6] + Sp * 64


def test_fused_pass_eligibility_and_workspace_arithmetic():
    """Host-side rules of the fused MLP + compositing pass: ops.fused_supported mirrors what pnr_mlp_forward_composite accepts
    (bf16, logits compositing, no sigma noise, N a multiple of 32 in [32, 256], C + K <= 128), and the workspace is one record of
    1 + C + K floats (padded to 4: Q and the logit sums) per 32-sample tile of whole 256-sample groups, plus one (lw, r, g, b)
    quadruple per sample."""
    lib = _lib.load()
    d = ops.make_desc(8, 256, 4, 10, 4, 45, 32, 128, "bf16")
    assert ops.fused_supported(d, 192) and ops.fused_supported(d, 32) and ops.fused_supported(d, 256)
    assert not ops.fused_supported(d, 100) and not ops.fused_supported(d, 16) and not ops.fused_supported(d, 288)
    assert ops.fused_supported(d, 192, sem_mode=1) and not ops.fused_supported(d, 192, noise=torch.zeros(1))
    # softmax compositing: only where a head's logit blocks are in registers together (the plan-1 kernel)
    assert not ops.fused_supported(ops.make_desc(8, 256, 4, 10, 4, 19, 40, 128, "bf16"), 192, sem_mode=1)
    assert ops.fused_supported(ops.make_desc(8, 256, 4, 10, 4, 19, 40, 128, "bf16"), 192, sem_mode=0)
    assert ops.fused_supported(ops.make_desc(8, 256, 4, 10, 4, 0, 0, 128, "bf16"), 192, sem_mode=1)       # no learned field: nothing to normalise
    assert ops.fused_image(0) is True and ops.fused_image(1) == "softmax"
    # ... and the plan question is flag-aware: softmax has the two-tile kernels for heads of depth 2 and 1, plan 1 (depth 2) below them
    d2 = ops.desc_for_mode(d, 1)
    assert lib.pnr_mlp_fused_plan(ctypes.byref(d2)) == 2 and lib.pnr_mlp_fused_plan(ctypes.byref(d)) == 2
    dd1 = ops.make_desc(8, 256, 4, 10, 4, 45, 32, 128, "bf16", "trunk", 1)
    assert lib.pnr_mlp_fused_plan(ctypes.byref(dd1)) == 2 and lib.pnr_mlp_fused_plan(ctypes.byref(ops.desc_for_mode(dd1, 1))) == 2
    assert ops.fused_supported(dd1, 192) and ops.fused_supported(dd1, 192, sem_mode=1)
    dt1 = ops.make_desc(8, 256, 4, 10, 4, 45, 32, 128, "bf16", "feature", 1)          # head_tap feature + one Linear: k_mlp_tt_fd1[sm]_*
    assert lib.pnr_mlp_fused_plan(ctypes.byref(dt1)) == 2 and lib.pnr_mlp_fused_plan(ctypes.byref(ops.desc_for_mode(dt1, 1))) == 2
    d96 = ops.make_desc(8, 256, 4, 10, 4, 96, 0, 128, "bf16", "trunk", 2)              # a third semantic block, no instance head: k_mlp_tt_*_s3i0
    assert lib.pnr_mlp_fused_plan(ctypes.byref(d96)) == 2 and lib.pnr_mlp_fused_plan(ctypes.byref(ops.desc_for_mode(d96, 1))) == 2
    d96i = ops.make_desc(8, 256, 4, 10, 4, 96, 32, 128, "bf16", "trunk", 2)            # ... beside an instance head: 8 + 2 accumulators
    assert lib.pnr_mlp_fused_plan(ctypes.byref(d96i)) == 0 and lib.pnr_mlp_fused_plan(ctypes.byref(ops.desc_for_mode(d96i, 1))) == 0
    dbig = ops.make_desc(8, 256, 4, 10, 4, 100, 0, 128, "bf16", "trunk", 1)            # four semantic blocks: no fused softmax at all
    assert lib.pnr_mlp_fused_plan(ctypes.byref(ops.desc_for_mode(dbig, 1))) == 0 and not ops.fused_supported(dbig, 192, sem_mode=1)
    d4 = ops.make_desc(4, 256, 1, 10, 4, 45, 32, 128, "bf16")            # another depth: no two-tile kernel, plan 1 has softmax
    assert lib.pnr_mlp_fused_plan(ctypes.byref(ops.desc_for_mode(d4, 1))) == 1
    assert not ops.fused_supported(ops.make_desc(8, 256, 4, 10, 4, 45, 32, 128, "fp32"), 192)
    assert not ops.fused_supported(ops.make_desc(8, 256, 4, 10, 4, 100, 60, 128, "bf16"), 192)
    ws = lib.pnr_mlp_forward_composite_workspace_bytes
    rec = (1 + 77 + 3) // 4 * 4
    assert rec == 80
    for R, N in ((1, 32), (37, 192), (65536, 192), (510, 64)):
        S = R * N
        tiles = (S + 255) // 256 * 8
        assert ws(ctypes.byref(d), R, N, 0) == tiles * rec * 4 + S * 16 + 256 + 128 * (R + 1)       # + the per-ray table of k_ray_aux
        assert ws(ctypes.byref(d), R, N, 1) == ws(ctypes.byref(d), R, N, 0)
    assert ws(ctypes.byref(d), 10, 100, 0) == -1 and ws(ctypes.byref(d), 10, 16, 0) == -1
    d0 = ops.make_desc(4, 128, -1, 10, 4, 0, 0, 64, "bf16")
    assert ws(ctypes.byref(d0), 8, 32, 0) == 8 * 4 * 4 + 8 * 32 * 16 + 256 + 128 * 9          # no heads: Q alone, padded to 4 floats
