"""CPU check of the ping-pong MLP's hand-counted LDS waits (pnr_mlp_pp.h): compile pnr_mlp.hip to gfx950 assembly (hipcc
cross-compiles without a GPU) and run tools/check_lds_pending.py over every kernel in it -- no instruction may read or
overwrite the destination of an LDS read that the lgkmcnt waits have not covered yet."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_lds_pending as lint  # noqa: E402


def test_lint_detects_a_read_before_its_wait():
    ok = ["ds_read_b128 v[4:7], v1 offset:16", "s_waitcnt lgkmcnt(0)", "v_mfma_f32_32x32x16_bf16 v[8:23], v[4:7], v[0:3], v[8:23]"]
    assert lint.check(ok) == []
    counted = ["ds_read_b128 v[4:7], v1", "ds_read_b128 v[8:11], v1 offset:1024", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v20, v4"]
    assert lint.check(counted) == []
    bad = ["ds_read_b128 v[4:7], v1", "ds_read_b128 v[8:11], v1 offset:1024", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v20, v9"]
    assert len(lint.check(bad)) == 1 and "reads" in lint.check(bad)[0][2]
    clobber = ["ds_read_b128 v[4:7], v1", "v_mov_b32_e32 v5, v0", "s_waitcnt lgkmcnt(0)"]
    assert len(lint.check(clobber)) == 1 and "overwrites" in lint.check(clobber)[0][2]


def test_mlp_kernels_never_touch_a_pending_lds_destination(tmp_path):
    src = os.path.join(ROOT, "panopticnerf_amd", "csrc", "pnr_mlp.hip")
    out = tmp_path / "pnr_mlp.s"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.dirname(src), "-Wno-unused-function", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only",
           "-o", str(out), src]
    subprocess.check_call(cmd)
    text = open(out).read().split("\n")
    assert any("k_mlp_pp" in l for l in text) and sum(l.strip().startswith("ds_read_b128") for l in text) > 500
    flags = lint.check(text)
    assert flags == [], flags[:5]
