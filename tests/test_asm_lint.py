"""CPU check of the hand-counted LDS waits of the ping-pong MLP (pnr_mlp_pp.h, pnr_mlp_fuse.h) and of k_wgrad's software pipeline
(pnr_mlp_wgrad.hip): compile the sources to gfx950 assembly (hipcc cross-compiles without a GPU) and run
tools/check_lds_pending.py over every kernel -- no instruction may read or overwrite the destination of an LDS read that the
lgkmcnt waits have not covered yet, on ANY path through the kernel's branches (check_cfg follows the control flow)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_lds_pending as lint  # noqa: E402
import asm_lint  # noqa: E402


def test_lint_detects_a_read_before_its_wait():
    ok = ["ds_read_b128 v[4:7], v1 offset:16", "s_waitcnt lgkmcnt(0)", "v_mfma_f32_32x32x16_bf16 v[8:23], v[4:7], v[0:3], v[8:23]"]
    assert lint.check(ok) == []
    counted = ["ds_read_b128 v[4:7], v1", "ds_read_b128 v[8:11], v1 offset:1024", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v20, v4"]
    assert lint.check(counted) == []
    bad = ["ds_read_b128 v[4:7], v1", "ds_read_b128 v[8:11], v1 offset:1024", "s_waitcnt lgkmcnt(1)", "v_mov_b32_e32 v20, v9"]
    assert len(lint.check(bad)) == 1 and "reads" in lint.check(bad)[0][2]
    clobber = ["ds_read_b128 v[4:7], v1", "v_mov_b32_e32 v5, v0", "s_waitcnt lgkmcnt(0)"]
    assert len(lint.check(clobber)) == 1 and "overwrites" in lint.check(clobber)[0][2]


def test_cfg_walk_sees_hazards_across_branches_and_ignores_other_paths():
    # the read at the loop bottom is consumed at the loop top without a wait: invisible in program order
    loop = ["k:", "s_waitcnt lgkmcnt(0)", ".L1:", "v_mov_b32_e32 v20, v4", "ds_read_b128 v[4:7], v1", "s_cbranch_scc1 .L1",
            "s_waitcnt lgkmcnt(0)", "s_endpgm"]
    assert lint.check(loop) == [] and any("reads" in f[2] for f in lint.check_cfg(loop))
    fine = ["k:", ".L1:", "ds_read_b128 v[4:7], v1", "s_waitcnt lgkmcnt(0)", "v_mov_b32_e32 v20, v4", "s_cbranch_scc1 .L1", "s_endpgm"]
    assert lint.check_cfg(fine) == []
    # a read on a path that branches away must not count against a block laid out behind it
    other = ["k:", "s_cbranch_scc1 .LB", "ds_read_b128 v[4:7], v1", "s_branch .LC", ".LB:", "v_mov_b32_e32 v5, v0", "s_endpgm",
             ".LC:", "s_waitcnt lgkmcnt(0)", "s_endpgm"]
    assert len(lint.check(other)) == 1 and lint.check_cfg(other) == []
    # a loop that only issues reads terminates (the pending list is capped where lgkmcnt saturates)
    spin = ["k:", ".L1:", "ds_read_b128 v[4:7], v1", "s_cbranch_scc1 .L1", "s_waitcnt lgkmcnt(0)", "s_endpgm"]
    assert all("overwrites" in f[2] for f in lint.check_cfg(spin))


def _hipcc():
    """hipcc of the ROCm install that builds libpnr.so; the lint is skipped (not failed) on hosts without one."""
    exe = shutil.which("hipcc") or os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    if not os.path.exists(exe):
        pytest.skip("hipcc not found: the LDS-wait lint needs the ROCm compiler")
    return exe


def _asm(tmp_path, name):
    src = os.path.join(ROOT, "panopticnerf_amd", "csrc", name)
    out = tmp_path / (name + ".s")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.dirname(src), "-Wno-unused-function", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only",
           "-o", str(out), src]
    subprocess.check_call(cmd)
    return open(out).read().split("\n")


def test_wgrad_kernel_never_touches_a_pending_lds_destination(tmp_path):
    text = _asm(tmp_path, "pnr_mlp_wgrad.hip")
    assert sum(l.strip().startswith("ds_read_b64_tr_b16") for l in text) > 200 and any("s_waitcnt lgkmcnt(12)" in l for l in text)
    flags = lint.check_cfg(text)
    assert flags == [], flags[:5]


def test_mlp_kernels_never_touch_a_pending_lds_destination(tmp_path):
    src = os.path.join(ROOT, "panopticnerf_amd", "csrc", "pnr_mlp.hip")
    out = tmp_path / "pnr_mlp.s"
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.dirname(src), "-Wno-unused-function", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only",
           "-o", str(out), src]
    subprocess.check_call(cmd)
    text = open(out).read().split("\n")
    assert any("k_mlp_pp" in l for l in text) and sum(l.strip().startswith("ds_read_b128") for l in text) > 500
    flags = lint.check(text) + lint.check_cfg(text)
    assert flags == [], flags[:5]
    # k_mlp_pp requests the next sample group's inputs behind the refill pieces of an L phase and lets the following
    # m_done() wait with `s_waitcnt vmcnt(N)`, N = the number of vector-memory instructions of that request (they complete in
    # order, so "all but the youngest N" = every LDS-DMA piece).  N is a constant in the source; the compiler decides how many
    # instructions the request becomes.  The source brackets it with markers: between them there must be exactly N loads and no
    # LDS-DMA piece -- otherwise a piece could still be in flight when its chunk is read.  (tools/asm_lint.py: the SAME check
    # fails `make all` on the assembly of the compile that produced the object.)
    assert sum("PNR_FETCH_BEGIN" in l for l in text) >= 6, "every k_mlp_pp instantiation carries the marker"
    assert asm_lint.fetch_markers(text) == []
    # and the checker does fire: one load fewer than the wait assumes
    b = next(i for i, l in enumerate(text) if "PNR_FETCH_BEGIN" in l)
    e = next(i for i in range(b + 1, len(text)) if "PNR_FETCH_END" in text[i])
    k = next(i for i in range(b + 1, e) if text[i].strip().startswith("global_load_dword"))
    assert any("the wait assumes" in m for m in asm_lint.fetch_markers(text[:k] + text[k + 1:]))


DMA_USERS = ("pnr_mlp.hip", "pnr_mlp_bwd.hip", "pnr_mlp_wgrad.hip")
ALL_HIP = sorted(f for f in os.listdir(os.path.join(ROOT, "panopticnerf_amd", "csrc")) if f.endswith(".hip"))


@pytest.mark.parametrize("name", ALL_HIP)
def test_the_dma_helper_is_the_only_user_of_m0(tmp_path, name):
    # pnr_dma_piece (pnr_common.h) writes M0 inside inline asm (M0 is reserved in the AMDGPU backend: a clobber is warned about
    # and ignored).  Safe only while the compiler itself never keeps a value in M0 in ANY
    # object of the library (LDS-DMA builtins, v_movrel register indexing, s_sendmsg, ...): every access must be the helper's
    # own s_mov, each directly followed by its LDS-DMA load in the scalar-base form.  Every .hip of the library is checked.
    text = _asm(tmp_path, name)
    mine, other = asm_lint.m0_accesses(text)
    assert other == [], other[:5]
    assert asm_lint.m0_rule(text) == []
    if name in DMA_USERS:
        assert len(mine) >= 20
    else:
        assert mine == []


def test_build_time_lint_runs_on_every_object():
    """`make all` lints the assembly each .hip object was built from (panopticnerf_amd/csrc/Makefile: -save-temps=obj +
    tools/asm_lint.py, a finding deletes the object and fails the build)."""
    mk = open(os.path.join(ROOT, "panopticnerf_amd", "csrc", "Makefile")).read()
    assert "-save-temps=obj" in mk and "tools/asm_lint.py" in mk and "rm -f $@; exit 1" in mk
    saved = [os.path.join(ROOT, "build", "obj", f[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s") for f in ALL_HIP]
    have = [p for p in saved if os.path.exists(p)]
    if not have:
        pytest.skip("no saved assembly (the library was not built on this host)")
    assert asm_lint.main(have) == 0
