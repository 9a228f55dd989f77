"""CPU tests of the host packer (pnr_mlp_pack): the packed MFMA-fragment image, pushed
through a numpy emulation of the kernel's dataflow (tests/_emulate.py), must reproduce the
dense oracle MLP.  Covers BASELINE configs' geometries: 8x256 skip@4 with heads (configs 2-5)
and 4x128 without skip (config 1)."""
import ctypes
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

from _emulate import PackedImage, emulate
from oracle import torch_oracle as to
from panopticnerf_amd import _lib, make_network, ops


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
@pytest.mark.parametrize("geom", [(8, 256, [4], 45, 32), (8, 256, [4], 0, 0), (4, 128, [], 0, 0),
                                  (3, 128, [1], 7, 0), (2, 128, [0], 0, 33)])
def test_packed_image_reproduces_dense_mlp(prec, geom):
    D, W, skips, C, K = geom
    torch.manual_seed(D * 1000 + W + C)
    net = make_network(NS(D=D, W=W, skips=skips, num_classes=C, num_instances=K))
    sd = net.nerf_0.state_dict()
    desc = net.nerf_0.desc(prec)
    img = ops.pack_mlp(desc, sd)
    n = 37
    pts = torch.rand(n, 3) * 40 - 20
    vd = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    ocfg = to.mlp_config(D=D, W=W, skips=tuple(skips), n_sem=C, n_inst=K, head_W=W // 2)
    ref = to.mlp_forward({k: v.detach() for k, v in sd.items()}, ocfg, pts, vd, emulate_bf16=(prec == "bf16")).numpy()
    em = emulate(img, pts.numpy(), vd.numpy())
    # bf16: an activation that lands on a rounding tie may flip (accumulation order) -> 1 bf16 ulp downstream
    np.testing.assert_allclose(em, ref, atol=2e-3 if prec == "bf16" else 2e-6)


def test_image_header_and_sizes():
    desc = ops.make_desc(n_sem=45, n_inst=32)
    net = make_network(NS(num_classes=45, num_instances=32))
    img = ops.pack_mlp(desc, net.nerf_0.state_dict())
    im = PackedImage(img)
    assert img.numel() == _lib.load().pnr_mlp_packed_bytes(ctypes.byref(desc))
    # bf16 chunks hold 4 (layer 0) / 2 (hidden) row blocks: trunk 2 + 7*4, feature 4, views 2, rgb/sigma 1, sem 2+2, inst 2+1
    assert im.n_chunks == 30 + 4 + 2 + 1 + 4 + 3
    assert im.max_frags == 41      # skip-layer chunk: 2 blocks x (4 gamma(x) + 16 h) k-steps + bias
    assert int(im.table[:, 1].sum()) * 1024 + im.data_off == img.numel()
    # fp32 image: twice the k-steps
    d32 = ops.make_desc(n_sem=45, n_inst=32, precision="fp32")
    im32 = PackedImage(ops.pack_mlp(d32, net.nerf_0.state_dict()))
    assert im32.n_chunks == 64 + 8 + 4 + 1 + 6 + 5 and im32.max_frags == 49     # fp32: one block per chunk


def test_pack_rejects_missing_parameters():
    desc = ops.make_desc(n_sem=5)
    net = make_network(NS(num_classes=0))
    with pytest.raises(RuntimeError, match="semantic"):
        ops.pack_mlp(desc, net.nerf_0.state_dict())


def test_two_tile_plan_image_matches_the_assembly_generator():
    """Plan 2 (the image k_mlp_tt consumes, csrc/asm/gen_mlp_tt.py): the generator carries its own copy of the chunk plan -- the chunk
    table of the packed image must be exactly what the generated kernels assume (offsets, sizes), no chunk above 33 fragments
    (four 33 KiB weight slots), a chunk count that is a multiple of 4 (slot = chunk % 4 is static); the geometry switch reports
    plan 2 only where a kernel exists, and a plan-2 image renders the same dense network (numpy emulation)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_mlp_tt", os.path.join(root, "panopticnerf_amd", "csrc", "asm", "gen_mlp_tt.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    for C, K, depth, tap in ((45, 32, 2, "trunk"), (19, 8, 2, "trunk"), (45, 0, 2, "trunk"), (19, 0, 2, "trunk"), (0, 0, 2, "trunk"),
                             (45, 32, 1, "trunk"), (19, 8, 1, "trunk"), (64, 0, 1, "trunk"), (19, 0, 1, "trunk"),
                             (45, 32, 2, "feature"), (19, 0, 2, "feature"), (45, 32, 1, "feature"), (64, 0, 1, "feature"),
                             (96, 0, 2, "trunk"), (70, 0, 1, "trunk"), (96, 0, 2, "feature"), (65, 0, 1, "feature")):     # a third semantic block
        net = make_network(NS(num_classes=C, num_instances=K, head_depth=depth, head_tap=tap))
        desc = net.nerf_0.desc("bf16")
        assert (desc.head_depth == 1) == (depth == 1)
        # head_depth 1 (round 6): the two-tile kernels k_mlp_tt_d1_*; it has no plan 1 (the ping-pong kernel's merged logit chunk)
        assert ops.fused_plan(desc, None) == 2 and ops.fused_plan(desc, 1) == (1 if (0 < C <= 64 and depth == 2) else 0)
        desc.plan = 2
        img = ops.pack_mlp(desc, net.nerf_0.state_dict())
        im = PackedImage(img)
        g = G.Gen((C + 31) // 32, (K + 31) // 32, "x", depth=depth, tap=int(tap == "feature"))
        # the kernel pads its group to a multiple of four chunks with DUMMY chunks (a piece of fragment 0, no unit): not in the image
        real = [c for c in g.chunks if not c.get("dummy")]
        assert all(c.get("dummy") for c in g.chunks[len(real):]) and all((c["off"], c["nfrag"]) == (0, 1) for c in g.chunks[len(real):])
        assert im.n_chunks == len(real) and g.NC % 4 == 0 and g.NC - len(real) < 4 and im.max_frags <= 33
        assert [(int(o), int(n)) for o, n in im.table] == [(c["off"], c["nfrag"]) for c in real]
    # an instance head alone / other depths / too many logits: no two-tile kernel -> plan 1 or 0
    assert ops.fused_plan(ops.make_desc(n_sem=0, n_inst=32), None) == 0
    assert ops.fused_plan(ops.make_desc(D=4, skip=1, n_sem=45, n_inst=32), None) == 1
    assert ops.fused_plan(ops.make_desc(n_sem=100, n_inst=32), None) == 0
    assert ops.fused_plan(ops.make_desc(n_sem=96, n_inst=32), None) == 0 and ops.fused_plan(ops.make_desc(n_sem=97, n_inst=0), None) == 0
