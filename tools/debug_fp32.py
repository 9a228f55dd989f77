"""Diagnostic: localise an MLP-path mismatch by enabling the network's layers one at a time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import torch_oracle as to
from panopticnerf_amd import ops

dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
D, W, C = 2, 128, 3
cfg = to.mlp_config(D=D, W=W, skips=(), n_sem=C, n_inst=0, head_W=W // 2)
full = to.init_params(cfg, seed=1)
g = torch.Generator().manual_seed(0)
rays = torch.cat([torch.randn(4, 3, generator=g), torch.randn(4, 3, generator=g) * 0.3 + torch.tensor([0, 0, 1.0]),
                  torch.full((4, 1), 0.5), torch.full((4, 1), 20.0)], 1)
z = to.stratified(rays, 16)
order = ["bias-only", "pts_linears.0", "alpha_linear", "pts_linears.1", "feature_linear", "views_linears.0",
         "rgb_linear", "semantic_linears.0", "semantic_linears.1"]
p = {k: (v.clone() if k.endswith("bias") else torch.zeros_like(v)) for k, v in full.items()}
for name in order:
    if name != "bias-only":
        p[name + ".weight"] = full[name + ".weight"].clone()
    desc = ops.make_desc(D, W, -1, 10, 4, C, 0, W // 2, prec)
    raw = ops.mlp_forward(desc, ops.pack_mlp(desc, p).to(dev), rays.to(dev), z.to(dev)).cpu().T.reshape(4, 16, -1)
    ref = to.run_network(p, cfg, rays, z, emulate_bf16=(prec == "bf16"))
    e = (raw - ref).abs()
    print(f"{prec} +{name:20s} max err rgb {e[..., :3].max():.2e} sigma {e[..., 3].max():.2e} sem {e[..., 4:].max():.2e}"
          f"   | ref scale {ref.abs().max():.2e}")
