"""The fused fine-level launch (65,536 rays x 192 samples, 45 + 32 heads) under every position of SURVEY.md 9 items 4 and 5 -- head_tap x
head_depth x semantic_activation -- as the two-tile assembly kernel (plan 2) and as the ping-pong kernel on the best image it has
(plan 1, or the classic image where plan 1 does not exist), same box, one process.
usage: python tools/tt_variants_time.py [n_sem n_inst]      (default 45 32; 96 0 = the three-block semantic head)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import benchlib, make_network, ops, synthetic

C, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (45, 32)
dev = torch.device("cuda:0")
torch.manual_seed(0)
rays = synthetic.camera_rays()[:65536].to(dev)
z = ops.stratified(rays, 192)
print("heads: %d semantic + %d instance logits" % (C, K))
print("%-8s %-6s %-8s  %-22s %-22s" % ("head_tap", "depth", "compos.", "k_mlp_tt (plan 2)", "k_mlp_pp (its best plan)"))
for tap in ("trunk", "feature"):
    for depth in (2, 1):
        net = make_network(NS(N_importance=128, num_classes=C, num_instances=K, head_depth=depth, head_tap=tap)).eval()
        synthetic.trained_like_(net)
        net = net.to(dev)
        for sem_mode in (0, 1):
            row = []
            for cap in (2, 1):
                base = net.nerf_1.desc("bf16")
                if cap == 2:
                    desc, img = net.packed(1, dev, fused="softmax" if sem_mode else True)
                else:
                    os.environ["PNR_FUSED_PLAN"] = "1"
                    net.invalidate_packed()
                    desc, img = net.packed(1, dev, fused="softmax" if sem_mode else True)
                    del os.environ["PNR_FUSED_PLAN"]
                    net.invalidate_packed()
                desc = ops.desc_for_mode(desc, sem_mode)
                if sem_mode and desc.plan == 0:
                    row.append("two-kernel path only")
                    continue
                benchlib.time_mlp_forward_tiles(desc, img, rays, z, 2)
                ms, mhz = min(benchlib.time_mlp_forward_tiles(desc, img, rays, z, 5) for _ in range(3))
                row.append("%.3f ms @%4.0f (plan %d)" % (ms, mhz, desc.plan))
            print("%-8s %-6d %-8s  %-22s %-22s" % (tap, depth, "softmax" if sem_mode else "logits", row[0], row[1]), flush=True)
