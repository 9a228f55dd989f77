"""Where a frame's wall time goes: one full 1408x376 frame of configs[4] through Renderer.render, 3 times, under
`rocprofv3 --kernel-trace`; `--summary <db>` prints, for the last frame, span, sum of kernel time, idle time between kernels
and the largest gaps with the kernels either side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--summary":
    import sqlite3
    cur = sqlite3.connect(sys.argv[2]).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    cuts = [i for i in range(len(rows) - 1) if rows[i + 1][1] - rows[i][2] > 20e6]      # the 50 ms sleeps between the frames
    last = rows[cuts[-1] + 1:] if cuts else rows
    span = last[-1][2] - last[0][1]
    busy = sum(e - s for _, s, e in last)
    print(f"last frame: {len(last)} kernels, span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
    g = sorted(((last[i + 1][1] - last[i][2], last[i][0][:40], last[i + 1][0][:40]) for i in range(len(last) - 1)), reverse=True)
    import collections
    by = collections.defaultdict(lambda: [0, 0.0])
    for d, a, b in g:
        by[(a.split("(")[0][:28], b.split("(")[0][:28])][0] += 1
        by[(a.split("(")[0][:28], b.split("(")[0][:28])][1] += d
    for (a, b), (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {a:28s} -> {b:28s} n={n:3d} total {t / 1e3:8.1f} us  avg {t / n / 1e3:6.1f}")
    sys.exit(0)
import torch
from types import SimpleNamespace as NS
from panopticnerf_amd import make_network, make_renderer, synthetic
dev = torch.device("cuda:0")
cfg = NS(N_samples=64, N_importance=128, num_classes=45, num_instances=32, keep_weights=False)
net = make_network(cfg).eval()
synthetic.trained_like_(net)
net = net.to(dev)
rend = make_renderer(cfg, net)
rays = synthetic.camera_rays().to(dev)
box, ids = (t.to(dev) for t in synthetic.random_boxes(64, 45, 32))
with torch.no_grad():
    for _ in range(3):
        out = rend.render({"rays": rays[None], "bbox": box, "bbox_ids": ids})
        torch.cuda.synchronize()
        import time; time.sleep(0.05)
