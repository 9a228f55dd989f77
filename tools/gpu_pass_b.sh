#!/bin/bash
# Round-2 pass B: ping-pong MLP kernel A/B (same box, one process per variant) + its per-chunk timeline
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
O=gpurun_out/r02b
timeout 900 python tools/mlp_variants.py c4:0 c4:1 c4b:1 c4u:1 c6:1 c4p1:1 c4:0 c4:1 c4b:1 c4u:1 c6:1 c4p1:1 > $O/variants.log 2>&1
cat $O/variants.log | grep -v amdgpu.ids


