#!/bin/bash
# ping-pong MLP kernel A/B (same box, one process per variant); usage: gpu_pass_b.sh "<lib:variant ...>" [trace lib]
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
O=gpurun_out/r02b
timeout 900 python tools/mlp_variants.py $1 > $O/variants.log 2>&1
cat $O/variants.log | grep -v amdgpu.ids
if [ -n "$2" ]; then PNR_MLP_VARIANT=1 timeout 200 python tools/mlp_trace_pp.py $2 2>&1 | cut -c1-200 | grep -v "refill0=" > $O/trace_pp.log; PNR_MLP_VARIANT=1 timeout 200 python tools/mlp_trace_pp.py $2 2>&1 | grep "wave [04]" | awk '{print $1,$2,$5,$6,$7,$8,$9,$10}' | head -24; fi
