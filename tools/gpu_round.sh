#!/bin/bash
# GPU pass: parity tests, smoke, variants A/B, bench, rocprof kernel trace + PMC.  Logs -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -8 gpurun_out/smoke.log
timeout 600 python tools/mlp_variants.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/variants.log
