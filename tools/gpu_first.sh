#!/bin/bash
# First GPU pass: parity tests (all, no -x), smoke, short bench.  Logs -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/dev.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -20 gpurun_out/smoke.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1; tail -5 gpurun_out/bench.log
