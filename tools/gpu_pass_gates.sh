#!/bin/bash
mkdir -p gpurun_out/r02g
O=gpurun_out/r02g
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_losses.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 | tee $O/pytest.log
for v in l8 cb8 l8 cb8; do
  echo "== $v"; PNR_LIB_PATH=build/ab/libpnr_$v.so timeout 200 python tools/train_profile.py 2>&1 | grep -E "composite" | tee -a $O/profile_${v}.log
done
