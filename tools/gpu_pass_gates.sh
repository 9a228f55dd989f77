#!/bin/bash
# wgrad rewrite A/B: correctness of the training path on the head build, then per-stage times (same box)
mkdir -p gpurun_out/r02g
O=gpurun_out/r02g
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_losses.py tests/test_gpu_convergence.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 | tee $O/pytest.log
for v in gates wg2 wg2s gates wg2 wg2s; do
  echo "== $v"; PNR_LIB_PATH=build/ab/libpnr_$v.so timeout 200 python tools/train_profile.py 2>&1 | grep -E "mlp_forward_train|mlp_backward|pnr_mlp_wgrad|max rel" | tee -a $O/profile_$v.log
done
