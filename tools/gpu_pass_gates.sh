#!/bin/bash
PNR_LIB_PATH=build/ab/libpnr_g3.so timeout 600 python -m pytest tests/test_gpu_backward.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | head
for v in hd g3 g3s16 hd g3 g3s16; do echo "== $v"; PNR_LIB_PATH=build/ab/libpnr_$v.so timeout 200 python tools/train_profile.py 2>&1 | grep -E "mlp_backward|pnr_mlp_wgrad"; done
