#!/bin/bash
for v in r4 r5 r8; do echo "== test $v"; PNR_LIB_PATH=build/ab/libpnr_$v.so timeout 300 python -m pytest tests/test_gpu_backward.py -x -q -m gpu -k "wgrad or mlp_backward" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -5; done
for v in r2 r4 r5 r8 r2 r4 r5 r8; do echo "== $v"; PNR_LIB_PATH=build/ab/libpnr_$v.so timeout 200 python tools/train_profile.py 2>&1 | grep -E "pnr_mlp_wgrad"; done
