#!/bin/bash
for v in h4 h8; do echo "== test $v"; PNR_LIB_PATH=build/ab/libpnr_$v.so timeout 300 python -m pytest tests/test_gpu_stages.py -x -q -m gpu -k "pingpong or fused_mlp" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -5; done
python tools/mlp_variants.py h0:1 h2:1 h4:1 h8:1 h0:1 h2:1 h4:1 h8:1 2>&1 | grep -v amdgpu.ids
