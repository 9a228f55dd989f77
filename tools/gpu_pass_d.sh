#!/bin/bash
mkdir -p gpurun_out/r02d
O=gpurun_out/r02d
python tools/pp_train_check3.py d4 2>&1 | grep -v amdgpu.ids
python tools/pp_train_check2.py 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/mlp_variants.py d4:0 d4:1 d4:0 d4:1 2>&1 | grep -v amdgpu.ids | tee $O/variants.log
