#!/bin/bash
mkdir -p gpurun_out/r02e
O=gpurun_out/r02e
timeout 300 python -m pytest tests/test_gpu_stages.py -m gpu -q -x -k "composite" -p no:cacheprovider 2>&1 | tail -5
PNR_CMP_VARIANT=2 timeout 300 python -m pytest tests/test_gpu_stages.py -m gpu -q -x -k "composite" -p no:cacheprovider 2>&1 | tail -5
for N in 192 64; do
  for v in cmp2 cmp2u8 cmp2u2; do
    timeout 200 python tools/composite_ab.py build/ab/libpnr_cmp0.so build/ab/libpnr_$v.so $N 2>&1 | grep -v amdgpu.ids | tee -a $O/composite_ab.log
  done
done
