#!/bin/bash
# training step: rocprofv3 kernel trace + HBM traffic (FETCH_SIZE / WRITE_SIZE) of the same script
mkdir -p gpurun_out/r02t
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r02t
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $R/tools/train_trace.py > $O/trace.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o t -- python $R/tools/train_trace.py > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o t -- python $R/tools/train_trace.py > $O/write.log 2>&1
cd $R
ls $O/*
