#!/bin/bash
# GPU pass: parity tests, smoke, bench, rocprofv3 kernel trace + PMC passes.  Logs -> gpurun_out/
# Every step has its own timeout: a hung kernel must not eat the GPU budget.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 300 python bench.py > gpurun_out/bench.log 2>&1; tail -2 gpurun_out/bench.log
if [ "$1" != "noprof" ]; then
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --train-steps 0 > $R/gpurun_out/prof_trace.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $R/gpurun_out/prof_fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $R/gpurun_out/prof_write.log 2>&1
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/prof_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $R/gpurun_out/prof_sq.log 2>&1
cd $R
du -sh gpurun_out
fi
