#!/bin/bash
# Round-2 pass F: round-2 head (ping-pong MLP, cheaper raw-store addressing, k_composite2 for N<=64) -- full GPU tests, bench, rocprofv3 kernel trace + PMC passes
mkdir -p gpurun_out/r02f
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r02f
timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=6 -s 2>&1 | grep -v "rel L2 errors" | tail -40 > $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 4500 $O/bench.json; tail -3 $O/bench.err
PNR_MLP_VARIANT=0 timeout 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 > $O/bench_lockstep.json 2> $O/bench_lockstep.err; tail -c 1800 $O/bench_lockstep.json
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_trace -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --train-steps 0 > $O/prof_trace.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_write.log 2>&1
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/prof_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_sq.log 2>&1
cd $R
du -sh gpurun_out/r02f
