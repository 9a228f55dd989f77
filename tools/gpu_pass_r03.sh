#!/bin/bash
# Round-3 pass: full GPU tests, smoke, bench (config 5 + the other BASELINE configs), rocprofv3 kernel trace + PMC passes of the
# bench, kernel trace + HBM traffic of the training step.  The rocpd databases are summarised on the box and deleted
# (gpurun_out/ is capped at 64 MiB).   usage: tools/gpu_pass_r03.sh <tag>
T=${1:-r03d}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/$T
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=6 -s 2>&1 | grep -v "rel L2 errors" | tail -40 > $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
for n in 1 2 3 4; do timeout 200 python bench.py --config $n --steps 5 --warmup 2 --cpu-seconds 0 > $O/bench_config$n.json 2>> $O/bench.err; done
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_trace -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --train-steps 0 > $O/prof_trace.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_write.log 2>&1
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/prof_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_sq.log 2>&1
timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $O/prof_l2 -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_l2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/train_trace -o t -- python $R/tools/train_trace.py 4 > $O/train_trace.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/train_fetch -o t -- python $R/tools/train_trace.py 3 > $O/train_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/train_write -o t -- python $R/tools/train_trace.py 3 > $O/train_write.log 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/prof_summary.py $(db prof_trace) $(db prof_fetch) $(db prof_write) $(db prof_sq) $(db prof_l2) > $O/rocprof_summary.txt 2> $O/summary.err
python tools/update_traffic.py $(db prof_fetch) $(db prof_write) ${T}_rocprof_summary.txt > $O/traffic.log 2>&1; cp profiles/latest_traffic.json $O/latest_traffic.json
python tools/train_summary.py $(db train_trace) 4 $(db train_fetch) $(db train_write) 3 > $O/train_summary.txt 2>> $O/summary.err
rm -rf $O/prof_l2 $O/prof_trace $O/prof_fetch $O/prof_write $O/prof_sq $O/train_trace $O/train_fetch $O/train_write
tail -12 $O/train_summary.txt; cat $O/summary.err | tail -5; grep -n "fused MLP fine-level" $O/rocprof_summary.txt
du -sh gpurun_out/$T
