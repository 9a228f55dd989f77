#!/bin/bash
# Round-2 pass J: round-2 head (fused MLP + compositing inference pass, ping-pong MLP, saved-tensor layout + gate bits + per-shape k_wgrad on the training path) -- full GPU
# tests, bench, rocprofv3 kernel trace + PMC passes of the bench, kernel trace + HBM traffic of the training step.  The rocpd
# databases are summarised on the box and deleted (gpurun_out/ is capped at 64 MiB).
mkdir -p gpurun_out/r02j
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/r02j
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=6 -s 2>&1 | grep -v "rel L2 errors" | tail -40 > $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -3 $O/bench.err
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_trace -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --train-steps 0 > $O/prof_trace.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_write.log 2>&1
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/prof_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_sq.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/train_trace -o t -- python $R/tools/train_trace.py 4 > $O/train_trace.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/train_fetch -o t -- python $R/tools/train_trace.py 3 > $O/train_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/train_write -o t -- python $R/tools/train_trace.py 3 > $O/train_write.log 2>&1
cd $R
db() { find $O/$1 -name "*.db" | head -1; }
python tools/prof_summary.py $(db prof_trace) $(db prof_fetch) $(db prof_write) $(db prof_sq) > $O/rocprof_summary.txt 2> $O/summary.err
python tools/update_traffic.py $(db prof_fetch) $(db prof_write) r02j_rocprof_summary.txt > $O/traffic.log 2>&1; cp profiles/latest_traffic.json $O/latest_traffic.json
python tools/train_summary.py $(db train_trace) 4 $(db train_fetch) $(db train_write) 3 > $O/train_summary.txt 2>> $O/summary.err
rm -rf $O/prof_trace $O/prof_fetch $O/prof_write $O/prof_sq $O/train_trace $O/train_fetch $O/train_write
tail -20 $O/train_summary.txt; cat $O/summary.err | tail -5
du -sh gpurun_out/r02j
