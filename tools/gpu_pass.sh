#!/bin/bash
# The gpurun command sets of this repo in ONE script (round 4: the nine gpu_pass_*.sh / gpu_round.sh of rounds 1-3 folded).
# Every step runs under its own `timeout`; rocpd databases are summarised on the box and deleted (gpurun_out/ is capped at 64 MiB).
#   usage: tools/gpu_pass.sh <mode> <tag> [args]
#   tests   GPU tests (+ smoke): -m gpu, optional pytest -k expression as $3
#   bench   bench.py (config 5) [+ configs 1-4 with $3 = all]
#   full    tests + smoke + bench for every BASELINE config + rocprofv3 kernel trace and FETCH / WRITE / SQ / L2 PMC passes of the
#           bench + kernel trace and HBM traffic of the training step  (the round's final pass; ~20 GPU-minutes)
#   ab      same-box A/B of builds under build/ab: $3 = composite|fused|wgrad, $4... = lib names
#   fidelity  tools/train_fidelity.py --hip (the HIP students, bf16 and fp32 parity mode) at $3 = unit|image weights
MODE=${1:-tests}
T=${2:-r04}
mkdir -p gpurun_out/$T
export TMPDIR=/tmp
R=$(pwd)
O=$R/gpurun_out/$T

run_tests() {
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 -s ${1:+-k "$1"} 2>&1 | grep -v "rel L2 errors" | tail -150 > $O/pytest_gpu.log
  tail -15 $O/pytest_gpu.log
}
run_smoke() { timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log; }
run_bench() {
  timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json; tail -3 $O/bench.err
  if [ "$1" = "all" ]; then
    for n in 1 2 3 4; do timeout 200 python bench.py --config $n --steps 5 --warmup 2 --cpu-seconds 0 > $O/bench_config$n.json 2>> $O/bench.err; done
  fi
}
db() { find $O/$1 -name "*.db" | head -1; }
run_prof() {
  cd /tmp
  # the rocprofv3 passes trace the SERIAL frame (PNR_OVERLAP=0): every k_mlp_tt dispatch is then a whole-device launch, which is what
  # bench.py's roofline launch (hipEvents around standalone launches) is compared with; the timed frames of bench.py itself run the
  # fine level of a chunk beside the coarse level of the next on 192 + 64 workgroups (profiles/r06/r06r)
  export PNR_OVERLAP=0
  timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof_trace -o bench -- python $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --train-steps 0 > $O/prof_trace.log 2>&1
  timeout 240 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_fetch.log 2>&1
  timeout 240 rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_write.log 2>&1
  timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -d $O/prof_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_sq.log 2>&1
  timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum -d $O/prof_l2 -o bench -- python $R/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/prof_l2.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --stats -d $O/train_trace -o t -- python $R/tools/train_trace.py 4 > $O/train_trace.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/train_fetch -o t -- python $R/tools/train_trace.py 3 > $O/train_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/train_write -o t -- python $R/tools/train_trace.py 3 > $O/train_write.log 2>&1
  unset PNR_OVERLAP
  cd $R
  python tools/prof_summary.py $(db prof_trace) $(db prof_fetch) $(db prof_write) $(db prof_sq) $(db prof_l2) > $O/rocprof_summary.txt 2> $O/summary.err
  python tools/update_traffic.py $(db prof_fetch) $(db prof_write) ${T:0:3}/${T}_rocprof_summary.txt $(db prof_trace) > $O/traffic.log 2>&1; cp profiles/latest_traffic.json $O/latest_traffic.json
  python tools/train_summary.py $(db train_trace) 4 $(db train_fetch) $(db train_write) 3 > $O/train_summary.txt 2>> $O/summary.err
  rm -rf $O/prof_l2 $O/prof_trace $O/prof_fetch $O/prof_write $O/prof_sq $O/train_trace $O/train_fetch $O/train_write
  tail -12 $O/train_summary.txt; tail -5 $O/summary.err; grep -n "fused MLP fine-level" $O/rocprof_summary.txt
}

case $MODE in
  tests) run_tests "$3"; run_smoke ;;
  bench) run_bench "$3" ;;
  full) run_tests; run_smoke; run_bench all; run_prof ;;
  fidelity) timeout 900 python tools/train_fidelity.py --hip --weights ${3:-unit} --steps ${4:-100} --students "" --out $O/fidelity_${3:-unit}.json > $O/fidelity_${3:-unit}.log 2>&1; tail -6 $O/fidelity_${3:-unit}.log ;;
  ab)
    kind=$3; shift 3
    case $kind in
      composite) for N in 192 64; do timeout 200 python tools/composite_ab.py panopticnerf_amd/libpnr.so build/ab/libpnr_$1.so $N 2>&1 | tail -5 | tee -a $O/ab_composite.log; done ;;
      fused) timeout 400 python tools/fused_ab.py default "$@" 2>&1 | tail -12 | tee $O/ab_fused.log ;;
      wgrad) for l in "$@"; do PNR_LIB_PATH=build/ab/libpnr_$l.so timeout 120 python tools/wgrad_time.py 2>&1 | tail -3 | tee -a $O/ab_wgrad.log; done ;;
    esac ;;
esac
du -sh gpurun_out/$T
