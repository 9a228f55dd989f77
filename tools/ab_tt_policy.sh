#!/bin/bash
# Same-box A/B of the weight pieces' cache policy (round 5): whole frames + the roofline launch, interleaved, two repeats.
#   default  libpnr.so as built: k_mlp_tt with the default policy          ttnt   k_mlp_tt with nt (build/ab/libpnr_ttnt.so, tools/build_tt_variant.sh)
#   pp / ppdef  PNR_FUSED_PLAN=1: k_mlp_pp with nt (as built) / with the default policy (build/ab/libpnr_ppdef.so, -DPNR_PP_DMA_AUX=0)
T=${1:-r05q}; mkdir -p gpurun_out/$T; O=gpurun_out/$T
for rep in 1 2; do for l in default ttnt pp ppdef; do
  unset PNR_LIB_PATH PNR_FUSED_PLAN
  case $l in ttnt) export PNR_LIB_PATH=$PWD/build/ab/libpnr_ttnt.so ;; pp) export PNR_FUSED_PLAN=1 ;; ppdef) export PNR_FUSED_PLAN=1 PNR_LIB_PATH=$PWD/build/ab/libpnr_ppdef.so ;; esac
  timeout 200 python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --train-steps 0 > $O/bench_${l}_$rep.json 2> $O/err_${l}_$rep.log
  python -c "
import json
d=json.load(open('$O/bench_${l}_$rep.json')); r=d['roofline']
print('%-8s run $rep  %8.2f Msamples/s  %8.3f ms/frame  frac %.4f  launch %.4f ms at %4.0f MHz  %s' % ('$l', d['value'], d['ms_per_step'], r['frac'], r['ms_per_launch'], r['shader_mhz_during_kernel'], r['kernel'][:40]))" | tee -a $O/ab.txt
done; done
