#!/bin/bash
# Same-box A/B per BASELINE config: the two-tile kernel (as built) against the ping-pong kernel (PNR_FUSED_PLAN=1 caps the plan), whole frames.
T=${1:-r05x}; mkdir -p gpurun_out/$T; O=gpurun_out/$T
for n in 2 3 4 5; do for l in tt pp tt pp; do
  unset PNR_FUSED_PLAN; [ $l = pp ] && export PNR_FUSED_PLAN=1
  timeout 200 python bench.py --config $n --steps 5 --warmup 2 --cpu-seconds 0 --train-steps 0 > $O/b.json 2> $O/b.err
  python -c "
import json
d=json.load(open('$O/b.json')); r=d['roofline']
print('config $n %-3s %8.2f Msamples/s  %8.3f ms/frame  frac %.4f  launch %.4f ms at %4.0f MHz  %s' % ('$l', d['value'], d['ms_per_step'], r['frac'], r['ms_per_launch'], r['shader_mhz_during_kernel'], r['kernel'][:44]))" | tee -a $O/configs_ab.txt
done; done
