#!/bin/bash
mkdir -p gpurun_out/r02g
O=gpurun_out/r02g
for v in l8 sc1 sc01 sc0; do
  echo "== $v"; PNR_LIB_PATH=build/ab/libpnr_$v.so timeout 200 python tools/train_profile.py 2>&1 | grep -E "mlp_forward_train|mlp_backward" | tee -a $O/profile_${v}.log
done
timeout 600 python bench.py 2>&1 | tail -1 > $O/bench.json; python - <<'P'
import json
d=json.loads(open('gpurun_out/r02g/bench.json').read())
print(d['value'], d['roofline']['frac'], d['roofline_composite']['frac'], json.dumps(d['train_step'])[:600])
P
