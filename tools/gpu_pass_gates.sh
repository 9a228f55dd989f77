#!/bin/bash
# one bench line per BASELINE config on the final head (fused inference pass where it applies)
mkdir -p gpurun_out/r02i
for c in 1 2 3 4; do
  timeout 200 python bench.py --config $c --steps 5 --warmup 2 --cpu-seconds 0 --train-steps 0 2>/dev/null | tail -1 > gpurun_out/r02i/bench_config$c.json
  python - <<P
import json
d=json.loads(open("gpurun_out/r02i/bench_config$c.json").read())
print($c, d["value"], d["ms_per_step"], d["roofline"]["kernel"][:60], d["roofline"]["frac"], d.get("roofline_composite",{}).get("frac"))
P
done
