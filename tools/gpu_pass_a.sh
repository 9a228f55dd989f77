#!/bin/bash
# Round-2 pass A: the new host-side work on a real MI355X -- all GPU tests, bench at every BASELINE config, strong-scaling path at world 1
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
O=gpurun_out/r02a
timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x --durations=12 -s 2>&1 | tail -150 > $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
timeout 300 python bench.py --steps 5 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 3000 $O/bench_c5.json; tail -3 $O/bench_c5.err
for c in 1 2 3 4; do
  timeout 200 python bench.py --config $c --steps 3 --warmup 1 --cpu-seconds 0 > $O/bench_c$c.json 2> $O/bench_c$c.err; tail -c 1500 $O/bench_c$c.json; tail -2 $O/bench_c$c.err
done
timeout 200 python bench.py --scaling strong --steps 3 --warmup 1 --cpu-seconds 0 --no-roofline --train-steps 0 > $O/bench_strong1.json 2> $O/bench_strong1.err; tail -c 1200 $O/bench_strong1.json; tail -2 $O/bench_strong1.err
