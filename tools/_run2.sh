mkdir -p gpurun_out/r06c
O=gpurun_out/r06c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_stages.py tests/test_gpu_configs.py -m gpu -q --tb=short -p no:cacheprovider -k "pdf or frame_scale" 2>&1 | tail -30 > $O/pytest.log
tail -8 $O/pytest.log
for v in head wpe8 general; do
  L=panopticnerf_amd/libpnr.so; E=""
  [ $v = wpe8 ] && L=build/ab/libpnr_wpe8.so
  [ $v = general ] && E="PNR_SAMPLE_PDF_GENERAL=1"
  env $E PNR_LIB_PATH=$L timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --cpu1-seconds 0 --train-steps 0 > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d=json.loads([l for l in open("$O/bench_$v.json") if l.startswith("{")][-1])
fa=d["frame_accounting"]
print("$v", d["value"], d["ms_per_step"], d["roofline"]["frac"], "small", fa["small_kernels_ms"], {k:v["ms"] for k,v in fa["small_kernels"].items()})
PY
done
