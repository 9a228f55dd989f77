mkdir -p gpurun_out/t18
python tools/mlp_trace_fused.py trace 2>&1 | tee gpurun_out/t18/trace.log | tail -62
