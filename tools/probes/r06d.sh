set -x
timeout 600 python tools/probes/wide_debug.py > gpurun_out/r06d_wide_debug.log 2>&1
VELESDB_HIP_LIB=velesdb_amd/lib/libvelesdb_hip_probe.so VELESDB_COSINE_NORMALISED=0 timeout 600 python tools/probes/wide_debug.py > gpurun_out/r06d_wide_debug_plain.log 2>&1
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_sweep.py -x -q 2>&1 | tail -8 > gpurun_out/r06d_split_tests.log
F="--steps 10 --warmup 3 --no-hnsw --no-metrics-leg --no-bf16-leg --no-sq8-leg --no-traffic-pass --no-latency-legs --no-sharded-leg --no-m128-leg --no-cpu-baseline"
python bench.py $F > gpurun_out/r06d_cosn.json 2> gpurun_out/r06d_cosn.err
cp bench_legs.json gpurun_out/r06d_cosn_legs.json
VELESDB_COSINE_NORMALISED=0 python bench.py $F --lib velesdb_amd/lib/libvelesdb_hip_probe.so > gpurun_out/r06d_plain.json 2> gpurun_out/r06d_plain.err
cp bench_legs.json gpurun_out/r06d_plain_legs.json
cat gpurun_out/r06d_wide_debug.log gpurun_out/r06d_wide_debug_plain.log gpurun_out/r06d_split_tests.log
