set -x
timeout 1500 python -m pytest tests/test_gpu_switches.py -x -q -k "BF16_SEED or COSINE_NORMALISED or POOL_SELECT or GATHER_ALL" 2>&1 | tail -12 > gpurun_out/r06h_tests.log
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_wide_k.py tests/test_gpu_hardening.py -x -q 2>&1 | tail -6 >> gpurun_out/r06h_tests.log
F="--steps 20 --warmup 5 --no-hnsw --no-metrics-leg --no-bf16-leg --no-sq8-leg --no-traffic-pass --no-latency-legs --no-sharded-leg --no-m128-leg --no-cpu-baseline"
python bench.py $F > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err
cp bench_legs.json gpurun_out/r06h_bench_legs.json
cat gpurun_out/r06h_tests.log
