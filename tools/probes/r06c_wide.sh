set -x
timeout 900 python -m pytest tests/test_gpu_wide_k.py -x -q 2>&1 | tail -25 > gpurun_out/r06c_wide_tests.log
timeout 600 python -m pytest tests/test_gpu_hnsw.py -x -q -k "canonical_order_vs_reference" 2>&1 | tail -8 >> gpurun_out/r06c_wide_tests.log
timeout 900 python -m pytest tests/test_gpu_headline_sizes.py -x -q -k "k50 or gemm_vs_oracle" 2>&1 | tail -15 >> gpurun_out/r06c_wide_tests.log
F="--steps 10 --warmup 3 --no-hnsw --no-metrics-leg --no-bf16-leg --no-sq8-leg --no-traffic-pass --no-latency-legs --no-sharded-leg --no-m128-leg --no-cpu-baseline"
python bench.py $F > gpurun_out/r06c_kcurve.json 2> gpurun_out/r06c_kcurve.err
cp bench_legs.json gpurun_out/r06c_kcurve_legs.json
cat gpurun_out/r06c_wide_tests.log
