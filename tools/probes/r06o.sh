set -x
timeout 1500 python -m pytest tests/test_gpu_wide_k.py tests/test_gpu_split.py tests/test_gpu_hardening.py tests/test_gpu_storage_modes.py -x -q 2>&1 | tail -25 > gpurun_out/r06o_tests.log
timeout 1500 python -m pytest tests/test_gpu_headline_sizes.py -x -q -k "gemm_vs_oracle or k50 or 6p25m_f32" 2>&1 | tail -8 >> gpurun_out/r06o_tests.log
timeout 1500 python -m pytest tests/test_gpu_switches.py -x -q -k "WIDE_SMALL_K or COSINE_NORMALISED or BF16_SEED" 2>&1 | tail -8 >> gpurun_out/r06o_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> gpurun_out/r06o_tests.log
F="--steps 20 --warmup 5 --no-hnsw --no-metrics-leg --no-bf16-leg --no-sq8-leg --no-traffic-pass --no-latency-legs --no-sharded-leg --no-m128-leg --no-cpu-baseline"
python bench.py $F > gpurun_out/r06o_bench.json 2> gpurun_out/r06o_bench.err
cp bench_legs.json gpurun_out/r06o_bench_legs.json
python bench.py $F --select-level 2 > gpurun_out/r06o_bench_l2.json 2> gpurun_out/r06o_bench_l2.err
cp bench_legs.json gpurun_out/r06o_bench_l2_legs.json
cat gpurun_out/r06o_tests.log
