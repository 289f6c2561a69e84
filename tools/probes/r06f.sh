set -x
timeout 900 python -m pytest tests/test_gpu_wide_k.py -x -q 2>&1 | tail -25 > gpurun_out/r06f_tests.log
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_sweep.py tests/test_gpu_storage_modes.py tests/test_gpu_round5_parity.py -x -q 2>&1 | tail -12 >> gpurun_out/r06f_tests.log
F="--steps 20 --warmup 5 --no-hnsw --no-metrics-leg --no-bf16-leg --no-sq8-leg --no-traffic-pass --no-latency-legs --no-sharded-leg --no-m128-leg --no-cpu-baseline"
python bench.py $F > gpurun_out/r06f_poolsel.json 2> gpurun_out/r06f_poolsel.err
cp bench_legs.json gpurun_out/r06f_poolsel_legs.json
VELESDB_POOL_SELECT=0 python bench.py $F --lib velesdb_amd/lib/libvelesdb_hip_probe.so > gpurun_out/r06f_merge.json 2> gpurun_out/r06f_merge.err
cp bench_legs.json gpurun_out/r06f_merge_legs.json
cat gpurun_out/r06f_tests.log
