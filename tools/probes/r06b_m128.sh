set -x
F="--steps 3 --warmup 2 --no-hnsw --no-tiles --no-metrics-leg --no-bf16-leg --no-sq8-leg --no-traffic-pass --no-latency-legs --no-sharded-leg"
( time python bench.py $F ) > gpurun_out/r06b_m128_1m.json 2> gpurun_out/r06b_m128_1m.err
cp bench_legs.json gpurun_out/r06b_m128_1m_legs.json
python bench.py $F --no-cpu-baseline --m128-rows 300000 --lib velesdb_amd/lib/libvelesdb_hip_probe.so > gpurun_out/r06b_m128_300k_lat.json 2> gpurun_out/r06b_m128_300k_lat.err
cp bench_legs.json gpurun_out/r06b_m128_300k_lat_legs.json
VELESDB_HNSW_LATENCY_MODE=0 python bench.py $F --no-cpu-baseline --m128-rows 300000 --lib velesdb_amd/lib/libvelesdb_hip_probe.so > gpurun_out/r06b_m128_300k_thr.json 2> gpurun_out/r06b_m128_300k_thr.err
cp bench_legs.json gpurun_out/r06b_m128_300k_thr_legs.json
tail -3 gpurun_out/r06b_m128_1m.err
