# round 6: SQ8 batches at k <= 10 through the WIDE selection (selector level 3) against the block-local lists — same box A / B on the
# bench's sq8 leg (probe build: VELESDB_WIDE_SMALL_K=0 restores the lists), then the tests of the paths it touches
set -x
P=$GRAFT_REPO_ROOT/velesdb_amd/lib/libvelesdb_hip_probe.so
F="--steps 5 --warmup 2 --no-hnsw --no-metrics-leg --no-bf16-leg --no-traffic-pass --no-latency-legs --no-sharded-leg --no-m128-leg --no-cpu-baseline --no-tiles --check-queries 0"
for i in 1 2; do
VELESDB_WIDE_SMALL_K=0 python bench.py --lib $P $F > /dev/null 2> gpurun_out/r06ac.err; python -c "import json; d=json.load(open('bench_legs.json'))['sq8_storage_mode']; print('lists', d['batch'])" >> gpurun_out/r06ac_ab.log
python bench.py --lib $P $F > /dev/null 2>> gpurun_out/r06ac.err; python -c "import json; d=json.load(open('bench_legs.json'))['sq8_storage_mode']; print('wide ', d['batch'], d.get('parity_check'))" >> gpurun_out/r06ac_ab.log
done
cat gpurun_out/r06ac_ab.log
timeout 1200 python -m pytest tests/test_gpu_storage_modes.py tests/test_gpu_wide_k.py -x -q 2>&1 | tail -5 > gpurun_out/r06ac_tests.log
cat gpurun_out/r06ac_tests.log
