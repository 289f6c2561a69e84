cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05n
python $R/tools/probes/bits_single_query_probe.py hamming 2>&1 | grep -v amdgpu.ids
python $R/tools/probes/bits_single_query_probe.py jaccard 2>&1 | grep -v amdgpu.ids
cd $R
timeout 1500 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_storage_modes.py tests/test_gpu_hardening.py tests/test_gpu_riders.py tests/test_gpu_vec_utils.py -x -q > gpurun_out/r05n/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05n/pytest.log
for f in "" "--engine 0" "--bits" "--euclid"; do timeout 200 python tools/fuzz_sweep.py $f --seconds 40 --seed 541 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 200 python tools/fuzz_index.py --seconds 60 --seed 542 2>&1 | grep -v amdgpu.ids | tail -1
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --check-queries 0 --no-tiles --no-hnsw --no-bf16-leg --no-sharded-leg --no-traffic-pass > gpurun_out/r05n/bench.json 2> gpurun_out/r05n/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
f=json.load(open("bench_legs.json"))
print("latency_mode", json.dumps(f.get("latency_mode")))
for m in f["other_metrics"]: print(m["metric"], json.dumps(m["single_query"]))
print("sq8", json.dumps({k: f["sq8_storage_mode"].get(k) for k in ("one_query","eight_queries","eight_queries_default_path")}))
PY
