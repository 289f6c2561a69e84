cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05x
mkdir -p $O
python $R/tools/probes/bits_single_query_probe.py hamming 2>&1 | grep -v amdgpu.ids
cd $R
timeout 2400 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_round5_parity.py tests/test_gpu_hardening.py tests/test_gpu_riders.py tests/test_gpu_storage_modes.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for f in "--bits" "" "--euclid" "--engine 0"; do timeout 200 python tools/fuzz_sweep.py $f --seconds 40 --seed 601 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 200 python tools/fuzz_index.py --seconds 40 --seed 602 2>&1 | grep -v amdgpu.ids | tail -1
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check-queries 0 --no-tiles --no-hnsw --no-bf16-leg --no-sharded-leg --no-traffic-pass --no-sq8-leg > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
f=json.load(open("bench_legs.json"))
print("latency_mode", json.dumps(f.get("latency_mode")))
for m in f["other_metrics"]: print(m["metric"], json.dumps(m["single_query"]))
PY
