set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_int8.py tests/test_gpu_callers.py tests/test_gpu_bf16.py tests/test_gpu_split.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
HL="--steps 20 --warmup 3 --no-cpu-baseline --check-queries 64 --no-tiles --no-hnsw --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass --no-latency-legs --no-bf16-leg"
for m in 0 1 2 0 1 2; do
  VELESDB_PP_PRIO=$m timeout 600 python bench.py $HL > $O/hl_prio$m.json 2> $O/hl_prio$m.err; echo "prio=$m rc=$?"
  python - <<PY
import json
l=json.loads(open("$O/hl_prio$m.json").read().strip().splitlines()[-1])
print("prio=$m", "value", l["value"], "ms_per_step", l["ms_per_step"], "frac", l["roofline"]["frac"], "kernel_ms", l["roofline"]["kernel_ms"], "parity", l["parity_check"])
PY
done
timeout 300 python tools/fuzz_hnsw.py --seconds 60 --seed 503 2>&1 | grep -v amdgpu.ids | tail -2
