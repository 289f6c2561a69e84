set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
HL="--steps 20 --warmup 3 --no-cpu-baseline --check-queries 64 --no-tiles --no-hnsw --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass --no-latency-legs"
for m in 1 2 1 2; do
  VELESDB_BF16_PP=$m timeout 600 python bench.py $HL > $O/hl_pp$m.json 2> $O/hl_pp$m.err; echo "pp=$m rc=$?"
  python - <<PY
import json
l=json.loads(open("$O/hl_pp$m.json").read().strip().splitlines()[-1])
print("pp=$m", "value", l["value"], "ms_per_step", l["ms_per_step"], "frac", l["roofline"]["frac"], "kernel_ms", l["roofline"]["kernel_ms"], "parity", l["parity_check"], "bf16:", l["legs"].get("bf16_gemm"))
PY
done
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py tests/test_gpu_round5_parity.py tests/test_gpu_storage_modes.py "tests/test_gpu_headline_sizes.py::test_headline_1m_gemm_vs_oracle" "tests/test_gpu_headline_sizes.py::test_configs3_full_size_10m_bf16_vs_oracle" -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
for f in "--select" "--bf16-big" "--bits-big" "--euclid"; do
  timeout 300 python tools/fuzz_sweep.py $f --seconds 60 --seed 501 2>&1 | grep -v amdgpu.ids | tail -2 > $O/fuzz$f.log; echo "fuzz $f: $(tail -1 $O/fuzz$f.log)"
done
timeout 300 python tools/fuzz_storage.py --select --seconds 60 --seed 502 2>&1 | grep -v amdgpu.ids | tail -2 > $O/fuzz_storage_select.log; echo "fuzz storage: $(tail -1 $O/fuzz_storage_select.log)"
