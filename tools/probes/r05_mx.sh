export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05mx
mkdir -p $O
cd $R
HL="--steps 20 --warmup 3 --no-cpu-baseline --check-queries 64 --no-tiles --no-hnsw --no-sq8-leg --no-sharded-leg --no-traffic-pass --no-latency-legs --no-metrics-leg --no-bf16-leg --lib velesdb_amd/lib/libvelesdb_hip_probe.so"
for v in 1 0 1 0; do
  VELESDB_MERGE_EXTRACT=$v timeout 600 python bench.py $HL > $O/hl.json 2> $O/hl.err; rc=$?
  python - <<PY
import json
l=json.loads(open("$O/hl.json").read().strip().splitlines()[-1])
print("extract=$v rc=$rc", "value", l["value"], "ms_per_step", l["ms_per_step"], "frac", l["roofline"]["frac"], "kernel_ms", l["roofline"]["kernel_ms"], "parity", l["parity_check"])
PY
done
timeout 2400 python -m pytest tests/test_gpu_split.py tests/test_gpu_sweep.py tests/test_gpu_bf16.py tests/test_gpu_storage_modes.py tests/test_gpu_round5_parity.py "tests/test_gpu_headline_sizes.py::test_headline_1m_gemm_vs_oracle" -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for f in "--select" "" "--euclid" "--bits"; do timeout 300 python tools/fuzz_sweep.py $f --seconds 50 --seed 731 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 300 python tools/fuzz_storage.py --select --seconds 50 --seed 732 2>&1 | grep -v amdgpu.ids | tail -1
