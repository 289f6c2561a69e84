set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
V=$R/tools/probes/out/libvelesdb_hip_reqm.so
HL="--steps 20 --warmup 3 --no-cpu-baseline --check-queries 64 --no-tiles --no-hnsw --no-sq8-leg --no-sharded-leg --no-traffic-pass --no-latency-legs"
for m in base reqm base reqm; do
  if [ $m = base ]; then L=""; else L="--lib $V"; fi
  timeout 900 python bench.py $HL $L > $O/hl_$m.json 2> $O/hl_$m.err; echo "$m rc=$?"
  python - <<PY
import json
l=json.loads(open("$O/hl_$m.json").read().strip().splitlines()[-1])
om=l["legs"].get("other_metrics") or {}
print("$m", "value", l["value"], "ms_per_step", l["ms_per_step"], "frac", l["roofline"]["frac"], "kernel_ms", l["roofline"]["kernel_ms"], "parity", l["parity_check"], "bf16:", l["legs"].get("bf16_gemm"), "| hamming", om.get("hamming"), "| jaccard", om.get("jaccard"))
PY
done
VDB_TEST_LIB=$V timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py tests/test_gpu_round5_parity.py "tests/test_gpu_headline_sizes.py::test_headline_1m_gemm_vs_oracle" "tests/test_gpu_sweep.py" -x -q > $O/pytest_reqm.log 2>&1; echo "pytest(reqm) rc=$?"
tail -5 $O/pytest_reqm.log
timeout 1200 python -m pytest "tests/test_gpu_sweep.py" tests/test_gpu_round5_parity.py tests/test_gpu_storage_modes.py -x -q -k "bit or hamming or jaccard or binary or Hamming or Jaccard" > $O/pytest_bits.log 2>&1; echo "pytest(bits, product) rc=$?"
tail -5 $O/pytest_bits.log
for f in "--bits-big" "--bits"; do
  timeout 300 python tools/fuzz_sweep.py $f --seconds 60 --seed 511 2>&1 | grep -v amdgpu.ids | tail -1
done
