set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
V=$R/tools/probes/out/libvelesdb_hip_stagger.so
HL="--steps 20 --warmup 3 --no-cpu-baseline --check-queries 64 --no-tiles --no-hnsw --no-sq8-leg --no-sharded-leg --no-traffic-pass --no-latency-legs --no-metrics-leg"
for m in base stagger base stagger; do
  if [ $m = base ]; then L=""; else L="--lib $V"; fi
  timeout 900 python bench.py $HL $L > $O/hl_$m.json 2> $O/hl_$m.err; echo "$m rc=$?"
  python - <<PY
import json
l=json.loads(open("$O/hl_$m.json").read().strip().splitlines()[-1])
print("$m", "value", l["value"], "ms_per_step", l["ms_per_step"], "frac", l["roofline"]["frac"], "kernel_ms", l["roofline"]["kernel_ms"], "parity", l["parity_check"], "bf16:", l["legs"].get("bf16_gemm"))
PY
done
timeout 300 python tools/probes/pp_stamp_probe.py tools/probes/out/libvelesdb_hip_stagger_stamp2.so > $O/stagger_stamp2.log 2>&1; grep -v amdgpu.ids $O/stagger_stamp2.log | tail -24
VDB_TEST_LIB=$V timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py "tests/test_gpu_headline_sizes.py::test_headline_1m_gemm_vs_oracle" -x -q > $O/pytest_stagger.log 2>&1; echo "pytest(stagger) rc=$?"
tail -3 $O/pytest_stagger.log
