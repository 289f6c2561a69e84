#!/bin/bash
# plain batched reads of the lists' bounds in the SQ8 and matrix-core streaming sweeps: parity, fuzz, the bench legs they serve
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04rt}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_storage_modes.py tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python tools/fuzz_sweep.py --seconds 80 --seed 91 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/fuzz_sweep.log
timeout 300 python tools/fuzz_storage.py --seconds 80 --seed 92 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/fuzz_storage.log
timeout 900 python bench.py --no-hnsw --no-bf16-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"])
print("latency_mode", d["latency_mode"])
print("tiles", [(t["queries"], t["kernel_ms"], t["qps"]) for t in d["tiles"]])
s=d["sq8_storage_mode"]; print("sq8 eight", s["eight_queries"], s["batch"]["qps"])
PY
