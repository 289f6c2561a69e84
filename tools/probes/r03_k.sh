#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03k
timeout 900 python -m pytest tests/test_gpu_headline_sizes.py -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r03k/pytest_headline_sizes.log
timeout 400 python tools/fuzz_sweep.py --bf16-big --seconds 200 --seed 3 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r03k/fuzz_bf16_big.log
( time timeout 1500 python bench.py > gpurun_out/r03k/bench_line.json 2> gpurun_out/r03k/bench_stderr.log ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03k/bench_line.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac_step", d.get("frac_step"))
r = d["roofline"]; print("roofline", {k: r[k] for k in ("achieved", "frac", "traffic", "kernel_ms", "traffic_source") if k in r})
for leg in ("hnsw", "hnsw_embedding_like", "bf16_gemm"):
    l = d.get(leg) or {}
    rr = l.get("roofline", {})
    print(leg, l.get("qps"), {k: rr.get(k) for k in ("achieved", "frac", "traffic", "traffic_over_algorithmic", "traffic_source")})
print("bf16 parity", (d.get("bf16_gemm") or {}).get("parity_check"))
print("config0", (d.get("config0_10k") or {}).get("search_median_us"), "hnsw lat", (d.get("hnsw") or {}).get("latency_mode"))
PY
tail -5 gpurun_out/r03k/bench_stderr.log
