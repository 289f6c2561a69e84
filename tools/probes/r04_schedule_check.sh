#!/bin/bash
# after moving gemm_schedule / sweep_gemm_bf16_plan into vdb_gemm_schedule.hpp: the users of the schedule on the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04sched}
mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.log
timeout 60 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-tiles --no-hnsw --no-bf16-leg --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass > $O/bench_headline.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_headline.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["unproven_queries_last_batch"], d["parity_check"])
PY
timeout 150 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_bf16.py tests/test_gpu_split.py -x -q -m gpu -k "bit_metric_batches or glds_exact or glds_result or level" --durations=6 2>&1 | tail -12 | tee $O/pytest.log
