#!/bin/bash
# two-steps-ahead row pipeline in split_rerank_verify: parity of the selection stage's users, fuzz, timeline, headline bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04rr}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_headline_sizes.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.log
timeout 200 python tools/fuzz_sweep.py --select --seconds 60 --seed 97 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/fuzz_select.log
bash tools/probes/r03_h.sh > /dev/null 2>&1; cp gpurun_out/r03h/timeline.txt $O/timeline.txt; grep -E "rerank|seed_scores|step span" $O/timeline.txt | cut -c1-110
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-tiles --no-hnsw --no-bf16-leg --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass > $O/bench_headline.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_headline.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["unproven_queries_last_batch"], d["parity_check"]["scores_bit_equal_oracle_canonical"])
PY
