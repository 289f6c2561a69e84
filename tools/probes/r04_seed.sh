#!/bin/bash
# split-k seed kernel: parity of everything that takes the selection stage, fuzz, timeline of one headline step, headline bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04seed}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_split.py tests/test_gpu_storage_modes.py tests/test_gpu_headline_sizes.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python tools/fuzz_sweep.py --select --seconds 100 --seed 95 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/fuzz_select.log
timeout 300 python tools/fuzz_storage.py --select --seconds 60 --seed 96 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/fuzz_storage_select.log
bash tools/probes/r03_h.sh > /dev/null 2>&1; cp gpurun_out/r03h/timeline.txt $O/timeline.txt; cut -c1-110 $O/timeline.txt
METRIC=euclidean bash tools/probes/r03_h.sh > /dev/null 2>&1; cp gpurun_out/r03h/timeline.txt $O/timeline_l2.txt; head -8 $O/timeline_l2.txt | cut -c1-110
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-tiles --no-hnsw --no-bf16-leg --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass > $O/bench_headline.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_headline.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["unproven_queries_last_batch"])
PY
