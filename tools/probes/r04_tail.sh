#!/bin/bash
# fused batch tail (selection_batch_tail / select_finish / gated merge) + all-pairs rank in merge_topk_select: parity, fuzz, timeline, bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04y}
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py tests/test_gpu_headline_sizes.py tests/test_gpu_storage_modes.py tests/test_gpu_sweep.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.log
timeout 400 python tools/fuzz_sweep.py --select --seconds 150 --seed 41 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_select.log
timeout 300 python tools/fuzz_sweep.py --bf16-big --seconds 60 --seed 42 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_bf16_big.log
timeout 300 python tools/fuzz_storage.py --seconds 90 --seed 43 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_storage.log
bash tools/probes/r03_h.sh > /dev/null 2>&1; cp gpurun_out/r03h/timeline.txt $O/timeline.txt; cut -c1-110 $O/timeline.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-tiles --no-hnsw --no-bf16-leg --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass > $O/bench_headline.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_headline.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["parity_check"])
PY
