#!/bin/bash
# self-listing unproven queries + stats in select_finish + gated fallback merge; merge_topk_select all-pairs rank for <= 256 keys:
# timelines of one headline step for the merge variants, then parity + fuzz + headline bench on the default build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04z}
mkdir -p $O
for v in default rank0 rank512; do
  if [ $v = default ]; then unset VELESDB_HIP_LIB; else export VELESDB_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/out/libvelesdb_hip_$v.so; fi
  bash tools/probes/r03_h.sh > /dev/null 2>&1; cp gpurun_out/r03h/timeline.txt $O/timeline_$v.txt; echo "== $v"; cut -c1-110 $O/timeline_$v.txt
done
unset VELESDB_HIP_LIB
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py tests/test_gpu_headline_sizes.py tests/test_gpu_storage_modes.py tests/test_gpu_sweep.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.log
timeout 400 python tools/fuzz_sweep.py --select --seconds 150 --seed 51 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_select.log
timeout 300 python tools/fuzz_storage.py --seconds 90 --seed 53 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_storage.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-tiles --no-hnsw --no-bf16-leg --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass > $O/bench_headline.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench_headline.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["parity_check"])
PY
