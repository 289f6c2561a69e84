#!/bin/bash
# group-of-8 scan, fused reseed, batched rerank loads, no pool fill: parity, fuzz, timing, launch-schedule sweep
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03s
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py tests/test_gpu_headline_sizes.py tests/test_gpu_storage_modes.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.log
timeout 400 python tools/fuzz_sweep.py --select --seconds 150 --seed 31 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_select.log
timeout 400 python tools/fuzz_sweep.py --bf16-big --seconds 90 --seed 32 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_bf16_big.log
for st in "1,4,16" "1,4" "2,8" "4,16" "2,16" "1,8" "4" "8" "1,3,12" "1,6,24"; do
  echo "== steps $st"
  VELESDB_TRACE_LEVELS=2 VELESDB_SEL_STEPS=$st timeout 300 python tools/probes/split_probe.py --reps 20 2>&1 | grep -E "split=2"
done 2>&1 | tee $O/schedule_sweep.log
timeout 300 python tools/probes/bf16_glds_probe.py --rows 10000000 --reps 3 2>&1 | tail -1 | tee $O/bf16_10m.log
bash tools/probes/r03_h.sh > /dev/null 2>&1; cp gpurun_out/r03h/timeline.txt $O/timeline.txt; cat $O/timeline.txt | cut -c1-100
