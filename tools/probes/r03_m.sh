#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03m
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_headline_sizes.py tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r03m/pytest.log
timeout 300 python tools/fuzz_sweep.py --select --seconds 150 --seed 11 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r03m/fuzz_select.log
for sd in 1 0; do VELESDB_BF16_SEED=$sd timeout 300 python tools/probes/split_probe.py --reps 10 2>&1 | grep -E "split=2|identical"; done | tee gpurun_out/r03m/seed_ab.log
timeout 300 python tools/probes/bf16_glds_probe.py --rows 10000000 --reps 3 2>&1 | tail -1 | tee gpurun_out/r03m/bf16_10m.log
bash tools/probes/r03_h.sh > /dev/null 2>&1; cp gpurun_out/r03h/timeline.txt gpurun_out/r03m/timeline.txt; head -14 gpurun_out/r03m/timeline.txt; tail -1 gpurun_out/r03m/timeline.txt
