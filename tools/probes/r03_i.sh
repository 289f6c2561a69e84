#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03i
timeout 2400 python -m pytest tests/test_gpu_split.py tests/test_gpu_sweep.py tests/test_gpu_bf16.py tests/test_gpu_storage_modes.py tests/test_gpu_headline_sizes.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r03i/pytest.log
timeout 600 python tools/probes/split_probe.py --reps 10 2>&1 | grep -E "split=|identical" | tee gpurun_out/r03i/split_probe.log
bash tools/probes/r03_h.sh > /dev/null 2>&1; cp gpurun_out/r03h/timeline.txt gpurun_out/r03i/timeline.txt; cat gpurun_out/r03i/timeline.txt
