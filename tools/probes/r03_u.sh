#!/bin/bash
# fused front kernel + Euclidean bf16 seed / measured bound: parity, fuzz, timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03u
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py tests/test_gpu_headline_sizes.py tests/test_gpu_storage_modes.py tests/test_gpu_riders.py tests/test_gpu_sweep.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.log
timeout 400 python tools/fuzz_sweep.py --select --seconds 200 --seed 41 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_select.log
timeout 300 python tools/probes/split_probe.py --reps 20 2>&1 | grep -E "split=|identical" | tee $O/split_cosine.log
timeout 300 python tools/probes/split_probe.py --reps 20 --metric euclidean 2>&1 | grep -E "split=|identical" | tee $O/split_euclidean.log
timeout 300 python tools/probes/split_probe.py --reps 20 --metric dot 2>&1 | grep -E "split=|identical" | tee $O/split_dot.log
