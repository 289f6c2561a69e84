#!/bin/bash
# scan + dense-finish epilogue: parity (bf16 result mode, selection, headline sizes, fuzz), then the launch-cost probe and the step timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03o
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py tests/test_gpu_headline_sizes.py tests/test_gpu_storage_modes.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.log
timeout 400 python tools/fuzz_sweep.py --select --seconds 120 --seed 21 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_select.log
timeout 400 python tools/fuzz_sweep.py --bf16-big --seconds 120 --seed 22 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_bf16_big.log
timeout 300 python tools/probes/t0_probe.py 2>&1 | grep "^T=" | tee $O/t0.log
timeout 300 python tools/probes/t0_probe.py --extra 64 --tiles 3,8 2>&1 | grep "^T=" | tee $O/t0_ragged.log
timeout 300 python tools/probes/split_probe.py --reps 10 2>&1 | grep -E "split=|identical" | tee $O/split.log
timeout 300 python tools/probes/bf16_glds_probe.py --rows 10000000 --reps 3 2>&1 | tail -1 | tee $O/bf16_10m.log
bash tools/probes/r03_h.sh > /dev/null 2>&1; cp gpurun_out/r03h/timeline.txt $O/timeline.txt; cat $O/timeline.txt | cut -c1-100
