#!/bin/bash
# round 3, first GPU call: the new parity tests of configs[3]'s kernel + the boundary tests added this round
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03a
timeout 1500 python -m pytest tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03a/pytest_bf16.log
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_hardening.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03a/pytest_boundary.log
cat gpurun_out/r03a/*.log
