#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03l
timeout 1500 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_sharded.py tests/test_gpu_split.py tests/test_gpu_storage_modes.py tests/test_gpu_hnsw.py -x -q -m gpu -s 2>&1 | grep -E "concurrency|passed|failed|Error|error|assert" | tail -12 | tee gpurun_out/r03l/pytest.log
