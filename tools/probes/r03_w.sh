#!/bin/bash
# batch admission in the traversal kernels: parity (tests + fuzz incl. forced latency mode), latency and throughput
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03w
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_build.py tests/test_gpu_int8.py tests/test_gpu_config0.py tests/test_gpu_riders.py tests/test_gpu_headline_sizes.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.log
timeout 300 python tools/fuzz_hnsw.py --seconds 120 --seed 51 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_hnsw.log
VELESDB_HNSW_LATENCY_MODE=2 timeout 300 python tools/fuzz_hnsw.py --seconds 90 --seed 52 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_hnsw_lat.log
timeout 600 python tools/hnsw_probe.py --rows 1000000 --efs 128 --nqs 1,4,16,64,8192 2>&1 | grep -v amdgpu.ids | grep "nq=\|build\|recall" | tee $O/probe_1m.log
timeout 300 python tools/hnsw_probe.py --rows 10000 --efs 128 --nqs 1,8192 2>&1 | grep -v amdgpu.ids | grep "nq=\|recall" | tee $O/probe_10k.log
