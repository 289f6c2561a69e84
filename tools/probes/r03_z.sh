#!/bin/bash
# ragged tile through the group masks; non-speculative latency mode for cache-resident corpora: parity, fuzz, timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03z
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py tests/test_gpu_headline_sizes.py tests/test_gpu_storage_modes.py tests/test_gpu_riders.py tests/test_gpu_sweep.py tests/test_gpu_hnsw.py tests/test_gpu_config0.py tests/test_gpu_int8.py -x -q -m gpu 2>&1 | tail -4 | tee $O/pytest.log
timeout 400 python tools/fuzz_sweep.py --select --seconds 150 --seed 61 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_select.log
timeout 300 python tools/fuzz_sweep.py --euclid --seconds 90 --seed 62 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_euclid.log
timeout 300 python tools/fuzz_sweep.py --bf16-big --seconds 60 --seed 63 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_bf16.log
VELESDB_HNSW_LATENCY_MODE=3 timeout 300 python tools/fuzz_hnsw.py --seconds 90 --seed 64 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_hnsw_lat3.log
timeout 300 python tools/fuzz_hnsw.py --seconds 60 --seed 65 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/fuzz_hnsw.log
for m in cosine euclidean; do timeout 300 python tools/probes/split_probe.py --reps 30 --metric $m 2>&1 | grep -E "split=2|identical"; done | tee $O/split.log
for rows in 10000 100000; do timeout 300 python tools/hnsw_probe.py --rows $rows --efs 128 --nqs 1,16 2>&1 | grep "nq=\|ef="; done | tee $O/hnsw_small.log
