#!/bin/bash
# neighbour-list prediction in the latency-mode walk: parity (tests + forced-latency-mode fuzz), then A / B at 10 K and 1 M rows
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04p2}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_riders.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.log
for lm in 2 3; do
  VELESDB_HNSW_LATENCY_MODE=$lm timeout 300 python tools/fuzz_hnsw.py --seconds 70 --seed $((70 + lm)) 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/fuzz_hnsw_lat$lm.log
done
for n in 10000 1000000; do
  for pf in 1 0 1 0; do
    VELESDB_HNSW_PREFETCH_IDS=$pf timeout 600 python tools/probes/walk_prefetch_probe.py $n iid 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_$n.log
  done
done
VELESDB_HNSW_PREFETCH_IDS=1 timeout 600 python tools/probes/walk_prefetch_probe.py 1000000 emb 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_emb.log
VELESDB_HNSW_PREFETCH_IDS=0 timeout 600 python tools/probes/walk_prefetch_probe.py 1000000 emb 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_emb.log
