#!/bin/bash
# walk / construction kernels' LDS state through typed LDS pointers (ds_* instead of volatile flat_*): parity, then A / B against
# the library built from the previous sources (tools/probes/out/libvelesdb_hip_flatlds.so)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04q2}
OLD=$GRAFT_REPO_ROOT/tools/probes/out/libvelesdb_hip_flatlds.so
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_int8.py tests/test_gpu_build.py tests/test_gpu_riders.py tests/test_gpu_config0.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.log
timeout 300 python tools/fuzz_hnsw.py --seconds 90 --seed 81 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/fuzz_hnsw.log
VELESDB_HNSW_LATENCY_MODE=2 timeout 300 python tools/fuzz_hnsw.py --seconds 60 --seed 82 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/fuzz_hnsw_lat2.log
for r in 1 2; do
  echo "== flat (previous sources)"; VELESDB_HIP_LIB=$OLD timeout 600 python tools/probes/int8_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/throughput_ab.log
  echo "== ds (this tree)"; timeout 600 python tools/probes/int8_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/throughput_ab.log
done
for r in 1 2; do
  echo "== flat (previous sources), no prediction"; VELESDB_HIP_LIB=$OLD VELESDB_HNSW_PREFETCH_IDS=0 timeout 600 python tools/probes/walk_prefetch_probe.py 1000000 iid 2>&1 | grep -v amdgpu.ids | tee -a $O/latency_ab.log
  echo "== ds, no prediction"; VELESDB_HNSW_PREFETCH_IDS=0 timeout 600 python tools/probes/walk_prefetch_probe.py 1000000 iid 2>&1 | grep -v amdgpu.ids | tee -a $O/latency_ab.log
  echo "== ds, prediction"; VELESDB_HNSW_PREFETCH_IDS=1 timeout 600 python tools/probes/walk_prefetch_probe.py 1000000 iid 2>&1 | grep -v amdgpu.ids | tee -a $O/latency_ab.log
done
echo "== 10 K: flat / ds / ds + prediction"
VELESDB_HIP_LIB=$OLD VELESDB_HNSW_PREFETCH_IDS=0 timeout 300 python tools/probes/walk_prefetch_probe.py 10000 iid 2>&1 | grep -v amdgpu.ids | tee -a $O/latency_ab_10k.log
VELESDB_HNSW_PREFETCH_IDS=0 timeout 300 python tools/probes/walk_prefetch_probe.py 10000 iid 2>&1 | grep -v amdgpu.ids | tee -a $O/latency_ab_10k.log
VELESDB_HNSW_PREFETCH_IDS=1 timeout 300 python tools/probes/walk_prefetch_probe.py 10000 iid 2>&1 | grep -v amdgpu.ids | tee -a $O/latency_ab_10k.log
