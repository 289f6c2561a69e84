export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02ak
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_hnsw.py tests/test_gpu_config0.py tests/test_gpu_build.py tests/test_gpu_hardening.py -x -q 2>&1 | tail -4 | tee $O/pytest.log
for m in 1 0; do
  echo "== VELESDB_HNSW_LATENCY_MODE=$m" | tee -a $O/latency_probe.log
  VELESDB_HNSW_LATENCY_MODE=$m timeout 600 python tools/hnsw_probe.py --rows 1000000 --efs 128 --nqs 1,4,8,16,17,64 2>&1 | grep -v amdgpu.ids | tee -a $O/latency_probe.log
done
