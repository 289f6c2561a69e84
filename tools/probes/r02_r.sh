export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02r
mkdir -p $O
cd $R
for bd in 1 0; do
  echo "== VELESDB_G16_BDIR=$bd" | tee -a $O/split_probe.log
  VELESDB_G16_BDIR=$bd timeout 300 python tools/probes/split_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/split_probe.log
  VELESDB_G16_BDIR=$bd timeout 300 python tools/probes/bf16_glds_probe.py --rows 4000000 2>&1 | grep -v amdgpu.ids | tee -a $O/bf16_probe.log
done
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_bf16.py -x -q --durations=5 2>&1 | tail -12 | tee $O/pytest_split.log
