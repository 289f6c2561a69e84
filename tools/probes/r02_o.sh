export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_sweep.py -q --durations=5 2>&1 | tail -12 | tee $O/pytest_split.log
timeout 300 python tools/probes/split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/split_probe.log
