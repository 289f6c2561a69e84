export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02t
mkdir -p $O
cd $R
timeout 300 python tools/probes/split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/split_probe.log
timeout 300 python tools/probes/split_probe.py --metric dot 2>&1 | grep -v amdgpu.ids | tee -a $O/split_probe.log
timeout 1200 python -m pytest tests/test_gpu_split.py tests/test_gpu_bf16.py -x -q --durations=5 2>&1 | tail -15 | tee $O/pytest_split.log
