export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_split.py -x -q --durations=5 2>&1 | tail -30 | tee $O/pytest_split.log
timeout 300 python tools/probes/split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/split_probe.log
timeout 300 python tools/probes/split_probe.py --metric dot --nq 900 --rows 500000 2>&1 | grep -v amdgpu.ids | tee -a $O/split_probe.log
