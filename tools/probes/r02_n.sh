export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
timeout 300 python tools/probes/split_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/split_probe.log
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -22 $O/pytest_gpu.log
