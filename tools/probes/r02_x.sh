export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02x
mkdir -p $O
cd $R
true
timeout 1500 python -m pytest tests/test_gpu_split.py tests/test_gpu_storage_modes.py tests/test_gpu_sweep.py -x -q --durations=6 2>&1 | tail -16 | tee $O/pytest.log
