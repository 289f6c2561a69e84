export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02j
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_sharded.py -q 2>&1 | tail -60 | tee $O/pytest_bf16.log
