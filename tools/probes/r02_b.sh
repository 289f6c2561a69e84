export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q --durations=8 > $O/pytest_sharded.log 2>&1; echo "pytest rc=$?" >> $O/pytest_sharded.log
tail -25 $O/pytest_sharded.log
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_headline_sizes.py --deselect tests/test_gpu_sharded.py > $O/pytest_rest.log 2>&1; echo "pytest rc=$?" >> $O/pytest_rest.log
tail -6 $O/pytest_rest.log
