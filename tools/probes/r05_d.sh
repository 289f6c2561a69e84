set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
timeout 120 tools/probes/out/multi_cu_walk_step 1000000 2000 > $O/multi_cu_1m.log 2>&1; echo "multi_cu 1M rc=$?"; cat $O/multi_cu_1m.log
timeout 120 tools/probes/out/multi_cu_walk_step 10000 2000 > $O/multi_cu_10k.log 2>&1; echo "multi_cu 10K rc=$?"; cat $O/multi_cu_10k.log
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
wc -c $O/bench_line.json; cat $O/bench_line.json; tail -3 $O/bench.err
cp bench_legs.json $O/ 2>/dev/null
timeout 2400 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_build.py tests/test_gpu_config0.py tests/test_gpu_switches.py tests/test_gpu_sweep.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 $O/pytest.log
