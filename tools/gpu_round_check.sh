set -x
export TMPDIR=/tmp
TAG=${TAG:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ -z "$SKIP_TESTS" ]; then
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
fi
SECONDS=0
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? wall ${SECONDS}s" | tee -a $O/bench.err
cp bench_legs.json $O/bench_legs_1gpu.json
tail -c 2500 $O/bench_line.json
tail -5 $O/bench.err
cd /tmp
# kernel stats of the headline step alone (every launch of the selection kernel in this pass belongs to a 1024-query step)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --check-queries 0 --no-tiles --no-hnsw --no-bf16-leg --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass --no-m128-leg > $O/bench_under_rocprof_headline.json 2> $O/rocprof_headline.err; echo "rocprof headline rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --check-queries 0 --no-embedding-leg --no-traffic-pass --no-sharded-leg --no-m128-leg > $O/bench_under_rocprof.json 2> $O/rocprof.err; echo "rocprof rc=$?"
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*.csv" | head -20
du -sh $O
