# FETCH_SIZE of the traversal kernel (own --pmc pass, counters + kernel trace only) -> profiles/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_hnsw
mkdir -p $O
CMD="bench.py --steps 3 --warmup 1 --no-cpu-baseline --check-queries 0 --hnsw-steps 1 --no-tiles --no-embedding-leg --no-bf16-leg --no-metrics-leg --no-sq8-leg --no-sharded-leg --no-traffic-pass --no-latency-legs --ef-curve 128"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc -- python $R/$CMD > $O/bench_line.json 2> $O/err.log; echo rc=$?
F=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_traffic.py "$F" $O/bench_line.json $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE --kernel-trace pass of $CMD (tools/probes/pmc_hnsw_traffic.sh, round 2 final code); bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section)"
cat $O/pmc_traffic.json
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
