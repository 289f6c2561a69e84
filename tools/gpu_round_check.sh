set -x
export TMPDIR=/tmp
TAG=${TAG:-r01}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/${TAG:-r01}
timeout 1500 python -m pytest tests -m gpu -x -q > $R/gpurun_out/${TAG:-r01}/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $R/gpurun_out/${TAG:-r01}/pytest_gpu.log
tail -5 $R/gpurun_out/${TAG:-r01}/pytest_gpu.log
timeout 900 python bench.py > $R/gpurun_out/${TAG:-r01}/bench_line.json 2> $R/gpurun_out/${TAG:-r01}/bench.err; echo "bench rc=$?"
tail -c 3000 $R/gpurun_out/${TAG:-r01}/bench_line.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG:-r01}/prof -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --check-queries 0 --no-embedding-leg > $R/gpurun_out/${TAG:-r01}/bench_under_rocprof.json 2> $R/gpurun_out/${TAG:-r01}/rocprof.err; echo "rocprof rc=$?"
# headline kernel alone: every launch of the GEMM kernel in this pass is one 1024-query step, so the AverageNs of the stats file is the per-step figure
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG:-r01}/prof_headline -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --check-queries 0 --no-tiles --no-hnsw --no-bf16-leg --no-metrics-leg > $R/gpurun_out/${TAG:-r01}/bench_under_rocprof_headline.json 2> $R/gpurun_out/${TAG:-r01}/rocprof_headline.err; echo "rocprof headline rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG:-r01}/pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --check-queries 0 --hnsw-steps 1 --no-tiles --no-embedding-leg --no-bf16-leg --no-metrics-leg > $R/gpurun_out/${TAG:-r01}/bench_under_pmc.json 2> $R/gpurun_out/${TAG:-r01}/pmc.err; echo "pmc rc=$?"
# the raw per-dispatch traces are tens of MiB (gpurun copies back <= 64 MiB): the stats / counter files are what gets summarised
find $R/gpurun_out/${TAG:-r01} -name "*_kernel_trace.csv" -delete
find $R/gpurun_out/${TAG:-r01} -name "*.csv" | head -20
du -sh $R/gpurun_out/${TAG:-r01}
