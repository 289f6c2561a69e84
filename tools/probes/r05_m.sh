cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05m
python $R/tools/probes/bits_single_query_probe.py hamming 2>&1 | grep -v amdgpu.ids
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05m/prof -- python $R/tools/probes/bits_single_query_probe.py hamming > $R/gpurun_out/r05m/under_rocprof.log 2>&1
find $R/gpurun_out/r05m -name "*kernel_stats.csv" -exec head -12 {} \; | cut -c1-200
find $R/gpurun_out/r05m -name "*_kernel_trace.csv" -delete
