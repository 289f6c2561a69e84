set -x
timeout 900 python -m pytest tests/test_gpu_wide_k.py tests/test_gpu_split.py -x -q 2>&1 | tail -5 > gpurun_out/r06n_tests.log
python tools/probes/wide_small_k_probe.py 10,11,20,32,33,10,11 > gpurun_out/r06n_ab.log 2>&1
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06n_prof -- python $R/tools/probes/wide_small_k_probe.py 10,11 > /dev/null 2>&1
find $R/gpurun_out/r06n_prof -name "*_kernel_trace.csv" -delete
cat $R/gpurun_out/r06n_tests.log $R/gpurun_out/r06n_ab.log
