# round 6: wide_rerank_verify with its rows requested towards L2 up front, per-query words read beside the lists: parity + per-kernel times
set -x
timeout 900 python -m pytest tests/test_gpu_wide_k.py tests/test_gpu_split.py -x -q 2>&1 | tail -4 > gpurun_out/r06t_tests.log
timeout 600 python -m pytest tests/test_gpu_headline_sizes.py -x -q -k "1m" 2>&1 | tail -4 >> gpurun_out/r06t_tests.log
python tools/probes/wide_small_k_probe.py 10,50,100 2>&1 | grep ms_per >> gpurun_out/r06t_tests.log
cd /tmp && export TMPDIR=/tmp
for K in 10 100; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06t_prof$K -- python $GRAFT_REPO_ROOT/tools/probes/wide_small_k_probe.py $K > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/summarize_prof.py stats $GRAFT_REPO_ROOT/gpurun_out/r06t_prof$K/*/*_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r06t_k${K}_kernel_stats.csv || true
done
cd $GRAFT_REPO_ROOT
cat gpurun_out/r06t_tests.log; grep -i "wide_\|seed_scores" gpurun_out/r06t_k10_kernel_stats.csv gpurun_out/r06t_k100_kernel_stats.csv | cut -c1-160
