# round 6: the WIDE satellites (small-list k-th by counting, final reseed inside the rerank kernel, adaptive stage width):
# parity on the wide / split / headline tests, A / B of the fused form on one box, per-kernel breakdown
set -x
timeout 900 python -m pytest tests/test_gpu_wide_k.py tests/test_gpu_split.py tests/test_gpu_hardening.py tests/test_gpu_round5_parity.py -x -q 2>&1 | tail -6 > gpurun_out/r06r_tests.log
timeout 900 python -m pytest tests/test_gpu_switches.py -x -q -k "WIDE" 2>&1 | tail -6 >> gpurun_out/r06r_tests.log
timeout 900 python -m pytest tests/test_gpu_headline_sizes.py -x -q 2>&1 | tail -6 >> gpurun_out/r06r_tests.log
P=$GRAFT_REPO_ROOT/velesdb_amd/lib/libvelesdb_hip_probe.so
for i in 1 2; do
  VELESDB_HIP_LIB=$P VELESDB_WIDE_FUSE=0 python tools/probes/wide_small_k_probe.py 10,50,100 2>&1 | grep ms_per | sed 's/^/fuse0 /' >> gpurun_out/r06r_ab.log
  VELESDB_HIP_LIB=$P python tools/probes/wide_small_k_probe.py 10,50,100 2>&1 | grep ms_per | sed 's/^/fuse1 /' >> gpurun_out/r06r_ab.log
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r06r_prof -- python $GRAFT_REPO_ROOT/tools/probes/wide_small_k_probe.py 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_prof.py stats gpurun_out/r06r_prof/*/*_kernel_stats.csv gpurun_out/r06r_k10_kernel_stats.csv || true
cat gpurun_out/r06r_tests.log gpurun_out/r06r_ab.log
