set -x
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06m_prof11 -- python $R/tools/probes/wide_small_k_probe.py 11 > $R/gpurun_out/r06m_k11.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06m_prof10 -- python $R/tools/probes/wide_small_k_probe.py 10 > $R/gpurun_out/r06m_k10.log 2>&1
find $R/gpurun_out/r06m_prof1* -name "*_kernel_trace.csv" -delete
python $R/tools/probes/wide_small_k_probe.py 10,11,10,11 > $R/gpurun_out/r06m_ab.log 2>&1
tail -3 $R/gpurun_out/r06m_k11.log $R/gpurun_out/r06m_k10.log; cat $R/gpurun_out/r06m_ab.log
