cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05t
python $R/tools/probes/bits_single_query_probe.py hamming 2>&1 | grep -v amdgpu.ids
python $R/tools/probes/bits_single_query_probe.py jaccard 2>&1 | grep -v amdgpu.ids
cd $R
timeout 1500 python -m pytest tests/test_gpu_round5_parity.py -k one_launch -x -q > gpurun_out/r05t/pytest1.log 2>&1; echo "pytest one_launch rc=$?"; tail -5 gpurun_out/r05t/pytest1.log
timeout 1500 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_hardening.py tests/test_gpu_riders.py -x -q > gpurun_out/r05t/pytest2.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05t/pytest2.log
for f in "--bits" ""; do timeout 200 python tools/fuzz_sweep.py $f --seconds 40 --seed 571 2>&1 | grep -v amdgpu.ids | tail -1; done
