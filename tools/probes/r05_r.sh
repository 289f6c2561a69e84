export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05r
mkdir -p $O
cd $R
VARIANTS="base noshared base noshared" bash tools/probes/r05_q.sh
for m in hamming cosine; do
  lib=tools/probes/out/libvelesdb_hip_stamp1.so; [ $m = hamming ] && lib=tools/probes/out/libvelesdb_hip_stamp1h.so
  timeout 300 python tools/probes/pp_stamp_probe.py $lib 1000000 $m > $O/stamp1_$m.log 2>&1; echo "rc=$?"
  grep -v amdgpu.ids $O/stamp1_$m.log | grep "sum of\|epilogue\|look\|row tiles\|^wave"
done
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py tests/test_gpu_round5_parity.py "tests/test_gpu_headline_sizes.py::test_headline_1m_gemm_vs_oracle" -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
for f in "--select" "--bf16-big" "--bits-big"; do
  timeout 300 python tools/fuzz_sweep.py $f --seconds 60 --seed 551 2>&1 | grep -v amdgpu.ids | tail -2 > $O/fuzz$f.log; echo "fuzz $f: $(tail -1 $O/fuzz$f.log)"
done
