export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05s
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py tests/test_gpu_round5_parity.py tests/test_gpu_sweep.py "tests/test_gpu_headline_sizes.py::test_headline_1m_gemm_vs_oracle" -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
for f in "--select" "--bf16-big" "--bits-big" ""; do
  timeout 300 python tools/fuzz_sweep.py $f --seconds 60 --seed 561 2>&1 | grep -v amdgpu.ids | tail -2 > $O/fuzz$f.log; echo "fuzz $f: $(tail -1 $O/fuzz$f.log)"
done
timeout 300 python tools/fuzz_storage.py --select --seconds 45 --seed 562 2>&1 | grep -v amdgpu.ids | tail -1
VARIANTS="base" bash tools/probes/r05_q.sh
