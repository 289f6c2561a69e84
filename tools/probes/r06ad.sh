# round 6: SQ8 batches at k <= 10 on the WIDE selection by default — the tests of the paths it touches, the SQ8 selection fuzzer, then every GPU test
set -x
timeout 1500 python -m pytest tests/test_gpu_storage_modes.py tests/test_gpu_wide_k.py tests/test_gpu_round5_parity.py -x -q 2>&1 | tail -5 > gpurun_out/r06ad_tests.log
cat gpurun_out/r06ad_tests.log
timeout 400 python tools/fuzz_storage.py --select --seconds 200 --seed 905 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r06ad_fuzz.log
cat gpurun_out/r06ad_fuzz.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06ad_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06ad_pytest_gpu.log
tail -4 gpurun_out/r06ad_pytest_gpu.log
