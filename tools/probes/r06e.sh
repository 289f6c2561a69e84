set -x
timeout 600 python tools/probes/wide_debug.py > gpurun_out/r06e_wide_debug.log 2>&1
timeout 900 python -m pytest tests/test_gpu_wide_k.py -x -q 2>&1 | tail -25 > gpurun_out/r06e_wide_tests.log
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_round5_parity.py -x -q 2>&1 | tail -8 >> gpurun_out/r06e_wide_tests.log
cat gpurun_out/r06e_wide_debug.log gpurun_out/r06e_wide_tests.log
