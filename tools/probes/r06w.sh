# round 6, last tree: the dual-precision tests (raw-ef branch), smoke(), the C ABI driver tests
set -x
timeout 900 python -m pytest tests/test_gpu_int8.py tests/test_gpu_callers.py -x -q 2>&1 | tail -4 > gpurun_out/r06w_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 >> gpurun_out/r06w_tests.log
cat gpurun_out/r06w_tests.log
