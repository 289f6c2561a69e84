set -x
timeout 600 python tools/fuzz_storage.py --select --seconds 330 --seed 903 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r06y_fuzz_storage_select_wide.log
tail -4 gpurun_out/r06y_fuzz_storage_select_wide.log
