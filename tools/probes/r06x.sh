# round 6, last tree: a longer fuzz of the newest paths (WIDE selection with the final bound inside the re-scoring kernel)
set -x
timeout 700 python tools/fuzz_sweep.py --wide --seconds 420 --seed 900 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r06x_fuzz_wide.log
timeout 400 python tools/fuzz_sweep.py --select --seconds 200 --seed 901 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r06x_fuzz_select.log
timeout 400 python tools/fuzz_storage.py --select --seconds 150 --seed 902 2>&1 | grep -v amdgpu.ids | tail -2 > gpurun_out/r06x_fuzz_storage_select.log
cat gpurun_out/r06x_fuzz_*.log
