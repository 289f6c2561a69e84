# round 6: the WIDE selection's launch schedule (a launch boundary = a kernel tail + a reseed launch) and the chip-settling probe
set -x
P=$GRAFT_REPO_ROOT/velesdb_amd/lib/libvelesdb_hip_probe.so
for S in 1,4,16 4,16,0 2,8,0 2,16,0 8,0,0 4,0,0 16,0,0 1,8,0; do
  VELESDB_HIP_LIB=$P VELESDB_WIDE_STEPS=$S python tools/probes/wide_small_k_probe.py 10,50,100 2>&1 | grep ms_per | sed "s/^/steps $S /" >> gpurun_out/r06s_steps.log
done
VELESDB_HIP_LIB=$P VELESDB_WIDE_STEPS=1,4,16 python tools/probes/wide_small_k_probe.py 10,50,100 2>&1 | grep ms_per | sed "s/^/steps 1,4,16 again /" >> gpurun_out/r06s_steps.log
python tools/probes/step_settle_probe.py > gpurun_out/r06s_settle.log 2>&1
cat gpurun_out/r06s_steps.log gpurun_out/r06s_settle.log
