export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i
timeout 300 python tools/probes/pp_stamp_probe.py tools/probes/out/libvelesdb_hip_stamp1.so > gpurun_out/r05i/stamp1.log 2>&1; echo "rc=$?"
grep -v amdgpu.ids gpurun_out/r05i/stamp1.log | tail -40
