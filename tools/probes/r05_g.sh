set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
for v in stamp1 stamp2; do
  timeout 300 python tools/probes/pp_stamp_probe.py tools/probes/out/libvelesdb_hip_$v.so > $O/$v.log 2>&1; echo "$v rc=$?"
  grep -v amdgpu.ids $O/$v.log | tail -30
done
