export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05o
mkdir -p $O
for m in hamming cosine; do
  lib=tools/probes/out/libvelesdb_hip_stamp1.so; [ $m = hamming ] && lib=tools/probes/out/libvelesdb_hip_stamp1h.so
  timeout 300 python tools/probes/pp_stamp_probe.py $lib 1000000 $m > $O/stamp1_$m.log 2>&1; echo "rc=$?"
  grep -v amdgpu.ids $O/stamp1_$m.log | grep "sum of\|epilogue\|look\|row tiles\|^wave"
done
