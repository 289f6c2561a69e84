#!/bin/bash
# is time = energy?  the ping-pong kernel with an artificial idle bubble per row tile (s_sleep), and the board's power limits
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03g
rocm-smi --showmaxpower --showpower 2>&1 | grep -E "Power|power" | tee gpurun_out/r03g/limits.log
rocm-smi --showperflevel --showclocks 2>&1 | grep -E "sclk|Perf" | tee -a gpurun_out/r03g/limits.log
for v in base sleep2 sleep8; do
  lib=velesdb_amd/lib/libvelesdb_hip.so
  [ $v != base ] && lib=tools/probes/out/libvelesdb_hip_$v.so
  ( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/r03g/smi_$v.log 2>&1 &
  SMI=$!
  echo "== $v"; VELESDB_HIP_LIB=$PWD/$lib timeout 300 python tools/probes/bf16_glds_probe.py --rows 4000000 --reps 400 2>&1 | tail -1
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  sort gpurun_out/r03g/smi_$v.log | uniq -c | sort -rn | head -4
done 2>&1 | tee gpurun_out/r03g/sleep.log
