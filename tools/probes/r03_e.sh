#!/bin/bash
# ablations of the ping-pong kernel: where does a k-tile's time go?  (timing only: variants 1/2/3/8 compute garbage)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03e
for e in ${EXPS:-0 8 9 10 11 12}; do
  lib=velesdb_amd/lib/libvelesdb_hip.so
  [ $e != 0 ] && lib=tools/probes/out/libvelesdb_hip_exp$e.so
  [ -f $lib ] || continue
  echo "== exp $e"
  VELESDB_HIP_LIB=$PWD/$lib timeout 300 python tools/probes/bf16_glds_probe.py --rows 4000000 --reps 5 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r03e/ablation.log
