#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04layout2}
mkdir -p $O
OLD=$GRAFT_REPO_ROOT/tools/probes/out/libvelesdb_hip_oldwalk.so
for r in 1 2; do
  echo "== walk kernel of commit 45d4f0a (volatile generic pointers)"; VELESDB_HIP_LIB=$OLD timeout 300 python tools/probes/walk_layout_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/ab.log
  echo "== this tree"; timeout 300 python tools/probes/walk_layout_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/ab.log
done
