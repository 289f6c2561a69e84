#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04layout}
mkdir -p $O
run() { env "$@" timeout 300 python tools/probes/walk_layout_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/layout.log; }
run X=1
run X=1
run EMPTY_CACHE=1
run PRE=1024
run PRE=3000
run FREE=3000
run PRE=517,33
run X=1
