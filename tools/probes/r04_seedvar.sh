#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04seedvar}
mkdir -p $O
for v in product noloop nostore nonorms; do
  if [ $v = product ]; then unset VELESDB_HIP_LIB; else export VELESDB_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/out/libvelesdb_hip_seed_$v.so; fi
  bash tools/probes/r03_h.sh > /dev/null 2>&1
  echo "== $v: $(grep seed_scores gpurun_out/r03h/timeline.txt | cut -c1-60)  | $(grep 'step span' gpurun_out/r03h/timeline.txt)" | tee -a $O/seed_variants.log
done
