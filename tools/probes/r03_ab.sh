#!/bin/bash
# where does the selection stage start to win for ONE partly filled 256-query tile?  (kSelectMinQueries)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03ab
mkdir -p $O
for m in cosine euclidean; do
for mn in 80 16; do
  echo "== $m VELESDB_SELECT_MIN_QUERIES=$mn"
  VELESDB_SELECT_MIN_QUERIES=$mn timeout 300 python tools/sweep_probe.py --metric $m --nqs 16,24,32,48,64,80,96,128,192,256,384,512 2>&1 | grep -v amdgpu.ids | tail -14
done
done 2>&1 | tee $O/min_queries.log
