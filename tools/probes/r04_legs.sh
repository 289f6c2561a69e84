#!/bin/bash
# per-expansion legs of the latency-mode walk (tools/probes/hnsw_time_variant.py builds the instrumented library), 1 M x 768 and 10 K x 768
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04legs}
mkdir -p $O
for n in 1000000 10000; do
  for pf in 0 1; do
    echo "== rows $n prediction $pf"
    VELESDB_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/out/libvelesdb_hip_hnswprobe.so VELESDB_HNSW_PREFETCH_IDS=$pf timeout 600 python tools/probes/walk_prefetch_probe.py $n iid 2>&1 | grep "probe:" | sort | uniq -c | sort -rn | head -400 > $O/legs_${n}_pf$pf.txt
    python3 - <<PY
import re
rows=[]
for l in open("$O/legs_${n}_pf$pf.txt"):
    m=re.search(r"expansions (\d+)  pop\+ids ([\d.]+) us  barrier\+rows\+dist ([\d.]+) us  admit ([\d.]+) us \(prologue ([\d.]+), batch ([\d.]+)\).*total ([\d.]+) us", l)
    if m: rows.append([float(x) for x in m.groups()])
import statistics as st
if rows:
    med=[st.median(c) for c in zip(*rows)]
    print("launches %d: median expansions %.0f | pop+ids %.2f us | barrier+rows+distances %.2f us | admission %.2f us (prologue %.2f, batch %.2f) | walk %.1f us" % (len(rows), *med))
PY
  done
done 2>&1 | tee $O/legs.log
