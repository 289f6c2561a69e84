#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/probes/walk_layout_probe.py > /tmp/kt.log 2>&1
tail -1 /tmp/kt.log | cut -c1-200
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/kt/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'hnsw_search_kernel' in r['Kernel_Name']]
for r in rows[-3:]:
    print(r['Kernel_Name'][:60], 'grid', r['Grid_Size_X'], 'wg', r['Workgroup_Size_X'], 'vgpr', r['VGPR_Count'], 'agpr', r['Accum_VGPR_Count'], 'sgpr', r['SGPR_Count'], 'lds', r['LDS_Block_Size'], 'scratch', r['Scratch_Size'], 'ms', (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
PY
