#!/bin/bash
# timeline of one SQ8 selection batch (level 3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03sq8
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/storage_probe.py --metric cosine --nqs 1024 > $O/probe.log 2>&1
tail -3 $O/probe.log
python3 - <<'PY'
import csv, glob, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r03sq8'
f = glob.glob(O + '/trace/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'sel16_prep_queries' in n]
s0 = idx[-1]
t0 = int(rows[s0]['Start_Timestamp']); prev = t0
for r in rows[s0:s0 + 40]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f us  dur %7.1f  gap %6.1f  %s' % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev) / 1e3, r['Kernel_Name'][:70]))
    prev = en
PY
find $O/trace -name "*.csv" -size +2M -delete
