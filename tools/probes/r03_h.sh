#!/bin/bash
# timeline of ONE headline step (1 M x 768, 1 024 queries, level 2): every kernel with start / duration / gap to the one before
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03h
rm -rf $O/trace; mkdir -p $O
VELESDB_TRACE_LEVELS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/probes/split_probe.py --reps 3 ${METRIC:+--metric $METRIC} > $O/probe.log 2>&1
tail -3 $O/probe.log
python3 - <<'PY'
import csv, glob, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r03h'
f = glob.glob(O + '/trace/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last level-2 step: from the last round_queries_bf16 before the first level-1 step
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'sel16_prep_queries' in n or 'l2_augment_queries' in n]
# level 2 runs first: 2 warm-up + 3 timed = 5 steps; take the 5th
s0 = idx[4]
s1 = idx[5] if len(idx) > 5 else len(rows)
# stop at the first kernel of the next step / level
out = open(O + '/timeline.txt', 'w')
t0 = int(rows[s0]['Start_Timestamp']); prev_end = t0
tot = 0
for r in rows[s0:s1]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'split_vectors' in r['Kernel_Name']: break
    print('%8.1f us  dur %7.1f  gap %6.1f  %s' % ((st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, r['Kernel_Name'][:70]), file=out)
    prev_end = en; tot = en - t0
print('step span %.1f us' % (tot / 1e3), file=out)
out.close()
print(open(O + '/timeline.txt').read())
PY
find $O/trace -name "*.csv" -size +2M -delete
