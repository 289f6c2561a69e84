#!/bin/bash
# where does a selection launch's time go?  pp-kernel launch durations of one headline step: product, no scan (quick test only),
# no quick test either
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03p
mkdir -p $O
for v in base noscan; do
  [ $v != base ] && export VELESDB_HIP_LIB=$R/tools/probes/out/libvelesdb_hip_$v.so
  VELESDB_TRACE_LEVELS=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$v -- python $R/tools/probes/split_probe.py --reps 3 > $O/probe_$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, os, sys
v = sys.argv[1]
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r03p'
f = glob.glob(O + '/trace_%s/*/*kernel_trace.csv' % v)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'round_queries_bf16' in n]
s0, s1 = (idx[4], idx[5] if len(idx) > 5 else len(rows)) if v == "base" else (idx[1], idx[2])
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows[s0:s1] if 'gemm_bf16_pp' in r['Kernel_Name']]
line = '%-8s pp launches (us): %s  sum %.1f' % (v, ' '.join('%.1f' % x for x in d), sum(d))
print(line); print(v, "all:", " ".join("%.1f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "gemm_bf16_pp" in r["Kernel_Name"]))
open(O + '/pp_launches.txt', 'a').write(line + '\n')
PY
  find $O/trace_$v -name "*.csv" -size +1M -delete
done
