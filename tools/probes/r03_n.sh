#!/bin/bash
# fixed cost of a launch of the ping-pong bf16 kernel: T-sweep (time = t0 + T * t_tile), an empty-range variant, a no-epilogue
# variant, instruction-cache counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03n
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/tools/probes/t0_probe.py > $O/base.log 2>&1
grep "^T=" $O/base.log
python3 - <<'PY'
import csv, glob, os
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r03n'
f = glob.glob(O + '/trace/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if 'gemm_bf16_pp' in r['Kernel_Name']]
print('pp launches (us):', ' '.join('%.1f' % x for x in d))
open(O + '/pp_durations.txt', 'w').write(' '.join('%.1f' % x for x in d) + '\n')
PY
find $O/trace -name "*.csv" -size +2M -delete
for v in empty noepi; do
  echo "== $v"
  VELESDB_HIP_LIB=$R/tools/probes/out/libvelesdb_hip_$v.so timeout 300 python $R/tools/probes/t0_probe.py --tiles 3,8,32 2>&1 | grep "^T="
done 2>&1 | tee $O/variants.log
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_[A-Z0-9_]*\|SQ_IFETCH[A-Z0-9_]*\|SQ_INST_LEVEL[A-Z0-9_]*" | sort -u > $O/avail_sqc.txt
cat $O/avail_sqc.txt | tr '\n' ' '
for cs in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  n=$(echo $cs | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $O/pmc_$n -- python $R/tools/probes/t0_probe.py --tiles 3,32 --reps 2 > $O/pmc_$n.log 2>&1
  find $O/pmc_$n -name "*_kernel_trace.csv" -delete
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r03n'
out = open(O + '/pmc_summary.txt', 'w')
for d in sorted(glob.glob(O + '/pmc_*')):
    if not os.path.isdir(d): continue
    fs = glob.glob(d + '/*/*counter_collection.csv')
    if not fs:
        print(os.path.basename(d), 'no counters', file=out); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'gemm_bf16_pp' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print(os.path.basename(d), k, ' '.join('%.4g' % x for x in v), file=out)
out.close()
print(open(O + '/pmc_summary.txt').read())
PY
