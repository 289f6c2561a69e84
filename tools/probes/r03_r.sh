#!/bin/bash
# instruction counts of the selection launches: product vs no-scan vs no-push variants (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03r
mkdir -p $O
for v in base noscan nopush; do
  unset VELESDB_HIP_LIB
  [ $v != base ] && export VELESDB_HIP_LIB=$R/tools/probes/out/libvelesdb_hip_$v.so
  i=0
  for cs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    VELESDB_TRACE_LEVELS=2 timeout 300 rocprofv3 --pmc $cs --kernel-trace --output-format csv -d $O/${v}_$i -- python $R/tools/probes/split_probe.py --reps 1 > $O/${v}_$i.log 2>&1
    find $O/${v}_$i -name "*_kernel_trace.csv" -delete
  done
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r03r'
out = open(O + '/summary.txt', 'w')
for d in sorted(glob.glob(O + '/*_[0-9]')):
    fs = glob.glob(d + '/*/*counter_collection.csv')
    if not fs:
        print(os.path.basename(d), 'no counters', file=out); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'gemm_bf16_pp' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()): print(os.path.basename(d), k, ' '.join('%.4g' % x for x in v[:8]), file=out)
out.close()
print(open(O + '/summary.txt').read())
PY
