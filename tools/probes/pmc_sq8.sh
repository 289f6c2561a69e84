cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq8 -- python $R/tools/storage_probe.py --nqs 4 > $R/gpurun_out/pmc_sq8.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv,glob,collections,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_sq8/*/*counter_collection.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'sq8' in r['Kernel_Name'] and 'sweep' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, len(v), sum(v)/len(v))
PY
