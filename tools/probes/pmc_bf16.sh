cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 800 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_bf16 -- python $R/tools/bf16_probe.py --rows 4000000 > $R/gpurun_out/pmc_bf16.log 2>&1
echo rc=$?
tail -3 $R/gpurun_out/pmc_bf16.log
python3 - <<'PY'
import csv,glob,collections,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_bf16/*/*counter_collection.csv')[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm' in r['Kernel_Name']: acc[r['Kernel_Name'][:70]].append(float(r['Counter_Value'])*2048)
for k,v in acc.items(): print(k, len(v), sum(v)/len(v)/1e9, 'GB')
PY
