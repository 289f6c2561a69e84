cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm -- python $R/tools/sweep_probe.py --nqs 128,512 > $R/gpurun_out/pmc_gemm.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv,glob,collections,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_gemm/*/*counter_collection.csv')[0]
acc=collections.defaultdict(lambda: [0,0.0])
for r in csv.DictReader(open(f)):
    if 'gemm' not in r['Kernel_Name']: continue
    k=(r['Kernel_Name'][:60],r['Grid_Size'],r['Counter_Name'])
    acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
for k,v in sorted(acc.items()): print(k, v[0], v[1]/v[0])
PY
