# LDS bank conflicts of the GEMM-structured sweep (separate --pmc pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_lds
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_lds/f32 -- python $R/tools/sweep_probe.py --nqs 1024 > $R/gpurun_out/pmc_lds/f32.log 2>&1
echo rc=$?
find $R/gpurun_out/pmc_lds -name "*_kernel_trace.csv" -delete
python3 - <<'PY'
import csv,glob,collections,os
fs=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_lds/f32/*/*counter_collection.csv')
acc=collections.defaultdict(lambda: [0,0.0])
for r in csv.DictReader(open(fs[0])):
    if 'gemm' not in r['Kernel_Name']: continue
    k=(r['Kernel_Name'][:70],r['Counter_Name'])
    acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
for k,v in sorted(acc.items()): print(k, v[0], v[1]/v[0])
PY
