# MFMA utilisation of the two GEMM-structured kernels (separate --pmc pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_mfma
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma/f32 -- python $R/tools/sweep_probe.py --nqs 1024 > $R/gpurun_out/pmc_mfma/f32.log 2>&1
echo f32 rc=$?
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma/bf16 -- python $R/tools/bf16_probe.py --rows 4000000 > $R/gpurun_out/pmc_mfma/bf16.log 2>&1
echo bf16 rc=$?
find $R/gpurun_out/pmc_mfma -name "*_kernel_trace.csv" -delete
python3 - <<'PY'
import csv,glob,collections,os
for leg in ('f32','bf16'):
    fs=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_mfma/'+leg+'/*/*counter_collection.csv')
    if not fs: print(leg,'no counter file'); continue
    acc=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(fs[0])):
        if 'gemm' not in r['Kernel_Name']: continue
        k=(r['Kernel_Name'][:70],r['Grid_Size'],r['Counter_Name'])
        acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
    for k,v in sorted(acc.items()): print(leg,k, v[0], v[1]/v[0])
PY
