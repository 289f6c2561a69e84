# round-2 first GPU call: headline-size parity tests + HBM traffic of the bf16 GEMM sweep at the configs[3] size
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_headline_sizes.py -x -q -s --durations=5 > $O/pytest_headline.log 2>&1; echo "pytest rc=$?" >> $O/pytest_headline.log
tail -15 $O/pytest_headline.log
nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_bf16 -- python $R/tools/bf16_probe.py --rows 10000000 > $O/pmc_bf16.log 2>&1
echo pmc rc=$?
tail -4 $O/pmc_bf16.log
python3 - <<'PY'
import csv,glob,collections,os
fs=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02a/pmc_bf16/*/*counter_collection.csv')
acc=collections.defaultdict(list)
for f in fs:
  for r in csv.DictReader(open(f)):
    if 'gemm' in r['Kernel_Name']: acc[r['Kernel_Name'][:90]].append(float(r['Counter_Value'])*2048)
for k,v in acc.items(): print(k, len(v), sum(v)/len(v)/1e9, 'GB (x2 corrected)')
PY
find $O -name "*_kernel_trace.csv" -delete
