export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02u
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/tools/probes/split_probe.py --reps 5 > $O/probe_rocprof.log 2>&1
find $O -name "*_kernel_trace.csv" | head -1 | xargs -I{} python3 - {} <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
# per-kernel durations in launch order for the LAST level-2 batch region: print sequence of the first 40 kernels after warmup
seq=[(r['Kernel_Name'][:70],(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows]
# find first glds<0,false> occurrences
idx=[i for i,(n,d) in enumerate(seq) if 'gemm_bf16_glds<0, false>' in n]
print('bf16 glds launches:',len(idx))
if idx:
    i0=idx[-2]  # first launch of the last level-2 batch
    for n,d in seq[i0-14:i0+12]: print('%9.1f us  %s'%(d,n))
PY
find $O -name "*_kernel_trace.csv" -delete
cd $R
timeout 600 python -m pytest tests/test_gpu_split.py -x -q -k "level2 or near_dup" 2>&1 | tail -5 | tee $O/pytest.log
