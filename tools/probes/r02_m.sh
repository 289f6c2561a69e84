export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02m
mkdir -p $O
cd /tmp
for v in base noepi noload+noepi; do
  if [ $v = base ]; then L=""; else L=$R/tools/probes/out/libvelesdb_hip_g16_$v.so; fi
  VELESDB_HIP_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python $R/tools/probes/split_probe.py --reps 5 > $O/split_$v.log 2>&1
  echo "== $v"; grep "split=1" $O/split_$v.log
  python3 - $v <<'PY'
import csv,glob,os,sys
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02m/prof_'+sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'bf16_glds' in r['Name']: print(f"{r['Name'][:70]} calls={r['Calls']} avg_us={float(r['AverageNs'])/1e3:.1f} min_us={float(r['MinNs'])/1e3:.1f} max_us={float(r['MaxNs'])/1e3:.1f}")
PY
done 2>&1 | tee $O/summary.log
find $O -name "*_kernel_trace.csv" -delete
