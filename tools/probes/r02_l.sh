export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_split.py -q --durations=5 2>&1 | tail -15 | tee $O/pytest_split.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/tools/probes/split_probe.py --reps 10 > $O/split_prof.log 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02l/prof/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:18]: print(f"{r['Name'][:95]:95s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
find $O -name "*_kernel_trace.csv" -delete
