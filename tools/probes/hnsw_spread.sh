cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hnsw_spread
mkdir -p $O
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- python $R/tools/probes/hnsw_spread.py 2>&1 | grep -v "amdgpu.ids\|^[EWI]2026" | tee $O/run.log
python3 - <<'PY' | tee -a $O/run.log
import csv,glob,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/hnsw_spread'
f=glob.glob(O+'/pmc/*/*counter_collection.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'hnsw_search_kernel' in r['Kernel_Name'] and r['Counter_Name']=='GRBM_GUI_ACTIVE']
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print('traversal dispatches:',len(rows))
for i,r in enumerate(rows):
    dur=int(r['End_Timestamp'])-int(r['Start_Timestamp']); g=float(r['Counter_Value'])
    print('%2d  %8.2f ms  clock %.3f GHz'%(i, dur/1e6, g/8/dur))
PY
find $O -name "*_kernel_trace.csv" -delete
cd $R; echo "== the same process without the profiler"; VDB_SPREAD_LONG=1 timeout 600 python tools/probes/hnsw_spread.py 2>&1 | grep -v amdgpu.ids | tee -a $O/run.log; (rocm-smi --showclocks --showpower 2>/dev/null | head -30) | tee -a $O/run.log
