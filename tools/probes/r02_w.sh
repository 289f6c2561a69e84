export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02w
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --check-queries 0 --no-tiles --no-hnsw --no-bf16-leg --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass --no-split > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof2 -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --check-queries 0 --no-tiles --no-hnsw --no-bf16-leg --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass > $O/bench2.json 2>$O/bench2.err
F=$(find $O/prof2 -name "*_kernel_trace.csv" | head -1)
echo "trace: $F"
python3 - "$F" > $O/sequence.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'gemm_bf16_glds' in r['Kernel_Name']]
print('glds launches', len(idx))
if idx:
    i0=idx[6]-12
    t0=int(rows[i0]['Start_Timestamp'])
    for r in rows[i0:i0+70]:
        s=(int(r['Start_Timestamp'])-t0)/1e3; d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        print('%10.1f us  +%9.1f us  grid=%s  %s'%(s,d,r.get('Grid_Size',''),r['Kernel_Name'][:80]))
PY
cat $O/sequence.txt | head -90
find $O -name "*_kernel_trace.csv" -delete
tail -1 $O/bench2.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
