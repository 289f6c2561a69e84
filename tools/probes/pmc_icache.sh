# does the selection kernel's epilogue miss the instruction cache?  (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_icache
mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_[A-Z0-9_]*\|SQ_IFETCH[A-Z0-9_]*\|SQ_INST_LEVEL[A-Z0-9_]*" | sort -u > $O/avail.txt
cat $O/avail.txt | tr '\n' ' '; echo
HL="--steps 10 --warmup 2 --no-cpu-baseline --check-queries 0 --no-tiles --no-hnsw --no-sq8-leg --no-sharded-leg --no-traffic-pass --no-latency-legs --no-bf16-leg"
run() { n=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python $R/bench.py $HL > $O/$n.log 2>&1
  echo "$n rc=$?"; find $O/$n -name "*_kernel_trace.csv" -delete; }
run p1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run p2 SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
run p3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_icache'
out=open(O+'/summary.txt','w')
for d in sorted(glob.glob(O+'/p*')):
    if not os.path.isdir(d): continue
    fs=glob.glob(d+'/*/*counter_collection.csv')
    if not fs:
        print(os.path.basename(d),'no counters', file=out); continue
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        kn=r['Kernel_Name']
        fam='pp<' if 'gemm_bf16_pp' in kn else None
        if not fam: continue
        fam=kn[kn.index('gemm_bf16_pp'):][:40]
        k=(fam,r['Counter_Name']); acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
    for k,v in sorted(acc.items()): print(os.path.basename(d), k[0], k[1], 'launches', v[0], 'per launch %.5g'%(v[1]/v[0]), file=out)
out.close()
print(open(O+'/summary.txt').read())
PY
