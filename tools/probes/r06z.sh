# round 6, last tree: matrix-pipe and LDS counters of the headline selection kernel (the WIDE instance), separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06z_pmc
mkdir -p $O
HL="--steps 10 --warmup 2 --settle-steps 0 --no-cpu-baseline --check-queries 0 --no-tiles --no-hnsw --no-sq8-leg --no-sharded-leg --no-traffic-pass --no-latency-legs --no-bf16-leg --no-metrics-leg --no-m128-leg"
run() { n=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python $R/bench.py $HL > $O/$n.log 2>&1
  echo "$n rc=$?"; find $O/$n -name "*_kernel_trace.csv" -delete; }
run p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
run p3 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r06z_pmc'
out=open(O+'/summary.txt','w')
for d in sorted(glob.glob(O+'/p*')):
    if not os.path.isdir(d): continue
    fs=glob.glob(d+'/*/*counter_collection.csv')
    if not fs:
        print(os.path.basename(d),'no counters:', open(d+'.log').read()[-300:].replace('\n',' | '), file=out); continue
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        kn=r['Kernel_Name']
        if 'gemm_bf16_pp' not in kn: continue
        fam=kn[kn.index('gemm_bf16_pp'):][:34]
        k=(fam,r['Counter_Name']); acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
    for k,v in sorted(acc.items()): print(os.path.basename(d), k[0], k[1], 'launches', v[0], 'per launch %.5g'%(v[1]/v[0]), file=out)
out.close()
print(open(O+'/summary.txt').read())
PY
cp $O/summary.txt $R/gpurun_out/r06z_pmc_summary.txt
find $O -name "*.csv" -delete
