# where do the waves of the bf16 / split selection kernel spend their cycles?  (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_stall
mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|GRBM_[A-Z0-9_]*" | sort -u > $O/avail.txt
wc -l $O/avail.txt
run() { # name, counters...
  n=$1; shift
  VELESDB_G16_BDIR=${BD:-0} timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python $R/tools/probes/bf16_glds_probe.py --rows 2000000 --reps 3 > $O/$n.log 2>&1
  echo "$n rc=$?"
  find $O/$n -name "*_kernel_trace.csv" -delete
}
for BD in 0 1; do
export BD
run p1_$BD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run p2_$BD SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
run p3_$BD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_MOPS_BF16
run p4_$BD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
run p5_$BD TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr
done
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_stall'
out=open(O+'/summary.txt','w')
for d in sorted(glob.glob(O+'/p*_*')):
    if not os.path.isdir(d): continue
    fs=glob.glob(d+'/*/*counter_collection.csv')
    if not fs:
        print(os.path.basename(d),'no counters', file=out); continue
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        if 'gemm_bf16_glds' not in r['Kernel_Name']: continue
        k=r['Counter_Name']; acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
    for k,v in sorted(acc.items()): print(os.path.basename(d), k, v[0], '%.4g'%(v[1]/v[0]), file=out)
out.close()
print(open(O+'/summary.txt').read())
PY
