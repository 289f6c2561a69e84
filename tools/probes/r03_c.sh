#!/bin/bash
# round 3: is the bf16 kernel power-limited?  clock + MFMA-busy + wave-state counters for the ping-pong (PP=1) and the
# lock-step (PP=0) kernel, and board power / sclk sampled by rocm-smi during a sustained run of each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c
mkdir -p $O
for PP in 1 0; do
  export VELESDB_BF16_PP=$PP
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/clk_$PP -- python $R/tools/probes/bf16_glds_probe.py --rows 4000000 --reps 3 > $O/clk_$PP.log 2>&1; echo "clk_$PP rc=$?"
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/stall_$PP -- python $R/tools/probes/bf16_glds_probe.py --rows 4000000 --reps 3 > $O/stall_$PP.log 2>&1; echo "stall_$PP rc=$?"
  find $O -name "*_kernel_trace.csv" -delete
  # sustained run with a power / clock sampler beside it
  ( for i in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.25; done ) > $O/smi_$PP.log 2>&1 &
  SMI=$!
  timeout 300 python $R/tools/probes/bf16_glds_probe.py --rows 4000000 --reps 600 2>&1 | tail -1 > $O/sustained_$PP.log
  kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
  cat $O/sustained_$PP.log
done
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r03c'
out=open(O+'/summary.txt','w')
for d in sorted(glob.glob(O+'/*_[01]')):
    if not os.path.isdir(d): continue
    fs=glob.glob(d+'/*/*counter_collection.csv')
    if not fs: continue
    disp=collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[0])):
        k=r['Dispatch_Id']
        disp[k]['name']=r['Kernel_Name'][:48]
        disp[k][r['Counter_Name']]=float(r['Counter_Value'])
        if 'Start_Timestamp' in r: disp[k]['dur']=float(r['End_Timestamp'])-float(r['Start_Timestamp'])
    for k,v in disp.items():
        if 'gemm_bf16' in v['name'] and v.get('dur',0)>2e6:
            g=v.get('GRBM_GUI_ACTIVE',0); dur=v['dur']
            s=' '.join('%s=%.4g'%(c,x) for c,x in sorted(v.items()) if c not in('name','dur'))
            extra=''
            if g: extra=' clk_GHz=%.3f mfma_busy_frac=%.3f'%(g/8/dur, v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(g/8*1024))
            if 'SQ_WAIT_ANY' in v: extra=' wait_any=%.3f wait_inst=%.3f active=%.3f wait_inst_lds=%.3f'%tuple(v.get(c,0)/v['SQ_WAVE_CYCLES'] for c in('SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_WAIT_INST_LDS'))
            print(os.path.basename(d), v['name'], 'dur_us=%.1f'%(dur/1e3), s, extra, file=out)
out.close(); print(open(O+'/summary.txt').read())
PY
for PP in 1 0; do echo "== smi PP=$PP"; sed -n '8,40p' $O/smi_$PP.log | cut -c1-300 | sort | uniq -c | sort -rn | head -8; done
