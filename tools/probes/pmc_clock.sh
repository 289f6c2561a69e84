# average shader clock during the selection kernels: GRBM_GUI_ACTIVE / dispatch duration (own --pmc pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_clock
mkdir -p $O
for BD in 0 1; do
VELESDB_G16_BDIR=$BD timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/bf16_$BD -- python $R/tools/probes/bf16_glds_probe.py --rows 4000000 --reps 3 > $O/bf16_$BD.log 2>&1; echo rc=$?
VELESDB_G16_BDIR=$BD timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/split_$BD -- python $R/tools/probes/split_probe.py --reps 3 > $O/split_$BD.log 2>&1; echo rc=$?
done
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_clock'
out=open(O+'/summary.txt','w')
for d in sorted(glob.glob(O+'/*_[01]')):
    fs=glob.glob(d+'/*/*counter_collection.csv')
    if not fs: continue
    rows=list(csv.DictReader(open(fs[0])))
    disp=collections.defaultdict(dict)
    for r in rows:
        k=(r['Dispatch_Id'])
        disp[k]['name']=r['Kernel_Name'][:60]
        disp[k][r['Counter_Name']]=float(r['Counter_Value'])
        if 'Start_Timestamp' in r: disp[k]['dur']=(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
    agg=collections.defaultdict(list)
    for k,v in disp.items():
        if 'gemm' in v['name'] and v.get('dur',0)>2e5: agg[v['name']].append(v)
    for n,vs in agg.items():
        for v in vs[-3:]:
            g=v.get('GRBM_GUI_ACTIVE',0); dur=v.get('dur',0)
            print(os.path.basename(d), n, 'dur_us=%.1f'%(dur/1e3), 'GUI_ACTIVE=%.4g'%g, 'clk_GHz(per XCD avg, /8)=%.3f'%(g/8/dur if dur else 0), 'clk_GHz(raw)=%.3f'%(g/dur if dur else 0), 'MFMA_BUSY=%.4g'%v.get('SQ_VALU_MFMA_BUSY_CYCLES',0), 'SQ_BUSY=%.4g'%v.get('SQ_BUSY_CYCLES',0), 'WAVE_CYC=%.4g'%v.get('SQ_WAVE_CYCLES',0), file=out)
out.close(); print(open(O+'/summary.txt').read())
PY
head -3 $O/bf16_0/*/*counter_collection.csv | cut -c1-400
