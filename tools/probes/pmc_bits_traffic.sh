# HBM traffic of a 1 024-query Hamming / Jaccard batch on the FP4 path ("add the FETCH_SIZE pass"): its own --pmc pass,
# kernel trace only; FETCH_SIZE is in KiB and reports half of the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md): x 1024 x 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_bits
mkdir -p $O
for m in hamming jaccard; do
  rm -rf $O/$m
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$m -- python $R/tools/probes/bits_batch_probe.py $m > $O/$m.log 2>&1
  echo "$m rc=$?"
done
python3 - <<'PY'
import csv,glob,os,collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_bits'
out=open(O+'/summary.txt','w')
for m in ('hamming','jaccard'):
    fs=glob.glob(O+'/'+m+'/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(m,'no counters',file=out); continue
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        if r.get('Counter_Name')!='FETCH_SIZE': continue
        n=r['Kernel_Name']
        if not any(t in n for t in ('sweep_topk','merge_topk','seed_scores','bits_','prep_rows','select_finish')): continue
        s=n.split('(')[0].replace('void ','')
        acc[s][0]+=1; acc[s][1]+=float(r['Counter_Value'])*1024.0*2.0
    steps=4
    tot=sum(v[1] for v in acc.values())/steps
    alg=1_000_000*384  # the four-bit image: 768 values x 4 bit per row, read once per batch
    print(f"{m}: {tot/1e6:.1f} MB per 1 024-query batch (4 steps averaged); four-bit image 384 MB => x {tot/alg:.2f}", file=out)
    for k,v in sorted(acc.items(), key=lambda kv:-kv[1][1])[:6]:
        print(f"    {k[:80]:80s} launches {v[0]:4d}  {v[1]/steps/1e6:9.1f} MB per batch", file=out)
out.close()
print(open(O+'/summary.txt').read())
PY
