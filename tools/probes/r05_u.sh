cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05u
mkdir -p $O
for b in 128 256; do
  echo "blocks=$b"; VELESDB_BITS_FUSED_BLOCKS=$b VDB_PROBE_LIB=1 python $R/tools/probes/bits_single_query_probe.py hamming 2>&1 | grep -v amdgpu.ids | head -2
done
echo "jaccard default"; python $R/tools/probes/bits_single_query_probe.py jaccard 2>&1 | grep -v amdgpu.ids
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o one -- python $R/tools/probes/bits_single_query_probe.py hamming > $O/prof.log 2>&1
python3 - <<'PY'
import csv,glob,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r05u'
for f in glob.glob(O+'/prof/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
cd $R
timeout 1500 python -m pytest tests/test_gpu_round5_parity.py -k one_launch -x -q 2>&1 | tail -3
for f in "--bits"; do timeout 200 python tools/fuzz_sweep.py $f --seconds 40 --seed 581 2>&1 | grep -v amdgpu.ids | tail -1; done
