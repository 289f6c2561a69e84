cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sk in 0 1; do
  echo "skip=$sk"; VELESDB_BITS_FUSED_SKIP=$sk VDB_PROBE_LIB=1 python $R/tools/probes/bits_single_query_probe.py hamming 2>&1 | grep -v amdgpu.ids | head -1
done
python $R/tools/probes/bits_single_query_probe.py jaccard 2>&1 | grep -v amdgpu.ids
cd $R; timeout 900 python -m pytest tests/test_gpu_round5_parity.py -k one_launch -x -q 2>&1 | tail -2
