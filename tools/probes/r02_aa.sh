export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02aa
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_int8.py tests/test_gpu_sweep.py -x -q -k "int8 or bits or hamming or jaccard or Hamming or Jaccard" 2>&1 | tail -5 | tee $O/pytest.log
for w in 1 0; do
  echo "== VELESDB_I8_WAVES2=$w" | tee -a $O/int8_probe.log
  VELESDB_I8_WAVES2=$w timeout 600 python tools/probes/int8_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/int8_probe.log
done
timeout 300 python tools/storage_probe.py --metric cosine --nqs 1,8 2>&1 | grep -v amdgpu.ids | tee $O/storage_probe.log
timeout 300 python tools/sweep_probe.py --metric hamming --nqs 1 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/bits_probe.log
