export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
P=tools/probes/bf16_glds_probe.py
for v in base noepi noload+noepi l2rows l2rows+noepi nob noa; do
  if [ $v = base ]; then L=""; else L=tools/probes/out/libvelesdb_hip_g16_$v.so; fi
  echo "== variant $v"
  VELESDB_HIP_LIB=$L timeout 300 python $P --rows 4000000 --reps 5 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/ablation_4m.log
