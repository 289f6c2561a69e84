export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
P=tools/probes/bf16_glds_probe.py
cmpf() { python - "$1" "$2" <<'PY'
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
print("ids equal:", bool(np.array_equal(a["ids"], b["ids"])), " score bits equal:", bool(np.array_equal(a["sc"], b["sc"])), " cnt equal:", bool(np.array_equal(a["cnt"], b["cnt"])),
      " differing queries:", int(np.sum(np.any(a["ids"] != b["ids"], axis=1))))
PY
}
for cfg in "--rows 300000 --nq 1024" "--rows 700001 --nq 900 --metric dot" "--rows 200000 --nq 256 --dead 7"; do
  echo "== $cfg"
  VELESDB_BF16_GLDS=1 timeout 120 python $P $cfg --reps 2 --save /tmp/new.npz 2>&1 | grep -v amdgpu.ids
  VELESDB_BF16_GLDS=0 timeout 120 python $P $cfg --reps 2 --save /tmp/old.npz 2>&1 | grep -v amdgpu.ids
  cmpf /tmp/new.npz /tmp/old.npz
done 2>&1 | tee $O/ab_correctness.log
for v in base noepi noload+noepi; do
  if [ $v = base ]; then L=""; else L=tools/probes/out/libvelesdb_hip_g16_$v.so; fi
  echo "== variant $v"
  VELESDB_HIP_LIB=$L timeout 120 python $P --rows 4000000 --reps 5 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/ablation_4m.log
VELESDB_BF16_GLDS=1 timeout 200 python $P --rows 10000000 --reps 5 2>&1 | grep -v amdgpu.ids | tee $O/perf_10m.log
