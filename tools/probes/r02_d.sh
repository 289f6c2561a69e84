export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02d
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
for cfg in "--rows 300000 --nq 1024" "--rows 700001 --nq 900 --metric dot" "--rows 200000 --nq 256 --dead 7" "--rows 150000 --nq 1000 --dim 128"; do
  echo "== $cfg"
  VELESDB_BF16_GLDS=1 timeout 300 python $P $cfg --reps 2 --save /tmp/new.npz 2>&1 | grep -v amdgpu.ids
  VELESDB_BF16_GLDS=0 timeout 300 python $P $cfg --reps 2 --save /tmp/old.npz 2>&1 | grep -v amdgpu.ids
  cmpf /tmp/new.npz /tmp/old.npz
done 2>&1 | tee $O/ab_correctness.log
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -3 | tee $O/pytest_bf16.log
for v in 1 0; do VELESDB_BF16_GLDS=$v timeout 600 python $P --rows 10000000 --reps 5 2>&1 | grep -v amdgpu.ids; done | tee $O/perf_10m.log
