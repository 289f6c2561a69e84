#!/bin/bash
# round 3: ping-pong bf16 kernel — parity tests, then A/B against the lock-step kernel (VELESDB_BF16_PP=0)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03b
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03b/pytest.log
cat gpurun_out/r03b/pytest.log
for pp in 1 0; do
  VELESDB_BF16_PP=$pp timeout 600 python tools/probes/bf16_glds_probe.py --rows 4000000 --reps 5 --save gpurun_out/r03b/bf16_pp$pp.npz 2>&1 | tail -2
  VELESDB_BF16_PP=$pp timeout 600 python tools/probes/split_probe.py --reps 10 2>&1 | tail -4
done 2>&1 | tee gpurun_out/r03b/ab.log
python - <<'PY' 2>&1 | tee -a gpurun_out/r03b/ab.log
import numpy as np
a, b = np.load("gpurun_out/r03b/bf16_pp1.npz"), np.load("gpurun_out/r03b/bf16_pp0.npz")
print("bf16 4M: pp vs lock-step ids equal:", np.array_equal(a["ids"], b["ids"]), "score bits equal:", np.array_equal(a["sc"], b["sc"]))
PY
