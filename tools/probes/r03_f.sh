#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03f
timeout 1200 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_split.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r03f/pytest.log
cat gpurun_out/r03f/pytest.log
for pp in 1 0; do
  VELESDB_BF16_PP=$pp timeout 600 python tools/probes/bf16_glds_probe.py --rows 4000000 --reps 5 --save gpurun_out/r03f/bf16_pp$pp.npz 2>&1 | tail -1
  VELESDB_BF16_PP=$pp timeout 600 python tools/probes/split_probe.py --reps 10 2>&1 | grep -E "split=2|identical"
done 2>&1 | tee gpurun_out/r03f/ab.log
python - <<'PY' 2>&1 | tee -a gpurun_out/r03f/ab.log
import numpy as np
a, b = np.load("gpurun_out/r03f/bf16_pp1.npz"), np.load("gpurun_out/r03f/bf16_pp0.npz")
print("bf16 4M: pp vs lock-step ids equal:", np.array_equal(a["ids"], b["ids"]), "score bits equal:", np.array_equal(a["sc"], b["sc"]))
PY
