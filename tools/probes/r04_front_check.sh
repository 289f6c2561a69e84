#!/bin/bash
# after moving the combining front's protocol and the handle lock into HIP-free headers: the GPU side of the same front
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04front}
mkdir -p $O
timeout 230 python -m pytest tests/test_gpu_callers.py tests/test_gpu_hardening.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.log
