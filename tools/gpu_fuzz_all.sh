#!/bin/bash
# every fuzzer against the oracle, a few minutes each, fresh seeds: TAG names the output directory under gpurun_out/
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04_fuzz}
S=${SEED0:-400}
mkdir -p $O
run() {  # name seconds args...
  n=$1; t=$2; shift 2
  timeout $((t + 200)) python "$@" --seconds $t --seed $S 2>&1 | grep -v amdgpu.ids | tail -2 > $O/$n.log
  echo "== $n (seed $S): $(tail -1 $O/$n.log)"
  S=$((S + 1))
}
run fuzz_index 120 tools/fuzz_index.py
run fuzz_misc 100 tools/fuzz_misc.py
run fuzz_hnsw 120 tools/fuzz_hnsw.py
run fuzz_hnsw_big 120 tools/fuzz_hnsw.py --big
run fuzz_sweep 100 tools/fuzz_sweep.py
run fuzz_sweep_valu 60 tools/fuzz_sweep.py --engine 0
run fuzz_sweep_big 90 tools/fuzz_sweep.py --big
run fuzz_sweep_euclid 80 tools/fuzz_sweep.py --euclid
run fuzz_sweep_select 150 tools/fuzz_sweep.py --select
run fuzz_sweep_wide 150 tools/fuzz_sweep.py --wide
run fuzz_sweep_bf16_big 80 tools/fuzz_sweep.py --bf16-big
run fuzz_sweep_bits 60 tools/fuzz_sweep.py --bits
run fuzz_sweep_bits_big 100 tools/fuzz_sweep.py --bits-big
run fuzz_storage 100 tools/fuzz_storage.py
run fuzz_storage_select 120 tools/fuzz_storage.py --select
