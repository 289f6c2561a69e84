#!/bin/bash
# the fuzzers on the final code of round 3
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03fuzz
mkdir -p $O
run() { n=$1; shift; timeout 500 "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$n: /" | tee -a $O/fuzz_final.log; }
run sweep_select   python tools/fuzz_sweep.py --select --seconds 300 --seed 201
run sweep_euclid   python tools/fuzz_sweep.py --euclid --seconds 120 --seed 202
run sweep_bf16_big python tools/fuzz_sweep.py --bf16-big --seconds 150 --seed 203
run sweep_default  python tools/fuzz_sweep.py --seconds 120 --seed 204
run storage_select python tools/fuzz_storage.py --select --seconds 200 --seed 205
run storage        python tools/fuzz_storage.py --seconds 100 --seed 206
run hnsw           python tools/fuzz_hnsw.py --seconds 150 --seed 207
VELESDB_HNSW_LATENCY_MODE=2 run hnsw_lat_spec python tools/fuzz_hnsw.py --seconds 100 --seed 208
VELESDB_HNSW_LATENCY_MODE=3 run hnsw_lat_nospec python tools/fuzz_hnsw.py --seconds 100 --seed 209
run hnsw_big       python tools/fuzz_hnsw.py --big --seconds 150 --seed 210
run index          python tools/fuzz_index.py --seconds 200 --seed 211
run misc           python tools/fuzz_misc.py --seconds 100 --seed 212
