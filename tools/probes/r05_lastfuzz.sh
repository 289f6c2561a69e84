cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05lastfuzz; mkdir -p $O
run() { n=$1; t=$2; shift 2; timeout $((t + 200)) python "$@" --seconds $t --seed $S 2>&1 | grep -v amdgpu.ids | tail -1 > $O/$n.log; echo "== $n (seed $S): $(tail -1 $O/$n.log)"; S=$((S + 1)); }
S=800
run fuzz_sweep_bits 200 tools/fuzz_sweep.py --bits
run fuzz_sweep 200 tools/fuzz_sweep.py
run fuzz_sweep_select 200 tools/fuzz_sweep.py --select
run fuzz_sweep_euclid 120 tools/fuzz_sweep.py --euclid
run fuzz_storage 150 tools/fuzz_storage.py
run fuzz_index 100 tools/fuzz_index.py
