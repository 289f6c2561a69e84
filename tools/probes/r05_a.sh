set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
timeout 300 python tools/probes/subnormal_mfma_probe.py > $O/subnormal.log 2>&1; echo "subnormal rc=$?"
tail -30 $O/subnormal.log
timeout 1200 python tools/probes/shard_size_other_paths.py > $O/shard_other.log 2>&1; echo "shard_other rc=$?"
tail -15 $O/shard_other.log
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
wc -c $O/bench_line.json; cat $O/bench_line.json
tail -5 $O/bench.err
cp bench_legs.json $O/ 2>/dev/null
