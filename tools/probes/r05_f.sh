set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_storage_modes.py tests/test_gpu_round5_parity.py tests/test_gpu_build.py -x -q -k "bit or hamming or jaccard or binary or Hamming or Jaccard or build or packed" > $O/pytest_bits.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest_bits.log
for f in "--bits" "--bits-big"; do
  timeout 300 python tools/fuzz_sweep.py $f --seconds 50 --seed 521 2>&1 | grep -v amdgpu.ids | tail -1
done
timeout 300 python tools/fuzz_storage.py --seconds 50 --seed 522 2>&1 | grep -v amdgpu.ids | tail -1
HL="--steps 20 --warmup 3 --no-cpu-baseline --check-queries 0 --no-tiles --no-sq8-leg --no-sharded-leg --no-traffic-pass --no-latency-legs --no-bf16-leg --no-embedding-leg"
timeout 900 python bench.py $HL > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(json.dumps(l["legs"]))
f=json.load(open("bench_legs.json"))
print(json.dumps(f["hnsw"]["build"]))
for m in f["other_metrics"]: print(m["metric"], json.dumps(m["single_query"]))
PY
cp bench_legs.json $O/
