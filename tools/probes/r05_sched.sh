export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05sched
mkdir -p $O
cd $R
HL="--steps 20 --warmup 3 --no-cpu-baseline --check-queries 64 --no-tiles --no-hnsw --no-sq8-leg --no-sharded-leg --no-traffic-pass --no-latency-legs --no-metrics-leg --no-bf16-leg --lib velesdb_amd/lib/libvelesdb_hip_probe.so"
for sch in "1,4,16" "4,16,0" "2,8,0" "1,8,0" "3,12,0" "6,0,0" "1,4,16" "2,6,18"; do
  VELESDB_SEL_STEPS=$sch timeout 600 python bench.py $HL > $O/hl.json 2> $O/hl.err; rc=$?
  python - <<PY
import json
l=json.loads(open("$O/hl.json").read().strip().splitlines()[-1])
print("steps=$sch rc=$rc", "value", l["value"], "ms_per_step", l["ms_per_step"], "frac", l["roofline"]["frac"], "kernel_ms", l["roofline"]["kernel_ms"], "launches", l["roofline"].get("launches_timed"), "parity", l["parity_check"])
PY
done
