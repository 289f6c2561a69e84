#!/bin/bash
# the final tree of the round: headline bench line (events) and the rocprofv3 kernel stats of the same command
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-r04final}
mkdir -p $O
cd $R
F="--no-cpu-baseline --no-tiles --no-hnsw --no-bf16-leg --no-sq8-leg --no-metrics-leg --no-sharded-leg --no-traffic-pass"
timeout 60 python bench.py --steps 20 --warmup 3 $F > $O/bench_line_headline.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -- python $R/bench.py --steps 10 --warmup 2 --check-queries 0 $F > $O/bench_under_rocprof_headline.json 2> $O/rocprof_headline.err; echo "rocprof rc=$?"
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*kernel_stats.csv" | head -3
python - <<PY
import json
for n in ("bench_line_headline","bench_under_rocprof_headline"):
    d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d.get("frac_step"))
PY
