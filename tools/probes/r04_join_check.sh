#!/bin/bash
# bench.py's sharded leg with torch's RCCL group initialised at world 1: the normal join, and a join that cannot succeed (the
# library pointed at a transport that does not exist) — the line must still be printed, with the leg's error in it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${TAG:-r04join}
mkdir -p $O
F="--steps 10 --warmup 2 --no-cpu-baseline --check-queries 0 --no-tiles --no-hnsw --no-bf16-leg --no-sq8-leg --no-metrics-leg --no-traffic-pass --dist-single"
timeout 150 python bench.py $F > $O/line_ok.json 2> $O/ok.err; echo "rc=$?"
VELESDB_RCCL_LIB=/nonexistent/librccl.so timeout 150 python bench.py $F > $O/line_fail.json 2> $O/fail.err; echo "rc=$?"
python - <<PY
import json
for n in ("line_ok","line_fail"):
    try:
        d=json.loads(open("$O/%s.json"%n).read().strip().splitlines()[-1])
        s=d["sharded"]
        print(n, d["value"], {k:s.get(k) for k in ("group_ok","error","join_error","qps","transport","equals_unsharded_bitwise")})
    except Exception as e:
        print(n, "no line:", e)
PY
grep -h "joining\|VELESDB_RCCL_LIB" $O/fail.err | head -3
