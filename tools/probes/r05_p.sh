export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R
HL="--steps 20 --warmup 3 --no-cpu-baseline --check-queries 64 --no-tiles --no-hnsw --no-sq8-leg --no-sharded-leg --no-traffic-pass --no-latency-legs"
for v in dump nodump dump nodump; do
  L=""; [ $v = nodump ] && L="--lib tools/probes/out/libvelesdb_hip_nodump.so"
  timeout 900 python bench.py $HL $L > $O/hl_$v.json 2> $O/hl_$v.err; echo "$v rc=$?"
  python - <<PY
import json
l=json.loads(open("$O/hl_$v.json").read().strip().splitlines()[-1])
f=json.load(open("bench_legs.json"))
print("$v", "value", l["value"], "ms_per_step", l["ms_per_step"], "frac", l["roofline"]["frac"], "kernel_ms", l["roofline"]["kernel_ms"], "parity", l["parity_check"], "bf16:", l["legs"].get("bf16_gemm"))
for m in f.get("other_metrics", []):
    print("   ", m["metric"], {k: m.get(k) for k in ("ms_per_step", "qps", "kernel_ms", "frac", "parity_check")}, json.dumps(m.get("batch") or m.get("roofline"))[:300])
PY
done
for m in hamming cosine; do
  lib=tools/probes/out/libvelesdb_hip_stamp1.so; [ $m = hamming ] && lib=tools/probes/out/libvelesdb_hip_stamp1h.so
  timeout 300 python tools/probes/pp_stamp_probe.py $lib 1000000 $m > $O/stamp1_$m.log 2>&1; echo "rc=$?"
  grep -v amdgpu.ids $O/stamp1_$m.log | grep "sum of\|epilogue\|look\|row tiles\|^wave"
done
