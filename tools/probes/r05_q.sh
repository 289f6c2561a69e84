export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05q
mkdir -p $O
cd $R
HL="--steps 20 --warmup 3 --no-cpu-baseline --check-queries 64 --no-tiles --no-hnsw --no-sq8-leg --no-sharded-leg --no-traffic-pass --no-latency-legs"
for v in ${VARIANTS:-base alignfirst base alignfirst}; do
  L=""; [ $v != base ] && L="--lib tools/probes/out/libvelesdb_hip_$v.so"
  timeout 900 python bench.py $HL $L > $O/hl_$v.json 2> $O/hl_$v.err; echo "$v rc=$?"
  python - <<PY
import json
l=json.loads(open("$O/hl_$v.json").read().strip().splitlines()[-1])
f=json.load(open("bench_legs.json"))
print("$v", "value", l["value"], "ms_per_step", l["ms_per_step"], "frac", l["roofline"]["frac"], "kernel_ms", l["roofline"]["kernel_ms"], "parity", l["parity_check"], "bf16:", l["legs"].get("bf16_gemm"))
for m in f.get("other_metrics", []):
    b=m.get("batch") or {}
    print("   ", m["metric"], "batch ms_per_call", b.get("ms_per_call"), "kernel", b.get("sweep_kernel_ms"), "parity", m.get("parity_check"))
PY
done
