F="--steps 20 --warmup 5 --no-hnsw --no-metrics-leg --no-bf16-leg --no-sq8-leg --no-traffic-pass --no-latency-legs --no-sharded-leg --no-m128-leg --no-cpu-baseline --no-tiles --check-queries 0"
for i in 1 2 3; do python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['repeat_ms_per_step'], d['roofline']['frac'])" >> gpurun_out/r06af.log; done
cat gpurun_out/r06af.log
