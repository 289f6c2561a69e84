#!/bin/bash
# full check of a commit: every -m gpu test, the bench line, the smoke entry
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03t
mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 | tee $O/pytest_all.log
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench_err.log; echo "bench rc=$?"; tail -3 $O/bench_err.log
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03t/bench_line.json'))
r = d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'frac_step', d.get('frac_step'), 'traffic/alg', r.get('traffic_over_algorithmic'))
print('bf16', d['bf16_gemm']['qps'], d['bf16_gemm']['roofline']['frac'], d['bf16_gemm']['roofline'].get('traffic_over_algorithmic'), d['bf16_gemm']['parity_check']['ok'])
print('hnsw', d['hnsw']['qps'], d['hnsw']['roofline']['frac'], [x['median_us_per_call'] for x in d['hnsw']['latency_mode']][:3])
print('cfg0', d['config0_10k']['search_median_us'])
PY
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
