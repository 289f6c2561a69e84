set -x
timeout 900 python -m pytest tests/test_gpu_wide_k.py -x -q 2>&1 | tail -30 > gpurun_out/r06k_tests.log
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_sweep.py -x -q 2>&1 | tail -5 >> gpurun_out/r06k_tests.log
F="--steps 20 --warmup 5 --no-hnsw --no-metrics-leg --no-bf16-leg --no-sq8-leg --no-traffic-pass --no-latency-legs --no-sharded-leg --no-m128-leg --no-cpu-baseline"
python bench.py $F > gpurun_out/r06k_bench.json 2> gpurun_out/r06k_bench.err
cp bench_legs.json gpurun_out/r06k_bench_legs.json
python - <<'PY' > gpurun_out/r06k_euclid_k.log 2>&1
import numpy as np, time, torch
import velesdb_amd as va
dev=torch.device("cuda",0)
g=torch.Generator(device=dev); g.manual_seed(42)
N,D,Q=1_000_000,768,1024
rows=torch.randn((N,D),generator=g,device=dev)
g.manual_seed(43); qs=torch.randn((Q,D),generator=g,device=dev)
ix=va.HnswIndex(D,va.DistanceMetric.Euclidean,va.HnswParams(32,400,N))
st=torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize(); ix.upload_dev(0,rows.data_ptr(),N,st); torch.cuda.synchronize()
for k in (10,50,100):
    ids=torch.empty((Q,k),dtype=torch.int64,device=dev); sc=torch.empty((Q,k),dtype=torch.float32,device=dev); n=torch.empty((Q,),dtype=torch.int32,device=dev)
    for _ in range(3): ix.search_batch_dev(qs.data_ptr(),Q,k,0,va.MODE_BRUTE,ids.data_ptr(),sc.data_ptr(),n.data_ptr(),st)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): ix.search_batch_dev(qs.data_ptr(),Q,k,0,va.MODE_BRUTE,ids.data_ptr(),sc.data_ptr(),n.data_ptr(),st)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print("euclidean k",k,"ms",round(dt*1e3,4),"qps",round(Q/dt,1),"level",ix.last_select_level(),"unproven",ix.last_split_stats())
PY
cat gpurun_out/r06k_tests.log gpurun_out/r06k_euclid_k.log
