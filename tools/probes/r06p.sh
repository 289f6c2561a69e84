set -x
python - <<'PY' > gpurun_out/r06p_euclid.log 2>&1
import numpy as np, time, torch
import velesdb_amd as va
from oracle import pyoracle as po
dev=torch.device("cuda",0)
g=torch.Generator(device=dev); g.manual_seed(42)
N,D,Q=1_000_000,768,1024
rows=torch.randn((N,D),generator=g,device=dev)
g.manual_seed(43); qs=torch.randn((Q,D),generator=g,device=dev)
ix=va.HnswIndex(D,va.DistanceMetric.Euclidean,va.HnswParams(32,400,N))
st=torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize(); ix.upload_dev(0,rows.data_ptr(),N,st); torch.cuda.synchronize()
for sel in (3,2,3,2):
  va.set_split_selector(sel)
  for k in (1,10):
    ids=torch.empty((Q,k),dtype=torch.int64,device=dev); sc=torch.empty((Q,k),dtype=torch.float32,device=dev); n=torch.empty((Q,),dtype=torch.int32,device=dev)
    for _ in range(3): ix.search_batch_dev(qs.data_ptr(),Q,k,0,va.MODE_BRUTE,ids.data_ptr(),sc.data_ptr(),n.data_ptr(),st)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): ix.search_batch_dev(qs.data_ptr(),Q,k,0,va.MODE_BRUTE,ids.data_ptr(),sc.data_ptr(),n.data_ptr(),st)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print("selector",sel,"euclidean k",k,"ms",round(dt*1e3,4),"qps",round(Q/dt,1),"level",ix.last_select_level(),"unproven",ix.last_split_stats(),flush=True)
PY
timeout 900 python -m pytest tests/test_gpu_split.py tests/test_gpu_sweep.py -x -q -k "euclid" 2>&1 | tail -5 >> gpurun_out/r06p_euclid.log
timeout 600 python tools/fuzz_sweep.py --euclid --seconds 60 --seed 77 2>&1 | tail -2 >> gpurun_out/r06p_euclid.log
cat gpurun_out/r06p_euclid.log
