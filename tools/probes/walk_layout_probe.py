"""Does the f32 throughput walk's time depend on WHERE the process's buffers landed?  (bench.py's hnsw leg: 43.8 ms per 8 192 queries in
some processes, 49.2 ms in others on the same box, same kernel, same counters.)  Each invocation is one process: optional dummy
allocations first (PRE = comma-separated MiB, kept; FREE = MiB allocated and released before the index exists), then index, graph, walk.
usage: PRE=.. FREE=.. walk_layout_probe.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import velesdb_amd as va
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D, NQ = 768, 8192
dev = torch.device("cuda", 0)
keep = [torch.empty(int(float(m) * (1 << 20)), dtype=torch.uint8, device=dev) for m in os.environ.get("PRE", "").split(",") if m]
fr = os.environ.get("FREE", "")
if fr:
    t = torch.empty(int(float(fr) * (1 << 20)), dtype=torch.uint8, device=dev); del t; torch.cuda.empty_cache()
g = torch.Generator(device=dev); g.manual_seed(42)
ix = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(32, 400, N))
st = torch.cuda.current_stream().cuda_stream
c = torch.randn((N, D), generator=g, device=dev); torch.cuda.synchronize()
ix.upload_dev(0, c.data_ptr(), N, st); del c
if os.environ.get("EMPTY_CACHE"): torch.cuda.empty_cache()
ix.build_graph(0)
g.manual_seed(43)
q = torch.randn((NQ, D), generator=g, device=dev)
ids = torch.empty((NQ, 10), dtype=torch.int64, device=dev); sc = torch.empty((NQ, 10), dtype=torch.float32, device=dev); n = torch.empty((NQ,), dtype=torch.int32, device=dev)
for _ in range(2):
    ix.search_batch_dev(q.data_ptr(), NQ, 10, 128, va.MODE_HNSW, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
torch.cuda.synchronize()
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    ix.search_batch_dev(q.data_ptr(), NQ, 10, 128, va.MODE_HNSW, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"PRE={os.environ.get('PRE','')} FREE={fr} EMPTY_CACHE={os.environ.get('EMPTY_CACHE','')}: walk {min(ts):.2f} ms (runs {' '.join('%.2f' % x for x in ts)}) = {NQ / min(ts) * 1e3:.0f} q/s; "
      f"torch reserved {torch.cuda.memory_reserved() >> 20} MiB", flush=True)
