#!/usr/bin/env python3
"""Why does the traversal leg move between 167 K and 192 K q/s from run to run?  One graph, the same 8 192 queries:
traversal launches cold, right behind a burst of matrix-core batches (what bench.py's leg order does), and after a pause —
per-launch HIP-event times here, per-dispatch GRBM_GUI_ACTIVE (= shader clock) when run under rocprofv3 --pmc."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import velesdb_amd as va
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
N, D, K, NQ, EF = 1_000_000, 768, 10, 8192, 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(42)
ix = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(32, 400, N))
st = torch.cuda.current_stream().cuda_stream
c = torch.randn((N, D), generator=g, device=dev); torch.cuda.synchronize()
ix.upload_dev(0, c.data_ptr(), N, st); del c
ix.build_graph(0)
g.manual_seed(43)
q = torch.randn((NQ, D), generator=g, device=dev)
ids = torch.empty((NQ, K), dtype=torch.int64, device=dev); sc = torch.empty((NQ, K), dtype=torch.float32, device=dev); n = torch.empty((NQ,), dtype=torch.int32, device=dev)
def trav(tag, reps):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ix.search_batch_dev(q.data_ptr(), NQ, K, EF, va.MODE_HNSW, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
    print(f"{tag}: " + " ".join(f"{x:.2f}" for x in out) + f" ms  -> {NQ / (min(out) * 1e-3):.0f} .. {NQ / (max(out) * 1e-3):.0f} q/s", flush=True)
trav("cold (first launches of the process)", 6)
for _ in range(60):  # ~0.2 s of matrix-core batches
    ix.search_batch_dev(q.data_ptr(), 1024, K, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
trav("right behind 60 matrix-core batches", 6)
time.sleep(5)
trav("after a 5 s pause", 6)
if os.environ.get("VDB_SPREAD_LONG"):  # (not under rocprofv3 --pmc: its counter buffers do not survive ~30 K dispatches)
    for _ in range(600):  # ~2 s of matrix-core batches
        ix.search_batch_dev(q.data_ptr(), 1024, K, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
    trav("right behind 600 matrix-core batches", 6)
