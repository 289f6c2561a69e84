#!/usr/bin/env python3
"""GPU probe for the graph path: batched build time, search throughput / recall / counted bytes for several
batch sizes and ef.  Not part of the product or the test-suite."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])

p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=100_000)
p.add_argument("--dim", type=int, default=768)
p.add_argument("--k", type=int, default=10)
p.add_argument("--M", type=int, default=32)
p.add_argument("--efc", type=int, default=400)
p.add_argument("--max-batch", type=int, default=0)
p.add_argument("--efs", default="64,128,256")
p.add_argument("--nqs", default="1,64,1024,8192")
a = p.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
corpus = torch.randn((a.rows, a.dim), generator=g, device=dev)
g.manual_seed(43)
queries = torch.randn((8192, a.dim), generator=g, device=dev)
ix = va.HnswIndex(a.dim, va.DistanceMetric.Cosine, va.HnswParams(a.M, a.efc, a.rows))
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
ix.upload_dev(0, corpus.data_ptr(), a.rows, st)
del corpus
t0 = time.perf_counter()
ix.build_graph(a.max_batch)
bt = time.perf_counter() - t0
print(f"build {a.rows}x{a.dim} M={a.M} efc={a.efc}: {bt:.2f} s = {a.rows/bt:.0f} inserts/s; graph_info={ix.graph_info()}", flush=True)
qh = queries[:200].cpu().numpy()
gt, _, _ = ix.search_batch_brute_force(qh, a.k)
for ef in [int(x) for x in a.efs.split(",")]:
    res = ix.search_batch_parallel(qh, a.k, va.SearchQuality.Custom(ef))
    nd, ne = ix.last_search_stats()
    rec = np.mean([len({x for x, _ in r} & set(gt[i].tolist())) / a.k for i, r in enumerate(res)])
    bytes_q = (nd * a.dim * 4 + ne * 2 * a.M * 4) / 200
    print(f"ef={ef}: recall@{a.k}={rec:.4f} n_dist/q={nd/200:.0f} n_expand/q={ne/200:.0f} bytes/q={bytes_q/1e6:.2f} MB", flush=True)
    for nq in [int(x) for x in a.nqs.split(",")]:
        ids = torch.empty((nq, a.k), dtype=torch.int64, device=dev)
        sc = torch.empty((nq, a.k), dtype=torch.float32, device=dev)
        n = torch.empty((nq,), dtype=torch.int32, device=dev)
        for _ in range(2):
            ix.search_batch_dev(queries.data_ptr(), nq, a.k, ef, va.MODE_HNSW, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search_batch_dev(queries.data_ptr(), nq, a.k, ef, va.MODE_HNSW, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        nd2, ne2 = ix.last_search_stats()
        gbs = (nd2 * a.dim * 4 + ne2 * 2 * a.M * 4) / dt / 1e9
        print(f"   nq={nq:5d}: {dt*1e3:9.3f} ms/batch  {nq/dt:10.0f} qps   counted {gbs:7.1f} GB/s ({gbs/8000:.3f} of 8 TB/s)", flush=True)
