#!/usr/bin/env python3
"""int8 traversal (VDB_SEARCH_HNSW_INT8) vs f32 traversal on one graph: throughput and the kernel's counters."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import velesdb_amd as va
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=1_000_000)
p.add_argument("--dim", type=int, default=768)
p.add_argument("--nq", type=int, default=8192)
p.add_argument("--ef", type=int, default=128)
a = p.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(42)
ix = va.HnswIndex(a.dim, va.DistanceMetric.Cosine, va.HnswParams(32, 400, a.rows))
st = torch.cuda.current_stream().cuda_stream
c = torch.randn((a.rows, a.dim), generator=g, device=dev); torch.cuda.synchronize()
ix.upload_dev(0, c.data_ptr(), a.rows, st); del c
t0 = time.perf_counter(); ix.build_graph(0); print(f"build {time.perf_counter()-t0:.1f} s", flush=True)
ix.train_quantizer(0)
g.manual_seed(43)
q = torch.randn((a.nq, a.dim), generator=g, device=dev)
ids = torch.empty((a.nq, 10), dtype=torch.int64, device=dev); sc = torch.empty((a.nq, 10), dtype=torch.float32, device=dev); n = torch.empty((a.nq,), dtype=torch.int32, device=dev)
for mode, name in ((va.MODE_HNSW, "f32"), (va.MODE_HNSW_INT8, "int8")):
    for _ in range(2):
        ix.search_batch_dev(q.data_ptr(), a.nq, 10, a.ef, mode, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ix.search_batch_dev(q.data_ptr(), a.nq, 10, a.ef, mode, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    nd, ne = ix.last_search_stats()
    per = (a.dim * 4 if name == "f32" else a.dim + 4)
    alg = nd * per + ne * 64 * 4 + (a.nq * 10 * 4 * a.dim * 4 if name == "int8" else 0)
    ck = int(ids.sum().item()) ^ int(sc.view(torch.int32).to(torch.int64).sum().item())
    print(f"{name}: {dt*1e3:.2f} ms / {a.nq} queries = {a.nq/dt:.0f} q/s; n_dist/q {nd/a.nq:.0f}; {alg/dt/1e9:.0f} GB/s = {alg/dt/8e12:.3f} of HBM; "
          f"result checksum {ck:x} counters {nd} {ne} (VELESDB_INT8_SPEC={os.environ.get('VELESDB_INT8_SPEC', 'default')})", flush=True)
