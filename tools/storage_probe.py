#!/usr/bin/env python3
"""GPU probe: exact scans over the SQ8 / binary codes of an N x D corpus (storage modes, core/quantization.rs)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])

p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=1_000_000)
p.add_argument("--dim", type=int, default=768)
p.add_argument("--k", type=int, default=10)
p.add_argument("--metric", default="cosine")
p.add_argument("--nqs", default="1,4,16,64")
a = p.parse_args()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
corpus = torch.randn((a.rows, a.dim), generator=g, device=dev)
queries = torch.randn((1024, a.dim), generator=g, device=dev)
metric = {"cosine": va.DistanceMetric.Cosine, "euclidean": va.DistanceMetric.Euclidean, "dot": va.DistanceMetric.DotProduct}[a.metric]
ix = va.HnswIndex(a.dim, metric, va.HnswParams(32, 400, a.rows))
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
ix.upload_dev(0, corpus.data_ptr(), a.rows, st)
del corpus
for mode, name, smode, row_bytes in ((va.MODE_BRUTE_SQ8, "sq8", va.StorageMode.SQ8, a.dim + 12 + (4 if a.metric == "cosine" else 0)),
                                     (va.MODE_BRUTE_BINARY, "binary", va.StorageMode.Binary, ((a.dim + 127) // 128) * 16)):
    t0 = time.perf_counter()
    ix.set_storage_mode(smode)
    print(f"{name}: encoded {a.rows} rows in {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    for nq in [int(x) for x in a.nqs.split(",")]:
        if name == "sq8" and os.environ.get("SQ8_EXACT") == "1":
            va.set_split_selector(0)
        ids = torch.empty((nq, a.k), dtype=torch.int64, device=dev)
        sc = torch.empty((nq, a.k), dtype=torch.float32, device=dev)
        n = torch.empty((nq,), dtype=torch.int32, device=dev)
        for _ in range(2):
            ix.search_batch_dev(queries.data_ptr(), nq, a.k, 0, mode, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
        torch.cuda.synchronize()
        va.set_kernel_timing(True)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search_batch_dev(queries.data_ptr(), nq, a.k, 0, mode, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        kms, nl = ix.last_kernel_ms()
        va.set_kernel_timing(False)
        alg = a.rows * row_bytes * (nq if name == "binary" else 1)
        print(f"  {name} nq={nq:3d}: call {dt*1e3:8.3f} ms ({nq/dt:9.1f} qps)  sweep kernel {kms:7.4f} ms x{nl}: "
              f"{alg/(kms*1e-3)/1e9:7.1f} GB/s ({alg/(kms*1e-3)/1e9/8000:.3f} of 8 TB/s)", flush=True)
