#!/usr/bin/env python3
"""GPU probe: per-launch time of the sweep kernel for each query-tile size, and end-to-end time per
search call, on a synthetic N x D corpus.  Not part of the product or the test-suite."""
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
p.add_argument("--nqs", default="16,32,48,64,96,128,192,256,512")
p.add_argument("--tile", type=int, default=128)
p.add_argument("--engine", type=int, default=1)
a = p.parse_args()
va.set_max_query_tile(a.tile)
va.set_sweep_engine(a.engine)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
corpus = torch.randn((a.rows, a.dim), generator=g, device=dev)
queries = torch.randn((1024, a.dim), generator=g, device=dev)
metric = {"cosine": va.DistanceMetric.Cosine, "euclidean": va.DistanceMetric.Euclidean, "dot": va.DistanceMetric.DotProduct,
          "hamming": va.DistanceMetric.Hamming, "jaccard": va.DistanceMetric.Jaccard}[a.metric]
if a.metric in ("hamming", "jaccard"):  # N(0,1) thresholded at 0.5 (the kernels sweep the packed bits: 96 B/row at 768-D)
    corpus = (corpus > 0.5).float()
    queries = (queries > 0.5).float()
ix = va.HnswIndex(a.dim, metric, va.HnswParams(32, 400, a.rows))
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
ix.upload_dev(0, corpus.data_ptr(), a.rows, st)
del corpus
alg = a.rows * a.dim * 4 + (a.rows * 4 if a.metric == "cosine" else 0)
if a.metric in ("hamming", "jaccard"):
    alg = a.rows * ((a.dim + 127) // 128 * 16)  # packed rows, words rounded to 4
for nq in [int(x) for x in a.nqs.split(",")]:
    ids = torch.empty((nq, a.k), dtype=torch.int64, device=dev)
    sc = torch.empty((nq, a.k), dtype=torch.float32, device=dev)
    n = torch.empty((nq,), dtype=torch.int32, device=dev)
    for _ in range(3):
        ix.search_batch_dev(queries.data_ptr(), nq, a.k, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
    torch.cuda.synchronize()
    va.set_kernel_timing(True)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ix.search_batch_dev(queries.data_ptr(), nq, a.k, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kms, nl = ix.last_kernel_ms()
    va.set_kernel_timing(False)
    print(f"nq={nq:3d}: call {dt*1e3:8.3f} ms  ({nq/dt:9.1f} qps)  sweep kernel {kms:7.4f} ms x{nl}  "
          f"{alg/(kms*1e-3)/1e9:7.1f} GB/s ({alg/(kms*1e-3)/1e9/8000:.3f} of 8 TB/s)  "
          f"{2.0*a.rows*a.dim*nq/nl/(kms*1e-3)/1e12:6.1f} TFLOP/s", flush=True)
