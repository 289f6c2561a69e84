#!/usr/bin/env python3
"""GPU probe for BASELINE configs[3]: N x 768 bf16 corpus, 1024 queries per batch as a bf16 MFMA GEMM distance,
k = 10, single MI355X.  Not part of the product or the test-suite."""
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
p.add_argument("--rows", type=int, default=10_000_000)
p.add_argument("--dim", type=int, default=768)
p.add_argument("--k", type=int, default=10)
p.add_argument("--nq", type=int, default=1024)
p.add_argument("--metric", default="cosine")
a = p.parse_args()
dev = torch.device("cuda", 0)
metric = {"cosine": va.DistanceMetric.Cosine, "dot": va.DistanceMetric.DotProduct}[a.metric]
ix = va.HnswIndex(a.dim, metric, va.HnswParams(32, 400, a.rows))
ix.enable_bf16()
g = torch.Generator(device=dev)
g.manual_seed(42)
st = torch.cuda.current_stream().cuda_stream
chunk = 1_000_000
for base in range(0, a.rows, chunk):
    n = min(chunk, a.rows - base)
    c = torch.randn((n, a.dim), generator=g, device=dev)
    torch.cuda.synchronize()
    ix.upload_dev(base, c.data_ptr(), n, st)
    del c
g.manual_seed(43)
queries = torch.randn((a.nq, a.dim), generator=g, device=dev)
ids = torch.empty((a.nq, a.k), dtype=torch.int64, device=dev)
sc = torch.empty((a.nq, a.k), dtype=torch.float32, device=dev)
cnt = torch.empty((a.nq,), dtype=torch.int32, device=dev)
for mode, name in ((va.MODE_BRUTE_BF16, "bf16 mfma"), (va.MODE_BRUTE, "f32 mfma")):
    for _ in range(1):
        ix.search_batch_dev(queries.data_ptr(), a.nq, a.k, 0, mode, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
    torch.cuda.synchronize()
    va.set_kernel_timing(True)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        ix.search_batch_dev(queries.data_ptr(), a.nq, a.k, 0, mode, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    kms, nl = ix.last_kernel_ms()
    va.set_kernel_timing(False)
    esz = 2 if mode == va.MODE_BRUTE_BF16 else 4
    per_pass = a.rows * a.dim * esz + a.rows * 4
    tflops = 2.0 * a.rows * a.dim * a.nq / dt / 1e12
    print(f"{name}: {a.rows}x{a.dim}, {a.nq} queries: {dt*1e3:.2f} ms/batch = {a.nq/dt:.0f} qps; sweep kernel "
          f"{kms:.3f} ms x{nl} launches = {per_pass/(kms*1e-3)/1e9:.0f} GB/s per pass "
          f"({per_pass/(kms*1e-3)/1e9/8000:.3f} of 8 TB/s); {tflops:.1f} TFLOP/s", flush=True)
    if mode == va.MODE_BRUTE_BF16:
        bf_ids = ids.cpu().numpy().copy()
    else:
        f_ids = ids.cpu().numpy()
        print("recall@10 of the bf16 sweep against the exact f32 sweep:",
              float(np.mean([len(set(bf_ids[i]) & set(f_ids[i])) / a.k for i in range(a.nq)])))
