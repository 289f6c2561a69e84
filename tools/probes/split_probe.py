#!/usr/bin/env python3
"""Timing probe of the exact f32 sweep for large batches: split-bf16 selector on / off (BASELINE configs[1] shape)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])

p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=1_000_000)
p.add_argument("--dim", type=int, default=768)
p.add_argument("--k", type=int, default=10)
p.add_argument("--nq", type=int, default=1024)
p.add_argument("--metric", default="cosine")
p.add_argument("--reps", type=int, default=10)
a = p.parse_args()
dev = torch.device("cuda", 0)
metric = {"cosine": va.DistanceMetric.Cosine, "dot": va.DistanceMetric.DotProduct, "euclidean": va.DistanceMetric.Euclidean}[a.metric]
ix = va.HnswIndex(a.dim, metric, va.HnswParams(32, 400, a.rows))
g = torch.Generator(device=dev)
g.manual_seed(42)
st = torch.cuda.current_stream().cuda_stream
for base in range(0, a.rows, 1_000_000):
    n = min(1_000_000, a.rows - base)
    c = torch.randn((n, a.dim), generator=g, device=dev)
    torch.cuda.synchronize()
    ix.upload_dev(base, c.data_ptr(), n, st)
    del c
g.manual_seed(43)
queries = torch.randn((a.nq, a.dim), generator=g, device=dev)
ids = torch.empty((a.nq, a.k), dtype=torch.int64, device=dev)
sc = torch.empty((a.nq, a.k), dtype=torch.float32, device=dev)
cnt = torch.empty((a.nq,), dtype=torch.int32, device=dev)
res = {}
for on in ((2, 0) if a.metric == "euclidean" else (2, 1, 0)):
    va.set_split_selector(on)
    for _ in range(2):
        ix.search_batch_dev(queries.data_ptr(), a.nq, a.k, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
    torch.cuda.synchronize()
    va.set_kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(a.reps):
        ix.search_batch_dev(queries.data_ptr(), a.nq, a.k, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    kms, nl = ix.last_kernel_ms()
    va.set_kernel_timing(False)
    res[on] = (ids.cpu().numpy().copy(), sc.cpu().numpy().view(np.uint32).copy())
    extra = f", level {ix.last_select_level()}, last batch: {ix.last_split_stats()}" if on else ""
    print(f"split={int(on)} {a.rows}x{a.dim} {a.metric} nq={a.nq} k={a.k}: {dt * 1e3:.3f} ms/batch = {a.nq / dt:.0f} q/s; timed region {kms:.3f} ms "
          f"= {2.0 * a.rows * a.dim * a.nq / (kms * 1e-3) / 1e12:.0f} algorithmic TFLOP/s{extra}", flush=True)
print("identical:", all(bool(np.array_equal(res[l][0], res[0][0]) and np.array_equal(res[l][1], res[0][1])) for l in res if l))
