#!/usr/bin/env python3
"""A/B probe of the bf16 GEMM-distance sweep: VELESDB_BF16_GLDS=1 (256 x 256 LDS-DMA kernel, sweep_gemm_bf16.hip) against
=0 (register-staged kernel of sweep_gemm.hip).  --save writes ids / score bits for a cross-process comparison."""
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
p.add_argument("--rows", type=int, default=10_000_000)
p.add_argument("--dim", type=int, default=768)
p.add_argument("--k", type=int, default=10)
p.add_argument("--nq", type=int, default=1024)
p.add_argument("--metric", default="cosine")
p.add_argument("--reps", type=int, default=5)
p.add_argument("--save", default="")
p.add_argument("--dead", type=int, default=0, help="soft-delete every n-th row")
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
if a.dead:
    for i in range(0, a.rows, a.dead):
        ix.remove(i)
g.manual_seed(43)
queries = torch.randn((a.nq, a.dim), generator=g, device=dev)
ids = torch.empty((a.nq, a.k), dtype=torch.int64, device=dev)
sc = torch.empty((a.nq, a.k), dtype=torch.float32, device=dev)
cnt = torch.empty((a.nq,), dtype=torch.int32, device=dev)


def run():
    ix.search_batch_dev(queries.data_ptr(), a.nq, a.k, 0, va.MODE_BRUTE_BF16, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)


run()
torch.cuda.synchronize()
va.set_kernel_timing(True)
t0 = time.perf_counter()
for _ in range(a.reps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
kms, nl = ix.last_kernel_ms()
va.set_kernel_timing(False)
tf = 2.0 * a.rows * a.dim * a.nq / (kms * 1e-3) / 1e12
print(f"GLDS={os.environ.get('VELESDB_BF16_GLDS', '1')} {a.rows}x{a.dim} {a.metric} nq={a.nq} k={a.k}: {dt * 1e3:.3f} ms/batch "
      f"({a.nq / dt:.0f} q/s), sweep kernel {kms:.3f} ms = {tf:.0f} TFLOP/s ({tf / 2500:.3f} of bf16 peak)", flush=True)
if a.save:
    np.savez(a.save, ids=ids.cpu().numpy(), sc=sc.cpu().numpy().view(np.uint32), cnt=cnt.cpu().numpy())
