#!/usr/bin/env python3
"""Fixed cost of one launch of the bf16 LDS-DMA kernel: result-mode searches over n = 16 384 (1 + T) rows run the seed sweep +
ONE launch with T row tiles per block (64 row groups x 4 query tiles at 1 024 queries).  Prints the HIP-event time of the launch
per T; under rocprofv3 --kernel-trace the launches appear in the same order (1 warm-up + reps per T)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])

p = argparse.ArgumentParser()
p.add_argument("--tiles", default="3,4,6,8,16,32")
p.add_argument("--reps", type=int, default=4)
p.add_argument("--nq", type=int, default=1024)
p.add_argument("--extra", type=int, default=0, help="rows added to every size (64: a ragged last tile)")
a = p.parse_args()
dev = torch.device("cuda", 0)
dim, k = 768, 10
g = torch.Generator(device=dev)
st = torch.cuda.current_stream().cuda_stream
g.manual_seed(43)
queries = torch.randn((a.nq, dim), generator=g, device=dev)
ids = torch.empty((a.nq, k), dtype=torch.int64, device=dev)
sc = torch.empty((a.nq, k), dtype=torch.float32, device=dev)
cnt = torch.empty((a.nq,), dtype=torch.int32, device=dev)
for T in [int(x) for x in a.tiles.split(",")]:
    rows = 16384 * (1 + T) + a.extra
    ix = va.HnswIndex(dim, va.DistanceMetric.Cosine, va.HnswParams(32, 400, rows))
    ix.enable_bf16()
    g.manual_seed(42)
    c = torch.randn((rows, dim), generator=g, device=dev)
    torch.cuda.synchronize()
    ix.upload_dev(0, c.data_ptr(), rows, st)
    del c
    ms = []
    for r in range(a.reps + 1):
        va.set_kernel_timing(True)
        ix.search_batch_dev(queries.data_ptr(), a.nq, k, 0, va.MODE_BRUTE_BF16, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
        torch.cuda.synchronize()
        kms, nl = ix.last_kernel_ms()
        if r:
            ms.append(kms)
    va.set_kernel_timing(False)
    print(f"T={T} rows={rows} launches={nl} sweep-kernel ms: " + " ".join(f"{m * 1e3:.1f}" for m in ms) + " us", flush=True)
    del ix
