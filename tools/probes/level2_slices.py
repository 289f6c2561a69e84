#!/usr/bin/env python3
"""Level-2 selection over successive query slices (as bench.py's step loop): unproven counts per slice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import velesdb_amd as va
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
N, D, K, Q = 1_000_000, 768, 10, 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(42)
ix = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(32, 400, N))
st = torch.cuda.current_stream().cuda_stream
c = torch.randn((N, D), generator=g, device=dev); torch.cuda.synchronize()
ix.upload_dev(0, c.data_ptr(), N, st); del c
g.manual_seed(43)
pool = torch.randn((8192, D), generator=g, device=dev)
ids = torch.empty((Q, K), dtype=torch.int64, device=dev); sc = torch.empty((Q, K), dtype=torch.float32, device=dev); cnt = torch.empty((Q,), dtype=torch.int32, device=dev)
for lvl in (2, 1):
    va.set_split_selector(lvl)
    for i in range(8):
        off = (i * Q) % (8192 - Q + 1)
        ix.search_batch_dev(pool[off:off + Q].data_ptr(), Q, K, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
        torch.cuda.synchronize()
        print(lvl, i, off, ix.last_select_level(), ix.last_split_stats(), flush=True)
# back-to-back without synchronising in between (the bench's loop)
va.set_split_selector(2)
t0 = time.perf_counter()
for i in range(16):
    off = (i * Q) % (8192 - Q + 1)
    ix.search_batch_dev(pool[off:off + Q].data_ptr(), Q, K, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), cnt.data_ptr(), st)
torch.cuda.synchronize()
print("16 back-to-back batches: %.3f ms each" % ((time.perf_counter() - t0) / 16 * 1e3), ix.last_split_stats())
