"""Probe (round 6): the WIDE selection at small k against the k <= 10 stage — per-call time of 1 024-query batches at 1 M x 768 for
k = 10 (block-local lists) and k = 11 (WIDE); run under `rocprofv3 --kernel-trace --stats` for the per-kernel breakdown."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])

ks = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10,11").split(",")]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
N, D, Q = 1_000_000, 768, 1024
rows = torch.randn((N, D), generator=g, device=dev)
g.manual_seed(43)
qs = torch.randn((4 * Q, D), generator=g, device=dev)
ix = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(32, 400, N))
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
ix.upload_dev(0, rows.data_ptr(), N, st)
torch.cuda.synchronize()
for k in ks:
    ids = torch.empty((Q, k), dtype=torch.int64, device=dev)
    sc = torch.empty((Q, k), dtype=torch.float32, device=dev)
    n = torch.empty((Q,), dtype=torch.int32, device=dev)
    for i in range(5):
        ix.search_batch_dev(qs[(i % 4) * Q:].data_ptr(), Q, k, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(20):
        ix.search_batch_dev(qs[(i % 4) * Q:].data_ptr(), Q, k, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    print(f"k={k} ms_per_batch {dt * 1e3:.4f} qps {Q / dt:.1f} level {ix.last_select_level()} unproven {ix.last_split_stats()}", flush=True)
