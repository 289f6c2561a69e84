"""One packed-bit query per call over 1 M x 768 bits (Hamming / Jaccard): per-call time and, under rocprofv3 --kernel-trace --stats,
the kernels of a call.  python tools/probes/bits_single_query_probe.py [metric]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from velesdb_amd import _ffi  # noqa: E402
if os.environ.get("VDB_PROBE_LIB"):  # the probe build: environment switches (VELESDB_BITS_FUSED, VELESDB_BITS_FUSED_PER_CU)
    _ffi.use_library(_ffi.PROBE_LIB_PATH)
import torch  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])

metric = {"hamming": va.DistanceMetric.Hamming, "jaccard": va.DistanceMetric.Jaccard}[sys.argv[1] if len(sys.argv) > 1 else "hamming"]
N, D, K = 1_000_000, 768, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
src = (torch.randn((N, D), generator=g, device=dev) > 0.5).float()
qs = (torch.randn((64, D), generator=g, device=dev) > 0.5).float()
ix = va.HnswIndex(D, metric, va.HnswParams(16, 100, N))
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
ix.upload_dev(0, src.data_ptr(), N, st)
torch.cuda.synchronize()
del src
ids = torch.empty((64, K), dtype=torch.int64, device=dev)
sc = torch.empty((64, K), dtype=torch.float32, device=dev)
n = torch.empty((64,), dtype=torch.int32, device=dev)
for nq in (1, 2, 8):
    for _ in range(3):
        ix.search_batch_dev(qs.data_ptr(), nq, K, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        ix.search_batch_dev(qs.data_ptr(), nq, K, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{sys.argv[1] if len(sys.argv) > 1 else 'hamming'} nq={nq}: {dt * 1e6:7.1f} us per call (device-resident queries, {reps} calls back to back)", flush=True)
ix.close()
