"""1 024 packed-bit queries per batch over 1 M x 768 bits on the FP4 path (Hamming / Jaccard), 4 batches: the workload of
tools/probes/pmc_bits_traffic.sh (rocprofv3 --pmc FETCH_SIZE).  python tools/probes/bits_batch_probe.py [metric]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])

metric = {"hamming": va.DistanceMetric.Hamming, "jaccard": va.DistanceMetric.Jaccard}[sys.argv[1] if len(sys.argv) > 1 else "hamming"]
N, D, K, Q = 1_000_000, 768, 10, 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
src = (torch.randn((N, D), generator=g, device=dev) > 0.5).float()
qs = (torch.randn((Q, D), generator=g, device=dev) > 0.5).float()
ix = va.HnswIndex(D, metric, va.HnswParams(16, 100, N))
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
ix.upload_dev(0, src.data_ptr(), N, st)
torch.cuda.synchronize()
del src
ids = torch.empty((Q, K), dtype=torch.int64, device=dev)
sc = torch.empty((Q, K), dtype=torch.float32, device=dev)
n = torch.empty((Q,), dtype=torch.int32, device=dev)
for _ in range(4):
    ix.search_batch_dev(qs.data_ptr(), Q, K, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
torch.cuda.synchronize()
assert ix.last_kernels() & va.KERNEL_BITS_GEMM
print("4 batches of", Q, "queries done")
ix.close()
