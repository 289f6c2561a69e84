"""PCIe-inclusive rate of the host-buffer entry points (vdb_hip_index_search_batch: queries in host memory, results back
to host memory) next to the device-pointer entry point bench.py times.  1M x 768 f32 cosine, k = 10."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import velesdb_amd as va
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(42)
N, D, K = 1_000_000, 768, 10
corpus = torch.randn((N, D), generator=g, device=dev)
ix = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(32, 400, N))
torch.cuda.synchronize()
ix.upload_dev(0, corpus.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
del corpus
rng = np.random.default_rng(1)
for nq, reps in ((1024, 10), (64, 20), (1, 50)):
    Q = rng.standard_normal((nq, D)).astype(np.float32)
    for _ in range(2): ix.search_batch_brute_force(Q, K)
    t0 = time.perf_counter()
    for _ in range(reps): ix.search_batch_brute_force(Q, K)
    dt = (time.perf_counter() - t0) / reps
    print(f"host buffers, {nq} queries per call: {dt*1e3:.3f} ms per call = {nq/dt:.0f} queries/s (H2D {nq*D*4/1e6:.2f} MB, D2H {nq*K*12/1e3:.1f} KB per call)")
