"""Where does the time of ONE single-query graph search through the host entry point go?  (1 M: 1.16 ms per call around a 0.87 ms
walk.)  Run under `rocprofv3 --hip-trace --kernel-trace --stats`: per-API and per-kernel durations of 200 one-query calls.
usage: hnsw_call_overhead.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import velesdb_amd as va
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
D, K = 768, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(42)
corpus = torch.randn((N, D), generator=g, device=dev)
ix = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(32, 400, N))
torch.cuda.synchronize()
ix.upload_dev(0, corpus.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
del corpus
ix.build_graph(0)
torch.cuda.synchronize()
rng = np.random.default_rng(1)
Q = rng.standard_normal((256, D)).astype(np.float32)
for mode, ef, name in ((va.MODE_HNSW, 128, "hnsw"), (va.MODE_BRUTE, 0, "brute")):
    for i in range(5):
        ix._search_raw(Q[i:i + 1], K, ef, mode)
    va.set_kernel_timing(True)
    t0 = time.perf_counter()
    kms = []
    for i in range(200):
        ix._search_raw(Q[i:i + 1], K, ef, mode)
        kms.append(ix.last_kernel_ms()[0])
    dt = (time.perf_counter() - t0) / 200
    va.set_kernel_timing(False)
    t0 = time.perf_counter()
    for i in range(200):
        ix._search_raw(Q[i:i + 1], K, ef, mode)
    dt2 = (time.perf_counter() - t0) / 200
    print(f"{name}: {dt2*1e6:.1f} us per call (python + ctypes included; {dt*1e6:.1f} with kernel timing on), dominant kernel {np.median(kms)*1e3:.1f} us (HIP events)", flush=True)
