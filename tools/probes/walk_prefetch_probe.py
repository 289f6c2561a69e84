"""A / B of the latency-mode walk's neighbour-list prefetch (VELESDB_HNSW_PREFETCH_IDS=0|1, read when the library loads: run twice).
Single-query and 64-query calls on an N-row graph: per-call time, the kernel's HIP-event time, the prediction's hit rate, and a
checksum of ids / score bits / counters (must be equal between the two runs).
usage: walk_prefetch_probe.py [rows] [data: iid|emb]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import velesdb_amd as va
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "iid"
D, K, EF = 768, 10, 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(42)
if kind == "emb":  # 32 latent factors x random projection + noise (bench.py's embedding-like corpus)
    proj = torch.randn((32, D), generator=g, device=dev)
    corpus = torch.randn((N, 32), generator=g, device=dev) @ proj + 0.25 * torch.randn((N, D), generator=g, device=dev)
    qs = torch.randn((512, 32), generator=g, device=dev) @ proj + 0.25 * torch.randn((512, D), generator=g, device=dev)
else:
    corpus = torch.randn((N, D), generator=g, device=dev)
    qs = torch.randn((512, D), generator=g, device=dev)
ix = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(32, 400, N))
torch.cuda.synchronize()
ix.upload_dev(0, corpus.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
del corpus
ix.build_graph(0)
torch.cuda.synchronize()
Q = qs.cpu().numpy()
crc = 0
for nq in (1, 64):
    reps = 200 if nq == 1 else 50
    for i in range(5):
        ix._search_raw(Q[:nq], K, EF, va.MODE_HNSW)
    t0 = time.perf_counter()
    for i in range(reps):
        o = (i * nq) % (512 - nq + 1)
        ix._search_raw(Q[o:o + nq], K, EF, va.MODE_HNSW)
    dt = (time.perf_counter() - t0) / reps
    va.set_kernel_timing(True)
    kms, hits, exps = [], 0, 0
    for i in range(reps):
        o = (i * nq) % (512 - nq + 1)
        ids, sc, cnt = ix._search_raw(Q[o:o + nq], K, EF, va.MODE_HNSW)
        kms.append(ix.last_kernel_ms()[0])
        nd, ne = ix.last_search_stats()
        hits += ix.last_prefetch_hits(); exps += ne
        crc = zlib.crc32(np.ascontiguousarray(ids).tobytes() + np.ascontiguousarray(sc).view(np.uint32).tobytes() + np.array([nd, ne], dtype=np.uint64).tobytes(), crc)
    va.set_kernel_timing(False)
    print(f"prefetch={os.environ.get('VELESDB_HNSW_PREFETCH_IDS', '1')} {N}x{D} {kind} nq={nq}: {dt*1e6:.1f} us per call, kernel median {np.median(kms)*1e3:.1f} us "
          f"(min {np.min(kms)*1e3:.1f}), expansions/query {exps/reps/nq:.1f}, prefetched {hits}/{exps} = {hits/max(exps,1):.3f}", flush=True)
print(f"checksum of ids / score bits / counters: {crc:08x}", flush=True)
