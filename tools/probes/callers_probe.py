"""The reference's calling pattern against the C ABI: T host threads x one query per vdb_hip_index_search call (tools/callers_bench.cpp),
on the 1 M x 768 graph and on the exact sweep, for several settings of the combining front; plus the host-pointer batch entry
point (PCIe-inclusive) next to the device-resident one.  usage: callers_probe.py [rows] [--no-graph]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import velesdb_amd as va
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])

N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1_000_000
D, K = 768, 10
dev = torch.device("cuda", 0)
cb = C.CDLL(os.path.join(ROOT, "tools", "libcallers_bench.so"))
cb.callers_run.restype = C.c_int
cb.callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int, C.c_double,
                           C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]


def callers(ix, q, mode, ef, threads, seconds=1.5, per_call=1, ref=None):
    out = np.zeros(8, dtype=np.float64)
    rid = rsc = rn = None
    if ref is not None:
        rid, rsc, rn = [np.ascontiguousarray(x) for x in ref]
    rc = cb.callers_run(ix._h, q.ctypes.data, q.shape[0], D, K, ef, mode, threads, seconds, 20, per_call,
                        rid.ctypes.data if rid is not None else None, rsc.ctypes.data if rsc is not None else None,
                        rn.ctypes.data if rn is not None else None, out.ctypes.data)
    assert rc == 0
    return {"threads": threads, "qps": round(out[0], 1), "p50_us": round(out[1], 1), "p99_us": round(out[2], 1), "mean_us": round(out[3], 1),
            "calls": int(out[4]), "mismatch": int(out[5]), "failed": int(out[6])}


g = torch.Generator(device=dev)
g.manual_seed(42)
corpus = torch.randn((N, D), generator=g, device=dev)
ix = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(32, 400, N))
torch.cuda.synchronize()
stream = torch.cuda.current_stream().cuda_stream
ix.upload_dev(0, corpus.data_ptr(), N, stream)
del corpus
rng = np.random.default_rng(1)
NQ = 4096
Q = rng.standard_normal((NQ, D)).astype(np.float32)

# ---- host-pointer batch entry point vs the device-resident one (exact sweep) ----
dq = torch.from_numpy(Q).to(dev)
o_i = torch.empty((1024, K), dtype=torch.int64, device=dev)
o_s = torch.empty((1024, K), dtype=torch.float32, device=dev)
o_n = torch.empty((1024,), dtype=torch.int32, device=dev)
for nq, reps in ((1024, 20), (256, 20), (64, 30), (16, 30), (1, 50)):
    for _ in range(3):
        ix.search_batch_brute_force(Q[:nq], K)
    t0 = time.perf_counter()
    for r in range(reps):
        ix.search_batch_brute_force(Q[(r * nq) % (NQ - nq + 1):][:nq], K)
    dt = (time.perf_counter() - t0) / reps
    for _ in range(3):
        ix.search_batch_dev(dq.data_ptr(), nq, K, 0, va.MODE_BRUTE, o_i.data_ptr(), o_s.data_ptr(), o_n.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        ix.search_batch_dev(dq.data_ptr(), nq, K, 0, va.MODE_BRUTE, o_i.data_ptr(), o_s.data_ptr(), o_n.data_ptr(), stream)
    torch.cuda.synchronize()
    dd = (time.perf_counter() - t0) / reps
    print(f"host_entry brute nq={nq}: host pointers {dt*1e3:.3f} ms/call = {nq/dt:.0f} q/s | device resident {dd*1e3:.3f} ms/call = {nq/dd:.0f} q/s "
          f"| stage_max={os.environ.get('VELESDB_STAGE_MAX_BYTES', 'default')}", flush=True)

# ---- callers on the exact sweep ----
ref_b = ix.search_batch_brute_force(Q, K)
for infl in (0, 2):
    ix.set_option(va.OPT_COMBINE_INFLIGHT, infl)
    for T in (1, 4, 16, 64, 128):
        print(f"callers brute inflight={infl}:", callers(ix, Q, va.MODE_BRUTE, 0, T, ref=ref_b), ix.combine_stats(), flush=True)
ix.set_option(va.OPT_COMBINE_MAX_BATCH, 0)
for T in (1, 16):
    print("callers brute combining off:", callers(ix, Q, va.MODE_BRUTE, 0, T, seconds=1.0, ref=ref_b), flush=True)
ix.set_option(va.OPT_COMBINE_MAX_BATCH, -1)
ix.set_option(va.OPT_COMBINE_INFLIGHT, -1)

if "--no-graph" not in sys.argv:
    t0 = time.perf_counter()
    ix.build_graph(0)
    torch.cuda.synchronize()
    print(f"graph built in {time.perf_counter()-t0:.1f} s", flush=True)
    rid, rsc, rn = ix._search_raw(Q, K, 128, va.MODE_HNSW)
    ref_h = (rid, rsc, rn)
    for nq, reps in ((1024, 10), (64, 30), (1, 50)):
        t0 = time.perf_counter()
        for r in range(reps):
            ix._search_raw(Q[(r * nq) % (NQ - nq + 1):][:nq], K, 128, va.MODE_HNSW)
        dt = (time.perf_counter() - t0) / reps
        print(f"host_entry hnsw nq={nq}: {dt*1e3:.3f} ms/call = {nq/dt:.0f} q/s (python wrapper included)", flush=True)
    for infl in (0, 1, 3):
        ix.set_option(va.OPT_COMBINE_INFLIGHT, infl)
        for T in (1, 4, 16, 64, 128):
            print(f"callers hnsw inflight={infl}:", callers(ix, Q, va.MODE_HNSW, 128, T, ref=ref_h), ix.combine_stats(), flush=True)
    ix.set_option(va.OPT_COMBINE_INFLIGHT, 0)
    for w in (0, 300):
        ix.set_option(va.OPT_COMBINE_WINDOW_US, w)
        for T in (4, 16, 64):
            print(f"callers hnsw window={w}us:", callers(ix, Q, va.MODE_HNSW, 128, T, ref=ref_h), flush=True)
    ix.set_option(va.OPT_COMBINE_WINDOW_US, -1)
    ix.set_option(va.OPT_COMBINE_MAX_BATCH, 0)
    for T in (1, 16):
        print("callers hnsw combining off:", callers(ix, Q, va.MODE_HNSW, 128, T, seconds=1.0, ref=ref_h), flush=True)
