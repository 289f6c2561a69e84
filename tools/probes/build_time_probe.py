"""Graph construction time (batched GPU build, M 32 / ef_construction 400) — A/B of kernel variants via VELESDB_HIP_LIB.
usage: build_time_probe.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import velesdb_amd as va
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = 768
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(42)
c = torch.randn((N, D), generator=g, device=dev)
ix = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(32, 400, N))
torch.cuda.synchronize()
ix.upload_dev(0, c.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
del c
t0 = time.perf_counter(); ix.build_graph(0); torch.cuda.synchronize()
dt = time.perf_counter() - t0
chk = sum(sum(ix.neighbors(0, n)) for n in (0, 1, N // 2, N - 1))
print(f"{N} x {D} graph build: {dt:.2f} s = {N/dt:.0f} inserts/s; neighbour checksum {chk} (lib {os.environ.get('VELESDB_HIP_LIB', 'product')})", flush=True)
