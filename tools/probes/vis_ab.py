#!/usr/bin/env python3
"""A/B of the visited set of the graph walks: HBM bitmap (atomicOr) against the exact LDS hash set, per kernel — latency mode
(1 .. 64 queries per call), throughput mode (8 192 queries) and the int8 walk.  The parent builds one graph and saves it; every
variant is a child process (the switches are read once per process) that loads it, times the calls and dumps ids / score bits /
counters, which must be identical across variants."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--rows", type=int, default=1_000_000)
p.add_argument("--dim", type=int, default=768)
p.add_argument("--ef", type=int, default=128)
p.add_argument("--dir", default="/tmp/vis_ab_index")
p.add_argument("--child", default="")
p.add_argument("--out", default="")
a = p.parse_args()

import torch  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
if not a.child:
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    ix = va.HnswIndex(a.dim, va.DistanceMetric.Cosine, va.HnswParams(32, 400, a.rows))
    c = torch.randn((a.rows, a.dim), generator=g, device=dev)
    torch.cuda.synchronize()
    ix.upload_dev(0, c.data_ptr(), a.rows, st)
    del c
    t0 = time.perf_counter()
    ix.build_graph(0)
    print(f"build {a.rows}x{a.dim}: {time.perf_counter() - t0:.1f} s", flush=True)
    ix.save(a.dir)
    ix.close()
    outs = {}
    for name, env in [("bitmap", {"VELESDB_HNSW_VIS_LDS": "0", "VELESDB_INT8_VIS_LDS": "0"}),
                      ("lds_lat_only", {"VELESDB_INT8_VIS_LDS": "0"}),
                      ("lds_everywhere", {"VELESDB_HNSW_VIS_LDS": "1", "VELESDB_INT8_VIS_LDS": "1"})]:
        out = f"/tmp/vis_ab_{name}.npz"
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, "--out", out, "--dir", a.dir, "--ef", str(a.ef),
                            "--rows", str(a.rows), "--dim", str(a.dim)], env=e, capture_output=True, text=True, timeout=900)
        print(r.stdout[-3000:], r.stderr[-1500:] if r.returncode else "", flush=True)
        if r.returncode == 0:
            outs[name] = np.load(out)
    base = outs.get("bitmap")
    for name, o in outs.items():
        if base is not None and name != "bitmap":
            same = all(np.array_equal(base[k], o[k]) for k in base.files)
            print(f"{name} == bitmap (ids, score bits, counters of every call): {same}", flush=True)
    sys.exit(0)

ix = va.HnswIndex.load(a.dir)
ix.train_quantizer(0)
g = torch.Generator(device=dev)
g.manual_seed(43)
q = torch.randn((8192, a.dim), generator=g, device=dev)
res = {}
print(f"== {a.child}", flush=True)
for mode, mname, nqs in ((va.MODE_HNSW, "f32", (1, 16, 64, 256, 8192)), (va.MODE_HNSW_INT8, "int8", (8192,))):
    for nq in nqs:
        ids = torch.empty((nq, 10), dtype=torch.int64, device=dev)
        sc = torch.empty((nq, 10), dtype=torch.float32, device=dev)
        n = torch.empty((nq,), dtype=torch.int32, device=dev)
        for _ in range(2):
            ix.search_batch_dev(q.data_ptr(), nq, 10, a.ef, mode, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
        torch.cuda.synchronize()
        reps = 20 if nq <= 256 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ix.search_batch_dev(q.data_ptr(), nq, 10, a.ef, mode, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        nd, ne = ix.last_search_stats()
        per = a.dim * 4 if mname == "f32" else a.dim + 4
        alg = nd * per + ne * 64 * 4 + (nq * 10 * 4 * a.dim * 4 if mname == "int8" else 0)
        print(f"{mname} nq={nq:5d}: {dt * 1e3:8.3f} ms/call = {nq / dt:9.0f} q/s; n_dist/q {nd / nq:.0f}; {alg / dt / 1e9:6.0f} GB/s = "
              f"{alg / dt / 8e12:.3f} of HBM; overflows {int((n.cpu().numpy() == -1).sum())}", flush=True)
        res[f"{mname}_{nq}_ids"] = ids.cpu().numpy()
        res[f"{mname}_{nq}_sc"] = sc.cpu().numpy().view(np.uint32)
        res[f"{mname}_{nq}_stats"] = np.array([nd, ne], np.uint64)
np.savez(a.out, **res)
