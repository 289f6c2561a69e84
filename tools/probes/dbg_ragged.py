import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import velesdb_amd as va
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
from oracle import pyoracle as po
n, dim = 70_000, 128
rng = np.random.default_rng(n * 13 + dim)
rows = rng.standard_normal((n, dim), dtype=np.float32)
ix = va.HnswIndex(dim, va.DistanceMetric.Cosine, va.HnswParams(16, 100, n))
ix.upload(np.arange(n), rows)
ix.enable_bf16()
for nq, k in [(224, 10), (600, 1), (1024, 10)]:
    qs = rng.standard_normal((nq, dim), dtype=np.float32)
    gi, gs, gc = ix.search_batch_brute_force_bf16(qs, k)
    eid, esc = po.scan_topk_bf16(po.COSINE, rows, qs, k, nthreads=8)
    bad = [(q, r) for q in range(nq) for r in range(k) if gi[q, r] != eid[q, r]]
    print(nq, k, "mismatching (query, rank):", len(bad))
    for q, r in bad[:12]:
        print("  q", q, "rank", r, "gpu", int(gi[q, r]), float(gs[q, r]), "oracle", int(eid[q, r]), float(esc[q, r]),
              "oracle row tile", int(eid[q, r]) // 256, "row in tile", int(eid[q, r]) % 256, "gpu cnt", int(gc[q]))
