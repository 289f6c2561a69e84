"""Debug probe (round 6): where does the WIDE selection lose rows?  Prints, per case, the queries whose ids differ from the oracle, the
missing rows and their exact ranks."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
from oracle import pyoracle as po  # noqa: E402

rng = np.random.default_rng(2026)
rows = rng.standard_normal((140_000, 768), dtype=np.float32)
qs = rng.standard_normal((300, 768), dtype=np.float32)
for n in (70_000, 140_000):
    ix = va.HnswIndex(768, va.DistanceMetric.Cosine)
    ix.upload(np.arange(n, dtype=np.uint64), rows[:n])
    for k, nq in ((100, 64), (100, 256), (100, 300), (64, 64), (65, 64), (128, 17), (50, 64)):
        ids, sc, cnt = ix.search_batch_brute_force(qs[:nq], k)
        lvl = ix.last_select_level()
        nql, unp = ix.last_split_stats()
        eid, esc = po.scan_topk(po.COSINE, rows[:n], qs[:nq], k, po.MODE_M, nthreads=po.host_threads())
        bad = [q for q in range(nq) if not np.array_equal(ids[q], eid[q])]
        print(f"n={n} k={k} nq={nq} level={lvl} unproven={unp} bad_queries={len(bad)} {bad[:12]}", flush=True)
        for q in bad[:3]:
            miss = [(int(r), int(np.nonzero(eid[q] == r)[0][0])) for r in eid[q] if r not in set(ids[q].tolist())]
            print(f"   q={q} cnt={cnt[q]} missing (row, exact rank): {miss[:10]}  tiles of missing rows: {[m[0] // 256 for m in miss[:10]]}", flush=True)
    ix.close()
