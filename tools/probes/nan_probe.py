import sys; sys.path.insert(0, '/root/repo')
import numpy as np, velesdb_amd as va
from oracle import pyoracle as po
DM = va.DistanceMetric
rng = np.random.default_rng(1)
rows = rng.standard_normal((100, 64)).astype(np.float32)
rows[17, 3] = np.nan
rows[18, 3] = -np.nan
rows[19, 0] = np.inf
q = rng.standard_normal(64).astype(np.float32)
for m, pm in ((DM.Euclidean, po.EUCLIDEAN), (DM.Cosine, po.COSINE), (DM.DotProduct, po.DOT)):
    got = va.HipDistance(m).batch_distance(q, rows[17:20])
    exp = po.batch_distance(pm, q, rows[17:20], po.MODE_C)
    print(m.name, [hex(x) for x in got.view(np.uint32)], [hex(x) for x in exp.view(np.uint32)])
    ix = va.HnswIndex(64, m); ix.upload(np.arange(100), rows)
    gi, gs, gc = ix.search_batch_brute_force(q[None, :], 100)
    ei, es = po.scan_topk(pm, rows, q[None, :], 100, po.MODE_C)
    print("  first/last ids gpu", gi[0, :3], gi[0, -3:], "oracle", ei[0, :3], ei[0, -3:])
