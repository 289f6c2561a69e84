import sys; sys.path.insert(0, '/root/repo')
import numpy as np, velesdb_amd as va
from oracle import pyoracle as po
DM = va.DistanceMetric
rng = np.random.default_rng(2)
for dim in (256, 768):
    n = 30000
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    ix = va.HnswIndex(dim, DM.Euclidean); ix.upload(np.arange(n), rows)
    for nq in (12, 16, 24, 32, 33, 64, 128):
        Q = rng.standard_normal((nq, dim)).astype(np.float32)
        for k in (10, 32, 33, 40, 48, 64, 100):
            gi, gs, gc = ix.search_batch_brute_force(Q, k)
            ei, es = po.scan_topk(po.EUCLIDEAN, rows, Q, k, po.MODE_C, nthreads=8)
            bad = int(np.sum(np.any(gi != ei, axis=1)))
            if bad: print(f"dim={dim} nq={nq} k={k}: {bad} bad queries")
print("done")
