#!/usr/bin/env python3
"""GPU fuzz of the smaller entry points against the oracle (not part of the test-suite): DistanceEngine::batch_distance for
all five metrics, the free vector utilities (norm / normalize / squared L2 / dot matrix / packed-u64 Hamming and Jaccard),
the MmapStorage import and vacuum — random shapes, zero / duplicate / integer rows.

    python tools/fuzz_misc.py --seconds 200 --seed 1
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
from velesdb_amd import simd as vs  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--seconds", type=float, default=200)
p.add_argument("--seed", type=int, default=1)
a = p.parse_args()
rng = np.random.default_rng(a.seed)
DM, SQ = va.DistanceMetric, va.SearchQuality
PO = {DM.Cosine: po.COSINE, DM.Euclidean: po.EUCLIDEAN, DM.DotProduct: po.DOT, DM.Hamming: po.HAMMING, DM.Jaccard: po.JACCARD}


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def make(kind, n, d, metric=None):
    if metric in (DM.Hamming, DM.Jaccard):
        return (rng.random((n, d)) > float(rng.choice([0.3, 0.6915, 0.95]))).astype(np.float32)
    r = rng.standard_normal((n, d)).astype(np.float32)
    if kind == "zeros":
        r[rng.random(n) < 0.3] = 0.0
    elif kind == "ints":
        r = rng.integers(-3, 4, size=(n, d)).astype(np.float32)
    elif kind == "dups" and n > 1:
        r = r[rng.integers(0, max(1, n // 5), n)]
    elif kind == "wide":
        r *= rng.choice([1e-8, 1.0, 1e6], size=(n, 1)).astype(np.float32)
    return r


t_end = time.time() + a.seconds
it = 0
stats = {"distance": 0, "utils": 0, "store": 0, "vacuum": 0}
while time.time() < t_end:
    it += 1
    what = str(rng.choice(["distance", "distance", "utils", "utils", "store", "vacuum"]))
    kind = str(rng.choice(["normal", "zeros", "ints", "dups", "wide"]))
    d = int(rng.choice([1, 2, 3, 5, 16, 31, 64, 100, 255, 256, 768, 1000, 1536]))
    n = int(rng.choice([1, 2, 63, 64, 65, 500, 3000]))
    tag = f"it={it} {what} n={n} d={d} {kind}"
    if what == "distance":
        metric = [DM.Cosine, DM.Euclidean, DM.DotProduct, DM.Hamming, DM.Jaccard][int(rng.integers(0, 5))]
        rows, q = make(kind, n, d, metric), make(kind, 1, d, metric)[0]
        got = va.HipDistance(metric).batch_distance(q, rows)
        exp = po.batch_distance(PO[metric], q, rows, po.MODE_C)
        assert np.array_equal(bits(got), bits(exp)), tag + f" {metric.name}"
    elif what == "utils":
        rows, q = make(kind, n, d), make(kind, 1, d)[0]
        nn = vs.batch_norm(rows)
        assert np.array_equal(bits(nn), bits(np.float32([np.sqrt(np.float32(po.norm_sq(r, po.MODE_C))) for r in rows]))), tag + " norm"
        u = vs.normalize_rows(rows)
        for i in range(n):
            e = rows[i] if nn[i] == 0.0 else rows[i] * (np.float32(1.0) / nn[i])
            assert np.array_equal(bits(u[i]), bits(e)), tag + " normalize"
        assert np.array_equal(bits(vs.batch_squared_l2(q, rows)), bits(np.float32([po.sql2(q, r, po.MODE_C) for r in rows]))), tag + " sql2"
        nq = int(rng.choice([1, 3, 17]))
        Q = make(kind, nq, d)
        m = vs.batch_dot_product(Q, rows[:200])
        assert np.array_equal(bits(m), bits(np.float32([[po.dot(x, v, po.MODE_C) for v in rows[:200]] for x in Q]))), tag + " dot matrix"
        w = int(rng.choice([1, 2, 12, 16, 33]))
        qb = rng.integers(0, 2**63, size=w, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=w, dtype=np.uint64)
        rb = rng.integers(0, 2**63, size=(min(n, 400), w), dtype=np.uint64) * np.uint64(2)
        if kind == "zeros":
            rb[::3] = 0
        assert vs.batch_hamming_binary(qb, rb).tolist() == [po.hamming_binary(qb, r) for r in rb], tag + " hamming_binary"
        ej = []
        for r in rb:
            inter = sum(bin(int(x) & int(y)).count("1") for x, y in zip(qb, r))
            uni = sum(bin(int(x) | int(y)).count("1") for x, y in zip(qb, r))
            ej.append(np.float32(1.0) if uni == 0 else np.float32(inter) / np.float32(uni))
        assert np.array_equal(bits(vs.batch_jaccard_binary(qb, rb)), bits(np.float32(ej))), tag + " jaccard_binary"
    elif what == "store":
        n, d = min(n, 500), min(d, 256)
        dd = tempfile.mkdtemp(prefix="vdb_fuzz_store_")
        try:
            st = po.MmapVectorStore(dd, d)
            ids = rng.choice(1 << 40, n, replace=False).astype(np.uint64)
            rows = make(kind, n, d)
            for i in range(n):
                st.store(int(ids[i]), rows[i])
            for i in rng.choice(n, n // 4, replace=False):
                st.store(int(ids[i]), make("normal", 1, d)[0])
            for i in rng.choice(n, n // 5, replace=False):
                st.delete(int(ids[i]))
            st.flush()
            st.close()
            sids, svecs = po.read_vector_store(dd, d)
            metric = [DM.Cosine, DM.Euclidean, DM.DotProduct][int(rng.integers(0, 3))]
            ix = va.HnswIndex(d, metric)
            assert ix.upload_vector_store(dd) == len(sids) == len(ix), tag
            if len(sids):
                Q = make("normal", 4, d)
                k = min(7, len(sids))
                gi, gs, gc = ix.search_batch_brute_force(Q, k)
                mode = po.MODE_M if ix.sweep_arith_mode(k) == "M" else po.MODE_C
                ei, es = po.scan_topk(PO[metric], svecs, Q, k, mode)
                assert np.array_equal(gi[:, :k], sids[ei.astype(np.int64)]) and np.array_equal(bits(gs[:, :k]), bits(es)), tag
            ix.close()
        finally:
            shutil.rmtree(dd, ignore_errors=True)
    else:  # vacuum: rebuild over the live rows == the oracle's batched build of those rows with HnswParams::auto
        n, d = int(rng.choice([50, 300, 900])), int(rng.choice([16, 48, 96]))
        metric = [DM.Cosine, DM.Euclidean, DM.DotProduct, DM.Hamming][int(rng.integers(0, 4))]
        rows = make(kind, n, d, metric)
        ids = rng.permutation(n).astype(np.uint64) * 2 + 5
        ix = va.HnswIndex(d, metric, va.HnswParams(8, 40, n))
        ix.insert_batch_parallel([(int(ids[i]), rows[i]) for i in range(n)], 64)
        dead = rng.choice(n, n // 3, replace=False)
        for i in dead:
            ix.remove(int(ids[i]))
        live = np.ones(n, bool)
        live[dead] = False
        assert ix.vacuum() == int(live.sum()), tag
        g = po.NativeHnsw(d, PO[metric], 24, 300, po.MODE_C)      # HnswParams::auto(dim <= 256)
        g.set_build_tie(po.TIE_CANONICAL)
        g.build_batched(rows[live], 2048)
        nl, ml, ep = ix.graph_info()
        assert (nl, ml, ep) == (g.num_layers, g.max_layer, g.entry_point), tag
        m = int(live.sum())
        for layer in range(g.num_layers):
            for node in range(m):
                assert ix.neighbors(layer, node) == g.neighbors(layer, node), tag + f" layer {layer} node {node}"
        ix.close()
    stats[what] += 1
    if it % 20 == 0:
        print(f"[fuzz-misc] {it} cases ok ({stats})", flush=True)
print(f"[fuzz-misc] done: {it} cases, all equal to the oracle ({stats})")
