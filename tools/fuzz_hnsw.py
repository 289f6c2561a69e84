#!/usr/bin/env python3
"""GPU fuzz of graph construction + traversal against the oracle (not part of the test-suite): random sizes, all five
metrics, tie-heavy data, sequential and batch-synchronous builds, random ef / k.  Bar: every adjacency list of every
node on every layer, entry point and max layer equal to the oracle's graph; traversal ids equal, distances bit-equal.

    python tools/fuzz_hnsw.py --seconds 240 --seed 1
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
from oracle import pyoracle as po  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--seconds", type=float, default=240)
p.add_argument("--seed", type=int, default=1)
p.add_argument("--big", action="store_true", help="graphs of 5K-20K nodes, batched GPU builds, bigger beams")
a = p.parse_args()
rng = np.random.default_rng(a.seed)
DM = va.DistanceMetric
SQ = va.SearchQuality
PO = {DM.Cosine: po.COSINE, DM.Euclidean: po.EUCLIDEAN, DM.DotProduct: po.DOT, DM.Hamming: po.HAMMING, DM.Jaccard: po.JACCARD}


def make(kind, n, d, metric):
    if metric in (DM.Hamming, DM.Jaccard):
        pr = 0.6915 if kind != "sparse" else 0.95
        return (rng.random((n, d)) > pr).astype(np.float32)
    if kind == "dups":
        base = rng.standard_normal((max(1, n // 8), d)).astype(np.float32)
        return base[rng.integers(0, base.shape[0], n)]
    if kind == "ints":
        return rng.integers(-2, 3, size=(n, d)).astype(np.float32)
    return rng.standard_normal((n, d)).astype(np.float32)


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


t_end = time.time() + a.seconds
it = 0
while time.time() < t_end:
    it += 1
    metric = [DM.Cosine, DM.Euclidean, DM.DotProduct, DM.Hamming, DM.Jaccard][int(rng.integers(0, 5))]
    n = int(rng.choice([1, 2, 17, 101, 300, 700]))
    d = int(rng.choice([3, 16, 33, 64, 200]))
    M = int(rng.choice([2, 4, 8, 16]))
    efc = int(rng.choice([10, 40, 100]))
    kind = str(rng.choice(["normal", "dups", "ints", "sparse"]))
    mb = [None, 1, 7, 64][int(rng.integers(0, 4))]
    if a.big:
        n = int(rng.choice([5000, 20000]))
        d = int(rng.choice([32, 128, 256]))
        M = int(rng.choice([8, 16]))
        efc = int(rng.choice([40, 100]))
        mb = [64, 256, 1024][int(rng.integers(0, 3))]
        kind = str(rng.choice(["normal", "normal", "dups"]))
    rows = make(kind, n, d, metric)
    tag = f"it={it} {metric.name} n={n} d={d} M={M} efc={efc} {kind} batch={mb}"
    g = po.NativeHnsw(d, PO[metric], M, efc, po.MODE_C)
    g.set_build_tie(po.TIE_CANONICAL)
    ix = va.HnswIndex(d, metric, va.HnswParams(M, efc, n))
    if mb is None:
        for i, v in enumerate(rows):
            g.insert(v)
            ix.insert(i, v)
    else:
        g.build_batched(rows, mb)
        ix.upload(np.arange(n), rows)
        ix.build_graph(mb)
    nl, ml, ep = ix.graph_info()
    assert (nl, ml, ep) == (g.num_layers, g.max_layer, g.entry_point), tag
    for layer in range(g.num_layers):
        for node in range(n):
            assert ix.neighbors(layer, node) == g.neighbors(layer, node), f"{tag} layer {layer} node {node}"
    nq = int(rng.integers(1, 6)) if not a.big else 40
    qs = make(kind, nq, d, metric)
    k = int(rng.choice([1, 5, 10, 30]))
    ef = int(rng.choice([1, 10, 64, 200, 300])) if not a.big else int(rng.choice([64, 128, 200, 400]))
    res = ix.search_batch_parallel(qs, k, SQ.Custom(ef))
    for q, r in zip(qs, res):
        oid, od = g.search(q, k, max(ef, k), po.TIE_CANONICAL)
        assert [x[0] for x in r] == oid.tolist(), tag + f" k={k} ef={ef}"
    # dual-precision traversal (int8 graph walk + exact f32 re-rank) on the same graph: integer distances, bit-exact
    if metric in (DM.Cosine, DM.Euclidean, DM.DotProduct) and n >= 2 and rng.random() < 0.5:
        ix.train_quantizer()
        sq = po.ScalarQuantizer(rows[:1000])
        codes = sq.quantize(rows)
        k8 = int(rng.choice([1, 5, 10, 30]))
        ef8 = int(rng.choice([8, 64, 200, 300]))
        r8 = ix.search_batch_int8(qs, k8, ef8)
        for qi, q in enumerate(qs):
            oid, od, _, _ = po.dual_search_int8(g, sq, codes, q, k8, ef8, 4, po.TIE_CANONICAL)
            osc = np.array([po.transform_score(PO[metric], float(x)) for x in od], dtype=np.float32)
            assert [x[0] for x in r8[qi]] == oid.tolist(), tag + f" int8 k={k8} ef={ef8}"
            assert np.array_equal(bits([x[1] for x in r8[qi]]), bits(osc)), tag + " int8 scores"
    ix.close()
    if it % (20 if not a.big else 2) == 0:
        print(f"[fuzz-hnsw] {it} cases ok", flush=True)
print(f"[fuzz-hnsw] done: {it} cases, every graph link for link equal to the oracle, traversal ids equal")
