#!/usr/bin/env python3
"""GPU fuzz at the HnswIndex level against the oracle's HnswIndex mirror (not part of the test-suite): external ids,
duplicate inserts, soft deletes, every SearchQuality (incl. the <= 100-vector and Perfect shortcuts), rerank, batch
search — ids and score bits equal under the canonical tie order.

    python tools/fuzz_index.py --seconds 240 --seed 1
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
a = p.parse_args()
rng = np.random.default_rng(a.seed)
DM, SQ = va.DistanceMetric, va.SearchQuality
PO = {DM.Cosine: po.COSINE, DM.Euclidean: po.EUCLIDEAN, DM.DotProduct: po.DOT, DM.Hamming: po.HAMMING, DM.Jaccard: po.JACCARD}
QUAL = [(SQ.Fast, po.Q_FAST, 0), (SQ.Balanced, po.Q_BALANCED, 0), (SQ.Accurate, po.Q_ACCURATE, 0), (SQ.Perfect, po.Q_PERFECT, 0),
        (SQ.Custom(37), po.Q_CUSTOM, 37), (SQ.Custom(250), po.Q_CUSTOM, 250), (SQ.Custom(700), po.Q_CUSTOM, 700)]


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def make(kind, n, d, metric):
    if metric in (DM.Hamming, DM.Jaccard):
        return (rng.random((n, d)) > (0.6915 if kind != "sparse" else 0.93)).astype(np.float32)
    if kind == "dups":
        base = rng.standard_normal((max(1, n // 6), d)).astype(np.float32)
        return base[rng.integers(0, base.shape[0], n)]
    if kind == "ints":
        return rng.integers(-2, 3, size=(n, d)).astype(np.float32)
    return rng.standard_normal((n, d)).astype(np.float32)


t_end = time.time() + a.seconds
it = 0
while time.time() < t_end:
    it += 1
    metric = [DM.Cosine, DM.Euclidean, DM.DotProduct, DM.Hamming, DM.Jaccard][int(rng.integers(0, 5))]
    n = int(rng.choice([3, 60, 100, 101, 400, 1200, 2500]))
    d = int(rng.choice([4, 32, 96, 256, 768]))
    M = int(rng.choice([4, 8, 16]))
    efc = int(rng.choice([20, 60, 120]))
    kind = str(rng.choice(["normal", "dups", "ints", "sparse"]))
    rows = make(kind, n, d, metric)
    ids = rng.choice(1 << 30, n, replace=False).astype(np.uint64)
    tag = f"it={it} {metric.name} n={n} d={d} M={M} efc={efc} {kind}"
    oix = po.HnswIndex(d, PO[metric], po.MODE_C, M, efc)
    oix.graph.set_build_tie(po.TIE_CANONICAL)
    ix = va.HnswIndex(d, metric, va.HnswParams(M, efc, n))
    for i in range(n):
        assert oix.insert(int(ids[i]), rows[i])
    try:
        assert ix.insert_batch_sequential([(int(ids[i]), rows[i]) for i in range(n)]) == n
    except Exception:
        print("FAILED BUILD:", tag, flush=True)
        raise
    # duplicate ids are ignored by both
    assert ix.insert_batch_sequential([(int(ids[0]), rows[-1])]) == 0 and not oix.insert(int(ids[0]), rows[-1])
    live = np.ones(n, bool)
    if n > 5 and rng.random() < 0.6:
        for i in rng.choice(n, max(1, n // int(rng.choice([3, 10]))), replace=False):
            assert ix.remove(int(ids[i])) == oix.remove(int(ids[i]))
            live[i] = False
    assert len(ix) == len(oix) == int(live.sum()), tag
    sel = np.nonzero(live)[0]

    def exact(q, k):  # the exact scan in the arithmetic the library reports for it (matrix cores: mode M for cosine / dot)
        mode = po.MODE_M if ix.sweep_arith_mode(k) == "M" else po.MODE_C
        kk = min(k, len(sel))
        ei, es = po.scan_topk(PO[metric], rows[sel], q.reshape(1, -1), max(kk, 1), mode)
        return ids[sel[ei[0, :kk].astype(np.int64)]].tolist(), es[0, :kk]
    qs = make(kind, int(rng.integers(1, 5)), d, metric)
    for q in qs:
        k = int(rng.choice([1, 5, 10, 40, 120, 250]))
        gq, oq, oef = QUAL[int(rng.integers(0, len(QUAL)))]
        r = ix.search_with_quality(q, k, gq)
        if oq == po.Q_PERFECT or len(sel) <= 100:  # search.rs:68-77: the exact path
            eid, esc = exact(q, k)
        else:
            eid, esc = oix.search_with_quality(q, k, oq, oef, po.TIE_CANONICAL)
            eid = eid.tolist()
        assert [x[0] for x in r] == eid, tag + f" quality={oq}/{oef} k={k}"
        assert np.array_equal(bits([x[1] for x in r]), bits(esc)), tag + f" quality scores {oq}/{oef} k={k}"
        rk = int(rng.choice([k, 2 * k + 3, 100]))
        gq, oq, oef = QUAL[int(rng.integers(0, len(QUAL)))]
        r = ix.search_with_rerank_quality(q, k, rk, gq)
        if len(sel) <= 100:  # the candidate stage is the exact scan (search.rs:75-77): its scores are already the raw ones
            eid, esc = exact(q, min(k, rk))
        else:
            eid, esc = oix.search_with_rerank_quality(q, k, rk, oq, oef, po.TIE_CANONICAL)
            eid = eid.tolist()
        assert [x[0] for x in r] == eid, tag + f" rerank quality={oq}/{oef} k={k} rk={rk}"
        assert np.array_equal(bits([x[1] for x in r]), bits(esc)), tag + " rerank scores"
        r = ix.search_brute_force(q, k)
        eid, esc = exact(q, k)
        assert [x[0] for x in r] == eid and np.array_equal(bits([x[1] for x in r]), bits(esc)), tag + " brute force"
    if rng.random() < 0.35:  # HnswIndex::save / ::load round trip (external ids, soft deletes, graph): same answers
        import shutil
        import tempfile
        dd = tempfile.mkdtemp(prefix="vdb_fuzz_")
        try:
            ix.save(dd)
            ix2 = va.HnswIndex.load(dd)
            assert len(ix2) == len(ix) and ix2.graph_info() == ix.graph_info(), tag + " load"
            for q in qs:
                assert ix2.search_with_quality(q, 10, SQ.Custom(80)) == ix.search_with_quality(q, 10, SQ.Custom(80)), tag + " load search"
                assert ix2.search_brute_force(q, 7) == ix.search_brute_force(q, 7), tag + " load brute"
            ix2.close()
        finally:
            shutil.rmtree(dd, ignore_errors=True)
    res = ix.search_batch_parallel(qs, 10, SQ.Custom(64))
    bi, bs, bc = oix.search_batch(qs, 10, po.Q_CUSTOM, 64, po.TIE_CANONICAL)
    for qi in range(qs.shape[0]):
        assert [x[0] for x in res[qi]] == bi[qi, :bc[qi]].tolist(), tag + " batch"
    ix.close()
    if it % 20 == 0:
        print(f"[fuzz-index] {it} cases ok", flush=True)
print(f"[fuzz-index] done: {it} indexes, every search equal to the oracle's HnswIndex")
