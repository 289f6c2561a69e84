"""The two callers that ride on the index (VERDICT r2 "Missing 5"): the reference's second `impl VectorIndex`
(NativeHnswIndex, index/hnsw/native_index.rs:225-249,403-427) and the index side of Collection::search_with_filter
(collection/search/vector.rs:164-235: over-fetch max(4 k, k + 10), post-filter, first k, metric order)."""
import numpy as np
import pytest

import velesdb_amd as va
from oracle import pyoracle as po
from velesdb_amd import DistanceMetric as DM
from velesdb_amd import SearchQuality as SQ

pytestmark = pytest.mark.gpu

PO_METRIC = {DM.Cosine: po.COSINE, DM.Euclidean: po.EUCLIDEAN, DM.DotProduct: po.DOT}


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def pair(tmp_path, cls, rows, metric, M=16, efc=100):
    g = po.NativeHnsw(rows.shape[1], PO_METRIC[metric], M, efc, po.MODE_C)
    for v in rows:
        g.insert(v)
    g.file_dump(str(tmp_path), "native_hnsw")
    ix = cls(rows.shape[1], metric, va.HnswParams(M, efc, len(rows)))
    ix.load_reference_files(str(tmp_path), "native_hnsw")
    return g, ix


def oracle_walk(g, metric, q, k, ef):
    oid, od = g.search(q, k, ef, po.TIE_CANONICAL)
    return [(int(i), float(np.float32(po.transform_score(PO_METRIC[metric], float(d))))) for i, d in zip(oid, od)]


def test_native_hnsw_index_always_walks_the_graph(tmp_path):
    # 80 vectors: HnswIndex answers by the exact scan (search.rs:59-66, raw scores), NativeHnswIndex by the walk (transform_score)
    rng = np.random.default_rng(41)
    rows = rng.standard_normal((80, 64)).astype(np.float32)
    g, nix = pair(tmp_path, va.NativeHnswIndex, rows, DM.Cosine)
    q = rng.standard_normal(64).astype(np.float32)
    for quality in (SQ.Fast, SQ.Balanced, SQ.Accurate, SQ.Perfect, SQ.Custom(37)):
        want = oracle_walk(g, DM.Cosine, q, 10, quality.ef_search(10))
        got = nix.search_with_quality(q, 10, quality)
        assert [i for i, _ in got] == [i for i, _ in want]
        assert np.array_equal(bits([s for _, s in got]), bits([s for _, s in want]))
        assert all(0.0 <= s <= 1.0 for _, s in got)  # clamp(1 - d, 0, 1): never the raw similarity of the exact scan
    assert nix.search(q, 10) == nix.search_with_quality(q, 10, SQ.Balanced)  # native_index.rs:225-227
    _, hix = pair(tmp_path, va.HnswIndex, rows, DM.Cosine)
    gt, _ = po.scan_topk(po.COSINE, rows, q.reshape(1, -1), 10, po.MODE_M if hix.sweep_arith_mode(10) == "M" else po.MODE_C)
    assert [i for i, _ in hix.search(q, 10)] == gt[0].tolist()  # (the shortcut HnswIndex takes and NativeHnswIndex does not)


def test_native_hnsw_index_remove_drops_after_the_cut(tmp_path):
    rng = np.random.default_rng(42)
    rows = rng.standard_normal((400, 32)).astype(np.float32)
    g, nix = pair(tmp_path, va.NativeHnswIndex, rows, DM.Euclidean, 8, 50)
    q = rows[7] + 0.01
    full = nix.search(q, 10)
    assert nix.remove(full[2][0]) and not nix.remove(10**9)
    assert nix.search(q, 10) == [r for r in full if r[0] != full[2][0]]  # native_index.rs:241-247: fewer than k
    assert nix.len() == 399


@pytest.mark.parametrize("metric", [DM.Cosine, DM.Euclidean])
def test_search_filtered_over_fetches_then_cuts(tmp_path, metric):
    rng = np.random.default_rng(43)
    rows = rng.standard_normal((1500, 48)).astype(np.float32)
    g, ix = pair(tmp_path, va.HnswIndex, rows, metric)
    q = rng.standard_normal(48).astype(np.float32)
    for k in (1, 3, 10, 25):
        ck = max(4 * k, k + 10)  # vector.rs:182
        cand = oracle_walk(g, metric, q, ck, SQ.Balanced.ef_search(ck))  # index.search(query, candidates_k)
        for keep in (lambda i: i % 3 != 0, lambda i: i % 7 == 1, lambda i: True, lambda i: False):
            want = [c for c in cand if keep(c[0])][:k]
            want = sorted(want, key=lambda c: -c[1] if metric == DM.Cosine else c[1])  # (the walk's order already: stable)
            got = ix.search_filtered(q, k, keep)
            assert [i for i, _ in got] == [i for i, _ in want]
            assert np.array_equal(bits([s for _, s in got]), bits([s for _, s in want]))
            assert len(got) <= k
