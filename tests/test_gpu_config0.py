"""BASELINE configs[0] as a parity case: the reference's own criterion workload (benches/hnsw_benchmark.rs:14-18,139-163):
10 000 x 768 vectors from `generate_vector(dim, seed) = ((sin(0.1 seed + 0.01 i) + 1) / 2)`, cosine,
HnswParams::auto(768) = M 32 / ef_construction 400, sequential inserts, query seed 99 999, k = 10 / 50 / 100 through
`index.search` (Balanced: ef = max(128, 4k)) and ef_search = 64 (Fast).  The generator is a one-parameter family with
period 62.8 seeds — near-duplicates and near-ties everywhere, the hardest input for link-for-link parity."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM, SQ = va.DistanceMetric, va.SearchQuality


def generate_vector(dim, seed):
    i = np.arange(dim, dtype=np.float32)
    return ((np.sin(np.float32(seed) * np.float32(0.1) + i * np.float32(0.01)) + np.float32(1.0)) / np.float32(2.0)).astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_config0_reference_bench_workload():
    n, dim = 10_000, 768
    rows = np.stack([generate_vector(dim, s) for s in range(n)])
    oix = po.HnswIndex(dim, po.COSINE, po.MODE_C)                 # HnswParams::auto(768): M 32, ef_construction 400
    oix.graph.set_build_tie(po.TIE_CANONICAL)
    ix = va.HnswIndex(dim, DM.Cosine)                             # same auto parameters
    assert (ix.params.max_connections, ix.params.ef_construction) == (32, 400)
    for s in range(n):
        assert oix.insert(s, rows[s])
    assert ix.insert_batch_sequential([(s, rows[s]) for s in range(n)]) == n    # VectorIndex::insert, one by one
    g = oix.graph
    nl, ml, ep = ix.graph_info()
    assert (nl, ml, ep) == (g.num_layers, g.max_layer, g.entry_point)
    for layer in range(g.num_layers):
        for node in range(n):
            assert ix.neighbors(layer, node) == g.neighbors(layer, node), (layer, node)
    query = generate_vector(dim, 99_999)
    for k in (10, 50, 100):                                       # hnsw_search_latency: index.search(&query, k)
        r = ix.search(query, k)
        eid, esc = oix.search(query, k, po.TIE_CANONICAL)
        assert [x[0] for x in r] == eid.tolist() and np.array_equal(bits([x[1] for x in r]), bits(esc)), k
    r = ix.search_with_quality(query, 10, SQ.Fast)                # ef_search = 64
    eid, esc = oix.search_with_quality(query, 10, po.Q_FAST, 0, po.TIE_CANONICAL)
    assert [x[0] for x in r] == eid.tolist()
    # hnsw_search_throughput: 100 queries, seeds 100 000 ..; recall gate of the reference's bench (>= 0.95, :324-328)
    qs = np.stack([generate_vector(dim, 100_000 + i) for i in range(100)])
    res = ix.search_batch_parallel(qs, 10, SQ.Balanced)
    bi, bs, bc = oix.search_batch(qs, 10, po.Q_BALANCED, 0, po.TIE_CANONICAL, nthreads=8)
    gt, gts, _ = ix.search_batch_brute_force(qs, 10)
    rec = 0.0
    for i in range(100):
        assert [x[0] for x in res[i]] == bi[i, :bc[i]].tolist(), i
        # tie-aware recall: this generator produces near-duplicate vectors every 62.8 seeds, so rank 10 sits inside a
        # cluster of (almost) equal scores — a hit is a result whose EXACT similarity reaches the 10th best one
        # (graph scores are clamp(1 - d, 0, 1) of the same cosine; 1e-6 covers the two arithmetic modes)
        rec += sum(1 for _, sc in res[i] if sc >= gts[i, 9] - 1e-6) / 10
    assert rec / 100 >= 0.95
    ix.close()
