"""GPU parity tests for the graph path: NativeHnsw::search (greedy descent + layer-0 search_layer) through
the C ABI, on graphs built by the oracle (sequential insert, canonical arithmetic = oracle mode C) and
handed over in the reference's on-disk format v1 (native_hnsw.vectors / .graph).

Bar: ids, ranks AND distances bit-identical to the oracle's canonical run (TIE_CANONICAL), the kernel's
distance-evaluation / expansion counters equal to the oracle's, for every metric; Hamming/Jaccard
(integer distances, many exact ties) included.  Against mode R (the reference's own summation order)
the tie-aware tolerance rule of SURVEY.md §8(c) applies."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM = va.DistanceMetric
SQ = va.SearchQuality
PO_METRIC = {DM.Cosine: po.COSINE, DM.Euclidean: po.EUCLIDEAN, DM.DotProduct: po.DOT, DM.Hamming: po.HAMMING,
             DM.Jaccard: po.JACCARD}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def build_pair(tmp_path, rows, metric, M, efc, mode=po.MODE_C):
    """oracle graph (sequential insert) -> reference files -> GPU index"""
    n, d = rows.shape
    g = po.NativeHnsw(d, PO_METRIC[metric], M, efc, mode)
    for v in rows:
        g.insert(v)
    g.file_dump(str(tmp_path), "native_hnsw")
    ix = va.HnswIndex(d, metric, va.HnswParams(M, efc, max(n, 1)))
    ix.load_reference_files(str(tmp_path), "native_hnsw")
    return g, ix


def check_batch(g, ix, metric, queries, k, ef, expect_stats=True):
    res = ix.search_batch_parallel(queries, k, SQ.Custom(ef))
    nd_gpu, ne_gpu = ix.last_search_stats()
    nd = ne = 0
    for qi, q in enumerate(queries):
        oid, od = g.search(q, k, ef, po.TIE_CANONICAL)
        a, b = po.NativeHnsw.last_stats()
        nd += a
        ne += b
        gid = np.array([r[0] for r in res[qi]], dtype=np.uint64)
        gsc = np.array([r[1] for r in res[qi]], dtype=np.float32)
        osc = np.array([po.transform_score(PO_METRIC[metric], float(x)) for x in od], dtype=np.float32)
        assert np.array_equal(gid, oid), f"query {qi}: ids/ranks differ\n gpu {gid}\n ora {oid}"
        assert np.array_equal(bits(gsc), bits(osc)), f"query {qi}: scores differ bitwise"
    if expect_stats:
        assert (nd_gpu, ne_gpu) == (nd, ne), "distance-evaluation / expansion counters differ from the oracle"


# ---------------------------------------------------------------- reference fixtures
def test_ramp_graph_euclidean(tmp_path):
    # native/graph_tests.rs:10-30: v_i[j] = 32 i + j, 100 x 32, M16 efc100, query v_0, k10 ef50
    rows = np.array([[32.0 * i + j for j in range(32)] for i in range(100)], dtype=np.float32)
    g, ix = build_pair(tmp_path, rows, DM.Euclidean, 16, 100)
    res = ix.search_batch_parallel(rows[:1], 10, SQ.Custom(50))[0]
    assert res[0][0] == 0 and len(res) <= 10
    check_batch(g, ix, DM.Euclidean, rows[[0, 17, 50, 99]], 10, 50)


@pytest.mark.parametrize("name,n,dim,M,efc,k,ef,fn", [
    ("A", 100, 128, 16, 100, 10, 50, lambda i, j: np.sin(0.01 * (i + j))),            # native/tests.rs:10-29
    ("B", 200, 128, 16, 100, 10, 128, lambda i, j: np.sin(0.001 * (128 * i + j))),    # native/tests.rs:32-91
    ("C", 500, 128, 32, 200, 10, 100, lambda i, j: np.sin(0.01 * (127 * i + j))),     # graph_tests.rs:169-199
])
def test_sinusoid_graphs_cosine(tmp_path, name, n, dim, M, efc, k, ef, fn):
    i, j = np.meshgrid(np.arange(n), np.arange(dim), indexing="ij")
    rows = fn(i.astype(np.float64), j.astype(np.float64)).astype(np.float32)
    g, ix = build_pair(tmp_path, rows, DM.Cosine, M, efc)
    qs = rows[:: max(1, n // 5)][:5]
    check_batch(g, ix, DM.Cosine, qs, k, ef)
    # the reference's own assertions
    res = ix.search_batch_parallel(qs, k, SQ.Custom(ef))
    for r in res:
        assert len(r) >= 5
        sims = [s for _, s in r]
        assert sims == sorted(sims, reverse=True)  # similarity, best first
    if name == "B":  # mean recall >= 0.8 vs exact cosine (native/tests.rs:60-91)
        rec = []
        for q, r in zip(qs, res):
            gt, _ = po.scan_topk(po.COSINE, rows, q.reshape(1, -1), k, po.MODE_C)
            rec.append(len(set(gt[0].tolist()) & {x for x, _ in r}) / k)
        assert np.mean(rec) >= 0.8


# ---------------------------------------------------------------- random data, every metric
@pytest.mark.parametrize("metric", [DM.Cosine, DM.Euclidean, DM.DotProduct, DM.Hamming, DM.Jaccard])
@pytest.mark.parametrize("n,dim,M,efc", [(1500, 768, 16, 100), (1200, 96, 8, 60), (400, 5, 4, 40)])
def test_random_all_metrics(tmp_path, metric, n, dim, M, efc):
    rng = np.random.default_rng(1234 + n + dim)
    if metric in (DM.Hamming, DM.Jaccard):
        rows = (rng.random((n, dim)) > 0.6915).astype(np.float32)
        qs = (rng.random((24, dim)) > 0.6915).astype(np.float32)
    else:
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((24, dim)).astype(np.float32)
    g, ix = build_pair(tmp_path, rows, metric, M, efc)
    check_batch(g, ix, metric, qs, 10, 64)
    check_batch(g, ix, metric, qs[:6], 1, 16)     # ef < list chunk, k = 1
    check_batch(g, ix, metric, qs[:6], 50, 200)   # ef > 128, k > ef/4


def test_walk_prefetch_is_a_measure_not_a_result(tmp_path):
    """The latency-mode walk asks for the neighbour list of the candidate it predicts to pop next together with the current one's
    (hnsw_kernels.hip pf_ids).  Results and the reference's counters are the oracle's either way (check_batch); the hit counter is
    bounded by the expansions, and non-zero whenever that kernel ran with the prediction on (default: corpora beyond the Infinity
    Cache only; tests/test_gpu_switches.py runs this file with it forced on and off)."""
    import os
    rng = np.random.default_rng(99)
    rows = rng.standard_normal((1500, 768)).astype(np.float32)
    g, ix = build_pair(tmp_path, rows, DM.Cosine, 16, 100)
    qs = rng.standard_normal((24, 768)).astype(np.float32)
    check_batch(g, ix, DM.Cosine, qs, 10, 64)
    _, ne = ix.last_search_stats()
    hits = ix.last_prefetch_hits()
    assert 0 <= hits <= ne
    lat_on = os.environ.get("VELESDB_HNSW_LATENCY_MODE", "1") != "0"
    pf = os.environ.get("VELESDB_HNSW_PREFETCH_IDS")  # unset: only over corpora beyond the Infinity Cache (not this one)
    if lat_on and pf == "1":
        assert hits > 0
    if pf != "1":
        assert hits == 0
    qmany = rng.standard_normal((600, 768)).astype(np.float32)  # more queries than CUs: the throughput kernel (no prediction)
    ix.search_batch_parallel(qmany, 10, SQ.Custom(64))
    assert ix.last_prefetch_hits() == 0


def test_many_queries_more_than_slots(tmp_path):
    # nq > resident slots: blocks loop over the batch and must leave the visited bitmaps clean
    rng = np.random.default_rng(7)
    rows = rng.standard_normal((3000, 256)).astype(np.float32)
    g, ix = build_pair(tmp_path, rows, DM.Cosine, 16, 100)
    qs = rng.standard_normal((2500, 256)).astype(np.float32)
    res1 = ix.search_batch_parallel(qs, 10, SQ.Custom(64))
    res2 = ix.search_batch_parallel(qs, 10, SQ.Custom(64))
    assert res1 == res2
    for qi in rng.choice(len(qs), 40, replace=False):
        oid, od = g.search(qs[qi], 10, 64, po.TIE_CANONICAL)
        assert [r[0] for r in res1[qi]] == oid.tolist()


def test_auto_mode_and_quality_presets(tmp_path):
    # search.rs:59-94: len <= 100 -> exact scan with RAW scores; otherwise graph with transform_score
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((300, 64)).astype(np.float32)
    g, ix = build_pair(tmp_path, rows, DM.Cosine, 16, 100)
    q = rng.standard_normal(64).astype(np.float32)
    for quality, ef in [(SQ.Fast, 64), (SQ.Balanced, 128), (SQ.Accurate, 512)]:
        r = ix.search_with_quality(q, 10, quality)
        oid, od = g.search(q, 10, ef, po.TIE_CANONICAL)
        assert [x[0] for x in r] == oid.tolist()
        assert all(0.0 <= s <= 1.0 for _, s in r)  # clamp(1-d, 0, 1)
    r = ix.search_with_quality(q, 10, SQ.Perfect)   # brute force: raw similarity, may be negative
    gt, gs = po.scan_topk(po.COSINE, rows, q.reshape(1, -1), 10,
                          po.MODE_M if ix.sweep_arith_mode(10) == "M" else po.MODE_C)
    assert [x[0] for x in r] == gt[0].tolist()
    assert np.array_equal(bits([s for _, s in r]), bits(gs[0]))
    assert ix.search(q, 10) == ix.search_with_quality(q, 10, SQ.Balanced)  # trait_impl.rs:38-42


def test_soft_delete_filters_after_cut(tmp_path):
    # search.rs:86-91: removed ids are traversed, then dropped after the top-k cut => fewer than k results
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((500, 32)).astype(np.float32)
    g, ix = build_pair(tmp_path, rows, DM.Euclidean, 8, 50)
    q = rows[10] + 0.01
    full = ix.search_batch_parallel(q.reshape(1, -1), 10, SQ.Custom(64))[0]
    victims = [full[0][0], full[3][0], full[9][0]]
    for v in victims:
        assert ix.remove(v)
    after = ix.search_batch_parallel(q.reshape(1, -1), 10, SQ.Custom(64))[0]
    assert [r for r in full if r[0] not in victims] == after
    assert len(after) == 7 and ix.len() == 497


def test_mode_r_tie_aware_tolerance(tmp_path):
    # same graph, reference summation order (mode R) vs GPU canonical order: ids equal wherever adjacent
    # reference distances are further apart than 1e-5 relative; distances within 1e-5 relative
    rng = np.random.default_rng(11)
    rows = rng.standard_normal((2000, 768)).astype(np.float32)
    g, ix = build_pair(tmp_path, rows, DM.Cosine, 16, 100, mode=po.MODE_R)
    qs = rng.standard_normal((16, 768)).astype(np.float32)
    res = ix.search_batch_parallel(qs, 10, SQ.Custom(128))
    same = 0
    for q, r in zip(qs, res):
        oid, od = g.search(q, 10, 128, po.TIE_REFERENCE)
        osc = np.array([po.transform_score(po.COSINE, float(x)) for x in od], dtype=np.float32)
        gid = [x[0] for x in r]
        gsc = np.array([x[1] for x in r], dtype=np.float32)
        if gid == oid.tolist():
            same += 1
            assert np.all(np.abs(gsc - osc) <= 1e-5 * np.maximum(np.abs(osc), 1e-3))
    # traversal decisions can legitimately flip on sub-ulp differences; it must be rare
    assert same >= 14


def test_empty_graph_and_unbuilt_rows(tmp_path):
    ix = va.HnswIndex(16, DM.Cosine, va.HnswParams(8, 50, 100))
    assert ix.search_batch_parallel(np.zeros((2, 16), np.float32), 5, SQ.Fast) == [[], []]
    ix.upload(np.arange(200), np.random.default_rng(0).standard_normal((200, 16)).astype(np.float32))
    with pytest.raises(va.VelesHipError):  # rows uploaded without a graph: HNSW mode must fail loudly
        ix.search_batch_parallel(np.zeros((1, 16), np.float32), 5, SQ.Fast)


@pytest.mark.parametrize("metric", [DM.Cosine, DM.Euclidean, DM.DotProduct, DM.Hamming, DM.Jaccard])
def test_search_with_rerank(tmp_path, metric):
    # search.rs:118-160 / 297-350: candidates from the graph (k = rerank_k), raw exact re-scoring, stable sort
    n, dim = 1500, 96
    rng = np.random.default_rng(31)
    if metric in (DM.Hamming, DM.Jaccard):
        rows = (rng.random((n, dim)) > 0.6915).astype(np.float32)
        qs = (rng.random((6, dim)) > 0.6915).astype(np.float32)
    else:
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((6, dim)).astype(np.float32)
    oix = po.HnswIndex(dim, PO_METRIC[metric], po.MODE_C, 8, 60)
    for i, v in enumerate(rows):
        oix.insert(i, v)
    g = oix.graph
    g.file_dump(str(tmp_path), "native_hnsw")
    ix = va.HnswIndex(dim, metric, va.HnswParams(8, 60, n))
    ix.load_reference_files(str(tmp_path), "native_hnsw")
    for q in qs:
        for k, rk, quality, oq, oef in [(10, 50, SQ.Accurate, po.Q_ACCURATE, 0), (5, 5, SQ.Fast, 0, 0),
                                        (10, 100, SQ.Custom(300), 4, 300), (20, 7, SQ.Balanced, po.Q_BALANCED, 0)]:
            r = ix.search_with_rerank_quality(q, k, rk, quality)
            eid, esc = oix.search_with_rerank_quality(q, k, rk, oq, oef, po.TIE_CANONICAL)
            assert [x[0] for x in r] == eid.tolist(), (metric, k, rk)
            assert np.array_equal(bits([x[1] for x in r]), bits(esc))
    assert ix.search_with_rerank(qs[0], 10, 50) == ix.search_with_rerank_quality(qs[0], 10, 50, SQ.Accurate)


@pytest.mark.parametrize("metric,n,dim", [(DM.Cosine, 700, 64), (DM.Euclidean, 400, 32), (DM.Hamming, 300, 96)])
def test_search_multi_entry_follows_the_reference_stream(tmp_path, metric, n, dim):
    """NativeHnsw::search_multi_entry (native/graph.rs:288-348): the descent's result plus up to min(num_probes, 4) - 1 nodes
    drawn from the graph's own xorshift stream, one search_layer from all of them.  A batch on the GPU == the same queries
    one after the other on the oracle (ids, score bits, counters), for every num_probes; afterwards both level streams stand
    at the same place: the next insert draws the same level and links the same neighbours."""
    rng = np.random.default_rng(n + dim)
    rows = (rng.random((n + 1, dim)) > 0.6).astype(np.float32) if metric == DM.Hamming else rng.standard_normal((n + 1, dim)).astype(np.float32)
    _, ix = build_pair(tmp_path, rows[:n], metric, 8, 60)
    g = po.NativeHnsw.file_load(str(tmp_path), "native_hnsw", PO_METRIC[metric], po.MODE_C)   # fresh stream, like the loaded GPU index
    g.set_build_tie(po.TIE_CANONICAL)
    g.dim = dim   # (file_load leaves it open)
    k, ef = 7, 40
    for probes in (1, 2, 3, 4, 9):
        Q = (rng.random((11, dim)) > 0.6).astype(np.float32) if metric == DM.Hamming else rng.standard_normal((11, dim)).astype(np.float32)
        s0 = g.rng_state()
        gid, gsc, gcnt = ix.search_multi_entry(Q, k, ef, probes)
        nd_gpu, ne_gpu = ix.last_search_stats()
        nd = ne = 0
        for qi in range(Q.shape[0]):
            oid, od = g.search_multi_entry(Q[qi], k, ef, probes, po.TIE_CANONICAL)
            a, b = po.NativeHnsw.last_stats()
            nd += a
            ne += b
            osc = np.array([po.transform_score(PO_METRIC[metric], float(x)) for x in od], dtype=np.float32)
            assert gcnt[qi] == len(oid), (probes, qi)
            assert np.array_equal(gid[qi, :gcnt[qi]], oid), (probes, qi, gid[qi], oid)
            assert np.array_equal(bits(gsc[qi, :gcnt[qi]]), bits(osc)), (probes, qi)
        assert (nd_gpu, ne_gpu) == (nd, ne), (probes, nd_gpu, nd, ne_gpu, ne)
        assert (g.rng_state() != s0) == (probes > 1)   # the stream moves exactly when entry points are drawn
    one = ix.search_multi_entry(Q[0], k, ef, 1)      # num_probes = 1 is the plain search
    ref = ix._search_raw(Q[:1], k, ef, va.MODE_HNSW)
    assert np.array_equal(one[0], ref[0]) and np.array_equal(bits(one[1]), bits(ref[1]))
    # both level streams stand at the same place: the next (sequential) insert links the same neighbours on every layer
    g.insert(rows[n])
    ix.insert(n, rows[n])
    nl, _, _ = ix.graph_info()
    assert nl == g.num_layers
    for layer in range(g.num_layers):
        assert ix.neighbors(layer, n) == g.neighbors(layer, n), layer
    # NativeHnsw's raw ef_search (graph.rs:343): no max(ef, k) — ef < k gives at most ef results; fewer than the entry points
    # (ef < 4 with several probes) gives as many as there are entry points (uncut `results`, graph.rs:463-468); ef = 0 acts as 1
    for k2, ef2, probes in ((7, 3, 1), (7, 5, 2), (2, 3, 3), (9, 2, 4), (9, 1, 9), (5, 0, 1), (5, 0, 4), (3, 1, 1)):
        gid, gsc, gcnt = ix.search_multi_entry(Q, k2, ef2, probes)
        for qi in range(Q.shape[0]):
            oid, od = g.search_multi_entry(Q[qi], k2, ef2, probes, po.TIE_CANONICAL)
            osc = np.array([po.transform_score(PO_METRIC[metric], float(x)) for x in od], dtype=np.float32)
            assert gcnt[qi] == len(oid) <= max(ef2, min(probes, 4), 1), (k2, ef2, probes, qi, gcnt[qi], len(oid))
            assert np.array_equal(gid[qi, :gcnt[qi]], oid), (k2, ef2, probes, qi, gid[qi], oid)
            assert np.array_equal(bits(gsc[qi, :gcnt[qi]]), bits(osc)), (k2, ef2, probes, qi)
    # ... and the ordinary search entry points keep SearchQuality::Custom's max(ef, k) (params.rs:317)
    r_small = ix._search_raw(Q[:1], 7, 3, va.MODE_HNSW)
    r_k = ix._search_raw(Q[:1], 7, 7, va.MODE_HNSW)
    assert np.array_equal(r_small[0], r_k[0]) and r_small[2][0] == 7
    ix.close()


# ---------------------------------------------------------------- the declared tie rule against the reference's own order, on the GPU
@pytest.mark.parametrize("metric", [DM.Hamming, DM.Jaccard])
def test_gpu_canonical_order_vs_reference_tie_order_on_integer_distances(tmp_path, metric):
    """SURVEY 8(a) note 8 / 8(c) rule 3: among EQUAL distances the reference's order is an artefact of its heaps' backing arrays;
    the ABI declares (distance, node id) ascending.  Integer distances (Hamming; Jaccard's few distinct ratios) tie in most result
    lists, so here the rule is held on the GPU itself against the oracle run in the REFERENCE's tie order (Rust BinaryHeap restated,
    VO_TIE_REFERENCE) over the same graph: the distance list is identical, ids are identical wherever a distance is unique in the
    list, inside a group of equal distances the GPU's ids ascend, and a group that does not touch the cut holds the same id set.
    (tests/test_oracle_graph.py::test_hamming_canonical_vs_reference_tie_order states the same rule CPU-side.)"""
    rng = np.random.default_rng(31 + int(metric))
    rows = (rng.random((900, 64)) > 0.6).astype(np.float32)
    g, ix = build_pair(tmp_path, rows, metric, 8, 60, mode=po.MODE_R)
    qs = (rng.random((48, 64)) > 0.6).astype(np.float32)
    k, ef = 10, 48
    res = ix.search_batch_parallel(qs, k, SQ.Custom(ef))
    tied_lists = 0
    for qi, q in enumerate(qs):
        ids_r, d_r = g.search(q, k, ef, po.TIE_REFERENCE)
        gid = np.array([r[0] for r in res[qi]], dtype=np.uint64)
        gsc = np.array([r[1] for r in res[qi]], dtype=np.float32)
        exp = np.array([po.transform_score(PO_METRIC[metric], float(x)) for x in d_r], dtype=np.float32)
        assert np.array_equal(bits(gsc), bits(exp)), f"query {qi}: the distance list differs from the reference-order run"
        n = len(gid)
        i = 0
        while i < n:
            j = i + 1
            while j < n and d_r[j] == d_r[i]:
                j += 1
            if j - i == 1:
                assert j == n or gid[i] == ids_r[i], f"query {qi}: ids differ outside a tie group at rank {i}"
            else:
                tied_lists += 1
                assert np.all(np.diff(gid[i:j].astype(np.int64)) > 0), f"query {qi}: ids inside a tie group do not ascend (canonical order)"
                if j < n:  # (a group cut by k may hold other members of the same distance on either side)
                    assert set(gid[i:j].tolist()) == set(ids_r[i:j].tolist()), f"query {qi}: a tie group inside the list holds different ids"
            i = j
    assert tied_lists > len(qs), "the data did not tie (fewer than one tie group per list): the test would be vacuous"
    ix.close()
