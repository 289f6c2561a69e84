"""GPU parity tests for the exact path: distance sweep + top-k (HnswIndex::search_brute_force)
and DistanceEngine::batch_distance / GpuAccelerator::batch_*.  Everything goes through the C ABI
(velesdb_amd -> libvelesdb_hip.so) and is compared BIT-EXACTLY with the oracle's canonical mode
(oracle mode C shares the HIP kernels' summation order); mode R (reference order) is checked with
the north-star tolerance (1e-5 relative) and the tie-aware rule."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM = va.DistanceMetric
METRICS = [DM.Cosine, DM.Euclidean, DM.DotProduct, DM.Hamming, DM.Jaccard]


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def sweep_mode(metric, dim, k=10):
    """oracle mode the exact sweep must match bit for bit: M (matrix-core order) for Cosine / Dot when the engine
    is on (default), C (canonical lane-chain order) otherwise — asked from the library, not assumed"""
    ix = va.HnswIndex(dim, metric)
    m = ix.sweep_arith_mode(k)
    ix.close()
    return po.MODE_M if m == "M" else po.MODE_C


def rand_rows(rng, n, d, metric):
    if metric in (DM.Hamming, DM.Jaccard):
        return (rng.random((n, d)) > 0.6915).astype(np.float32)  # P(bit)=0.3085 like N(0,1)>0.5
    return rng.standard_normal((n, d)).astype(np.float32)


# ------------------------------------------------------------------ batch_distance
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("dim", [1, 3, 4, 5, 7, 16, 33, 63, 64, 65, 128, 255, 256, 257, 768, 1000, 1536])
def test_batch_distance_bit_exact(gpu_required, metric, dim):
    rng = np.random.default_rng(dim * 7 + int(metric))
    rows = rand_rows(rng, 97, dim, metric)
    q = rand_rows(rng, 1, dim, metric)[0]
    eng = va.HipDistance(metric)
    got = eng.batch_distance(q, rows)
    exp = po.batch_distance(int(metric), q, rows, po.MODE_C)
    assert np.array_equal(bits(got), bits(exp)), (got[:4], exp[:4])
    if metric in (DM.Cosine, DM.Euclidean, DM.DotProduct):
        gpu = va.GpuAccelerator.new()
        assert gpu is not None
        fn = {DM.Cosine: gpu.batch_cosine_similarity, DM.Euclidean: gpu.batch_euclidean_distance,
              DM.DotProduct: gpu.batch_dot_product}[metric]
        raw = fn(rows.reshape(-1), q, dim)
        expr = po.batch_compute_distance(int(metric), q, rows, po.MODE_C)
        assert np.array_equal(bits(raw), bits(expr))


def test_batch_distance_vs_reference_order_tolerance(gpu_required):
    # north-star: f32 distances within 1e-5 relative of the reference CPU path (mode R)
    rng = np.random.default_rng(1)
    rows = rng.standard_normal((2000, 768)).astype(np.float32)
    q = rng.standard_normal(768).astype(np.float32)
    for metric in (DM.Cosine, DM.Euclidean, DM.DotProduct):
        got = va.HipDistance(metric).batch_distance(q, rows)
        ref = po.batch_distance(int(metric), q, rows, po.MODE_R)
        if metric == DM.Euclidean or metric == DM.Cosine:  # distances ~1 / ~39: pure relative bound
            assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-5
        else:  # dot of random vectors is near 0: relative to |q||v| like the reference's own tests
            scale = np.linalg.norm(q) * np.linalg.norm(rows, axis=1)
            assert np.max(np.abs(got - ref) / scale) < 1e-6


def test_batch_distance_edge_cases(gpu_required):
    gpu = va.GpuAccelerator.new()
    assert gpu.batch_cosine_similarity(np.empty(0, np.float32), np.ones(4, np.float32), 4).size == 0
    assert gpu.batch_cosine_similarity(np.ones(8, np.float32), np.ones(4, np.float32), 0).size == 0
    z = np.zeros((3, 8), np.float32)
    out = gpu.batch_cosine_similarity(z.reshape(-1), np.ones(8, np.float32), 8)
    assert np.array_equal(out, np.zeros(3, np.float32))  # zero norm -> 0.0 (simd_avx512.rs:347-349)
    nanrow = np.full((1, 8), np.nan, np.float32)
    h = va.HipDistance(DM.Hamming).batch_distance(np.ones(8, np.float32), nanrow)
    assert h[0] == 8.0  # NaN > 0.5 is false
    # subnormals and signed zeros survive (no flush)
    tiny = np.full((1, 8), 1e-30, np.float32)
    d = va.HipDistance(DM.DotProduct).batch_distance(np.full(8, 1e-10, np.float32), tiny)
    assert bits(d)[0] == bits(po.batch_distance(po.DOT, np.full(8, 1e-10, np.float32), tiny, po.MODE_C))[0]


# ------------------------------------------------------------------ brute-force search
def oracle_brute(metric, rows, ids, queries, k, live=None):
    sel = np.arange(rows.shape[0]) if live is None else np.nonzero(live)[0]
    r, s = po.scan_topk(int(metric), rows[sel], queries, min(k, len(sel)) if len(sel) else 1,
                        sweep_mode(metric, rows.shape[1], k))
    out = []
    for qi in range(queries.shape[0]):
        n = min(k, len(sel))
        out.append((ids[sel[r[qi, :n].astype(np.int64)]], s[qi, :n]))
    return out


@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("n,dim", [(10000, 768), (5000, 128), (3000, 100), (777, 3), (4096, 256), (1500, 1024)])
def test_brute_force_ids_ranks_scores_exact(gpu_required, metric, n, dim):
    rng = np.random.default_rng(n + dim)
    rows = rand_rows(rng, n, dim, metric)
    ids = (np.arange(n, dtype=np.uint64) * 3 + 11)
    ix = va.HnswIndex(dim, metric)
    assert ix.upload(ids, rows) == n
    # nq >= 12 on dims that are multiples of 256 takes the LDS-resident query tiles (16 / 32 per pass)
    for nq, k in [(1, 10), (3, 1), (8, 10), (17, 5), (2, 64), (1, 100), (5, 200), (45, 10), (33, 70), (13, 3)]:
        Q = rand_rows(rng, nq, dim, metric)
        gid, gsc, gcnt = ix.search_batch_brute_force(Q, k)
        exp = oracle_brute(metric, rows, ids, Q, k)
        for qi in range(nq):
            eid, esc = exp[qi]
            assert gcnt[qi] == len(eid)
            assert np.array_equal(gid[qi, :gcnt[qi]], eid), (metric, nq, k, qi)
            assert np.array_equal(bits(gsc[qi, :gcnt[qi]]), bits(esc))
    ix.close()


def test_query_tile_size_does_not_change_results(gpu_required):
    rng = np.random.default_rng(99)
    rows = rng.standard_normal((20000, 768)).astype(np.float32)
    Q = rng.standard_normal((70, 768)).astype(np.float32)
    ix = va.HnswIndex(768, DM.Cosine)
    ix.upload(np.arange(20000), rows)
    ref = None
    try:
        for tile in (1, 8, 16, 32, 48, 128):
            va.set_max_query_tile(tile)
            out = ix.search_batch_brute_force(Q, 10)
            if ref is None:
                ref = out
            else:
                assert np.array_equal(out[0], ref[0]) and np.array_equal(bits(out[1]), bits(ref[1]))
    finally:
        va.set_max_query_tile(128)
    eid, esc = po.scan_topk(po.COSINE, rows, Q, 10, sweep_mode(DM.Cosine, 768), nthreads=4)
    assert np.array_equal(ref[0], eid) and np.array_equal(bits(ref[1]), bits(esc))


@pytest.mark.parametrize("metric", [DM.Cosine, DM.DotProduct])
@pytest.mark.parametrize("n,dim", [(10007, 768), (4000, 100), (3000, 17), (50, 64), (129, 128), (2500, 1001)])
def test_gemm_sweep_large_batches_bit_exact(gpu_required, metric, n, dim):
    # batches of >= 64 queries take the GEMM-structured matrix-core kernel (sweep_gemm.hip): one launch for the
    # whole batch, 128-row x <=128-query block tiles.  Same mode-M chain as the streaming kernel => same bits.
    rng = np.random.default_rng(n * 31 + dim)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64) * 7 + 3
    ix = va.HnswIndex(dim, metric)
    assert ix.upload(ids, rows) == n
    assert ix.sweep_arith_mode(10) == "M"
    for nq, k in [(64, 10), (65, 1), (100, 10), (128, 10), (129, 7), (192, 10), (200, 64), (300, 10)]:
        Q = rng.standard_normal((nq, dim)).astype(np.float32)
        gid, gsc, gcnt = ix.search_batch_brute_force(Q, k)
        eid, esc = po.scan_topk(int(metric), rows, Q, min(k, n), po.MODE_M, nthreads=8)
        kk = min(k, n)
        assert np.all(gcnt == kk)
        assert np.array_equal(gid[:, :kk], ids[eid.astype(np.int64)]), (nq, k)
        assert np.array_equal(bits(gsc[:, :kk]), bits(esc)), (nq, k)
    # soft-deleted rows are filtered inside the kernel's offer path
    dead = rng.choice(n, max(1, n // 10), replace=False)
    for d in dead:
        assert ix.remove(int(ids[d]))
    live = np.ones(n, bool)
    live[dead] = False
    Q = rng.standard_normal((96, dim)).astype(np.float32)
    gid, gsc, gcnt = ix.search_batch_brute_force(Q, 10)
    exp = oracle_brute(metric, rows, ids, Q, 10, live)
    for qi in range(96):
        eid, esc = exp[qi]
        assert gcnt[qi] == len(eid)
        assert np.array_equal(gid[qi, :gcnt[qi]], eid) and np.array_equal(bits(gsc[qi, :gcnt[qi]]), bits(esc))
    ix.close()


@pytest.mark.parametrize("metric", [DM.Cosine, DM.DotProduct])
@pytest.mark.parametrize("n,dim", [(10007, 768), (700, 128), (300, 256)])
def test_gemm_sweep_many_query_tiles_bit_exact(gpu_required, metric, n, dim):
    # 2 .. 8 query tiles per launch (row groups = resident block slots / query tiles, whole XCD rounds), ragged last
    # tiles, an exact tie inside a row tile.
    rng = np.random.default_rng(n * 13 + dim)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[n // 2] = rows[n // 2 - 1]  # an exact tie inside a tile
    ix = va.HnswIndex(dim, metric)
    ix.upload(np.arange(n), rows)
    for nq, k in [(230, 10), (480, 16), (700, 3), (1024, 10)]:
        Q = rng.standard_normal((nq, dim)).astype(np.float32)
        Q[5] = rows[n // 2]
        gid, gsc, gcnt = ix.search_batch_brute_force(Q, k)
        eid, esc = po.scan_topk(int(metric), rows, Q, k, po.MODE_M, nthreads=8)
        assert np.all(gcnt == k)
        assert np.array_equal(gid, eid), (nq, k)
        assert np.array_equal(bits(gsc), bits(esc)), (nq, k)
    assert ix.remove(int(gid[0, 0]))
    g2, _, _ = ix.search_batch_brute_force(Q, 10)
    assert int(gid[0, 0]) not in g2[0].tolist()
    ix.close()


@pytest.mark.parametrize("n,dim", [(10007, 768), (4000, 100), (3000, 17), (50, 64), (2500, 1001)])
def test_euclidean_large_batches_bit_exact(gpu_required, n, dim):
    # >= 64 Euclidean queries: approximate selection of k + 16 candidates on the matrix cores (|v|^2 + |q|^2 - 2 q.v),
    # canonical (q - v)^2 re-scoring, per-query proof of exactness, exact vector-ALU sweep for the unproven ones.
    # Whatever the route, ids and score bits equal the oracle's mode C — the same bits a single query returns.
    rng = np.random.default_rng(n * 17 + dim)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64) * 7 + 3
    ix = va.HnswIndex(dim, DM.Euclidean)
    assert ix.upload(ids, rows) == n
    for nq, k in [(64, 10), (129, 1), (300, 32), (200, 40)]:        # k = 40: beyond the candidate slack, VALU route
        Q = rng.standard_normal((nq, dim)).astype(np.float32)
        Q[3] = rows[n // 3]                                           # an exact hit (distance 0)
        gid, gsc, gcnt = ix.search_batch_brute_force(Q, k)
        kk = min(k, n)
        eid, esc = po.scan_topk(po.EUCLIDEAN, rows, Q, kk, po.MODE_C, nthreads=8)
        assert np.all(gcnt == kk)
        assert np.array_equal(gid[:, :kk], ids[eid.astype(np.int64)]), (nq, k)
        assert np.array_equal(bits(gsc[:, :kk]), bits(esc)), (nq, k)
        one = ix.search_batch_brute_force(Q[:2], k)                   # small batch: same bits
        assert np.array_equal(one[0][:, :kk], gid[:2, :kk]) and np.array_equal(bits(one[1][:, :kk]), bits(gsc[:2, :kk]))
    dead = rng.choice(n, max(1, n // 10), replace=False)
    for d in dead:
        assert ix.remove(int(ids[d]))
    live = np.ones(n, bool)
    live[dead] = False
    Q = rng.standard_normal((96, dim)).astype(np.float32)
    gid, gsc, gcnt = ix.search_batch_brute_force(Q, 10)
    exp = oracle_brute(DM.Euclidean, rows, ids, Q, 10, live)
    for qi in range(96):
        assert np.array_equal(gid[qi, :gcnt[qi]], exp[qi][0]) and np.array_equal(bits(gsc[qi, :gcnt[qi]]), bits(exp[qi][1]))
    ix.close()


def test_euclidean_large_batches_near_duplicates_and_specials(gpu_required):
    # near-duplicate clouds: hundreds of rows within 1e-4 of the query, far more than the candidate slack — the
    # approximate values cannot separate them (cancellation), the per-query verdict must send those queries to the exact
    # sweep; NaN / inf rows; duplicates (tie by row)
    rng = np.random.default_rng(41)
    n, dim = 6000, 256
    rows = rng.standard_normal((n, dim)).astype(np.float32) * 3
    centre = rng.standard_normal(dim).astype(np.float32) * 3
    rows[1000:1400] = centre + rng.standard_normal((400, dim)).astype(np.float32) * 1e-4
    rows[2000] = rows[1999]
    rows[17, 3] = np.nan
    rows[29, 0] = np.inf
    Q = rng.standard_normal((80, dim)).astype(np.float32) * 3
    Q[:40] = centre + rng.standard_normal((40, dim)).astype(np.float32) * 1e-4
    Q[50] = rows[1999]
    ix = va.HnswIndex(dim, DM.Euclidean)
    ix.upload(np.arange(n), rows)
    gid, gsc, gcnt = ix.search_batch_brute_force(Q, 10)
    eid, esc = po.scan_topk(po.EUCLIDEAN, rows, Q, 10, po.MODE_C, nthreads=8)
    assert np.array_equal(gid, eid)
    assert np.array_equal(bits(gsc), bits(esc))
    ix.close()


@pytest.mark.parametrize("dim", [256, 512, 1024])
def test_lds_query_tile_kernel_many_row_groups_per_wave(gpu_required, dim):
    # regression (found by tools/fuzz_sweep.py --euclid): sweep_topk_f32_qlds prefetches row chunks two steps ahead; with
    # ONE 256-float chunk per row (dim 256) "two steps ahead" is the group after the next, and it used to read the
    # neighbouring row instead — wrong results once a wave owned three or more row groups (tens of thousands of rows;
    # the other tests of this kernel are smaller).  16 and 32 queries per pass, Euclidean and the vector-ALU engine of
    # cosine / dot.
    rng = np.random.default_rng(dim)
    n = 70000 if dim == 256 else 40000
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    for metric in (DM.Euclidean, DM.Cosine, DM.DotProduct):
        ix = va.HnswIndex(dim, metric)
        ix.upload(np.arange(n), rows)
        try:
            if metric != DM.Euclidean:
                va.set_sweep_engine(0)
            for nq, k in [(16, 10), (40, 10), (32, 48)]:
                Q = rng.standard_normal((nq, dim)).astype(np.float32)
                gid, gsc, gcnt = ix.search_batch_brute_force(Q, k)
                eid, esc = po.scan_topk(int(metric), rows, Q, k, po.MODE_C, nthreads=8)
                assert np.array_equal(gid, eid), (metric, nq, k)
                assert np.array_equal(bits(gsc), bits(esc)), (metric, nq, k)
        finally:
            va.set_sweep_engine(1)
        ix.close()


def test_gemm_sweep_more_queries_than_one_launch(gpu_required):
    # > 1024 queries: several GEMM launches (kGemmMaxQueries), the tail through whatever kernel its size selects
    rng = np.random.default_rng(31)
    n, dim = 3000, 64
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((1100, dim)).astype(np.float32)
    ix = va.HnswIndex(dim, DM.DotProduct)
    ix.upload(np.arange(n), rows)
    gid, gsc, gcnt = ix.search_batch_brute_force(Q, 10)
    eid, esc = po.scan_topk(po.DOT, rows, Q, 10, po.MODE_M, nthreads=8)
    assert np.all(gcnt == 10) and np.array_equal(gid, eid) and np.array_equal(bits(gsc), bits(esc))
    ix.close()


def test_gemm_sweep_special_values(gpu_required):
    # zero rows / zero queries (cosine 0.0), NaN and inf rows: same bits and order as the oracle's total order
    rng = np.random.default_rng(77)
    n, dim = 1000, 64
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[5] = 0.0
    rows[17, 3] = np.nan
    rows[29, 0] = np.inf
    rows[41] = rows[40]  # exact duplicate: tie broken by row index
    Q = rng.standard_normal((80, dim)).astype(np.float32)
    Q[3] = 0.0
    Q[9] = rows[40]
    for metric in (DM.Cosine, DM.DotProduct):
        ix = va.HnswIndex(dim, metric)
        ix.upload(np.arange(n), rows)
        gid, gsc, gcnt = ix.search_batch_brute_force(Q, 20)
        eid, esc = po.scan_topk(int(metric), rows, Q, 20, po.MODE_M, nthreads=4)
        assert np.array_equal(gid, eid), metric
        assert np.array_equal(bits(gsc), bits(esc)), metric
        ix.close()


@pytest.mark.parametrize("metric", [DM.Cosine, DM.DotProduct])
def test_both_sweep_engines_bit_exact_against_their_oracle_mode(gpu_required, metric):
    # engine 1: matrix cores, oracle mode M; engine 0: vector ALUs, oracle mode C.  Same ids (up to sub-ulp
    # near-ties), scores within a few ulps of each other, each bit-identical to its own restatement.
    rng = np.random.default_rng(123)
    pm = int(metric)
    for n, dim in [(6000, 768), (3000, 200), (1000, 17), (2048, 128), (500, 2500)]:
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        Q = rng.standard_normal((37, dim)).astype(np.float32)
        ix = va.HnswIndex(dim, metric)
        ix.upload(np.arange(n), rows)
        try:
            va.set_sweep_engine(1)
            assert ix.sweep_arith_mode(10) == ("M" if dim <= 2400 else "C")
            mode1 = po.MODE_M if ix.sweep_arith_mode(10) == "M" else po.MODE_C
            g1 = ix.search_batch_brute_force(Q, 10)
            va.set_sweep_engine(0)
            assert ix.sweep_arith_mode(10) == "C"
            g0 = ix.search_batch_brute_force(Q, 10)
        finally:
            va.set_sweep_engine(1)
        e1 = po.scan_topk(pm, rows, Q, 10, mode1, nthreads=4)
        e0 = po.scan_topk(pm, rows, Q, 10, po.MODE_C, nthreads=4)
        assert np.array_equal(g1[0], e1[0]) and np.array_equal(bits(g1[1]), bits(e1[1]))
        assert np.array_equal(g0[0], e0[0]) and np.array_equal(bits(g0[1]), bits(e0[1]))
        scale = np.maximum(np.abs(g0[1]), 1e-3) if metric == DM.Cosine else np.sqrt(dim) * 4
        assert np.all(np.abs(g1[1] - g0[1]) <= 2e-6 * scale)
        ix.close()


def test_brute_force_single_query_api_and_order(gpu_required):
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((2000, 64)).astype(np.float32)
    for metric in (DM.Cosine, DM.Euclidean, DM.DotProduct):
        ix = va.HnswIndex(64, metric)
        ix.upload(np.arange(2000), rows)
        q = rng.standard_normal(64).astype(np.float32)
        res = ix.search_brute_force(q, 25)
        assert len(res) == 25
        sc = [s for _, s in res]
        if metric.higher_is_better():
            assert all(sc[i] >= sc[i + 1] for i in range(24))  # descending (distance.rs:96-98)
        else:
            assert all(sc[i] <= sc[i + 1] for i in range(24))
        # Perfect quality == brute force (search.rs:68-70)
        assert ix.search_with_quality(q, 25, va.SearchQuality.Perfect) == res
        assert ix.search_brute_force_buffered(q, 25) == res
        assert ix.search_brute_force_gpu(q, 25) == res


def test_reference_gpu_template_stricter(gpu_required):
    # hnsw/index_tests.rs:1551-1588: 100x128 sin vectors, query cos(0.02 j), k=10 brute force.
    # The reference accepts >= 8/10 id overlap; we require identical ids AND ranks vs the oracle,
    # in both orders: canonical (bit-exact) and reference wide16 (tie-aware == exact here).
    j = np.arange(128)
    rows = np.array([np.sin(((i + j).astype(np.float32)) * np.float32(0.01)) for i in range(100)], dtype=np.float32)
    q = np.cos(j.astype(np.float32) * np.float32(0.02)).astype(np.float32)
    ix = va.HnswIndex(128, DM.Cosine)
    for i in range(100):
        ix.upload(np.array([i], dtype=np.uint64), rows[i:i + 1])
    res = ix.search_brute_force(q, 10)
    c_ids, c_sc = po.scan_topk(po.COSINE, rows, q, 10, sweep_mode(DM.Cosine, 128))
    r_ids, r_sc = po.scan_topk(po.COSINE, rows, q, 10, po.MODE_R)
    assert [i for i, _ in res] == c_ids[0].tolist() == r_ids[0].tolist()
    assert np.array_equal(bits(np.float32([s for _, s in res])), bits(c_sc[0]))
    assert np.allclose([s for _, s in res], r_sc[0], rtol=1e-5, atol=0)


def test_soft_delete_and_small_counts(gpu_required):
    rng = np.random.default_rng(9)
    rows = rng.standard_normal((500, 32)).astype(np.float32)
    ids = np.arange(500, dtype=np.uint64) + 1000
    ix = va.HnswIndex(32, DM.Euclidean)
    ix.upload(ids, rows)
    q = rows[7] + 0.01
    assert ix.search_brute_force(q, 3)[0][0] == 1007
    assert ix.remove(1007) and not ix.remove(1007) and not ix.remove(5)
    assert ix.len() == 499 and ix.node_count() == 500
    live = np.ones(500, bool)
    live[7] = False
    got = ix.search_brute_force(q, 10)
    eid, esc = oracle_brute(DM.Euclidean, rows, ids, q[None, :], 10, live)[0]
    assert [i for i, _ in got] == eid.tolist()
    # k larger than the live count: every live row, best first
    small = va.HnswIndex(32, DM.Cosine)
    small.upload(np.arange(5), rows[:5])
    assert len(small.search_brute_force(q, 10)) == 5
    assert small.search_brute_force(q, 0) == []
    empty = va.HnswIndex(32, DM.Cosine)
    assert empty.search_brute_force(q, 10) == []
    assert empty.search(q, 10) == []


def test_duplicates_and_dimension_panics(gpu_required):
    ix = va.HnswIndex(3, DM.Cosine)
    assert ix.upload(np.array([1, 2, 1], dtype=np.uint64), np.eye(3, dtype=np.float32)) == 2
    assert ix.len() == 2
    assert ix.upload(np.array([2], dtype=np.uint64), np.ones((1, 3), np.float32)) == 0
    with pytest.raises(AssertionError, match="Vector dimension mismatch: expected 3, got 2"):
        ix.insert(9, [1.0, 2.0])
    with pytest.raises(AssertionError, match="Query dimension mismatch: expected 3, got 4"):
        ix.search([1, 2, 3, 4], 1)
    # the C ABI reports the same condition as a status, never by unwinding
    import ctypes as C
    from velesdb_amd import _ffi
    v = np.ones(2, np.float32)
    rc = _ffi.lib().vdb_hip_index_insert(ix._h, 9, v.ctypes.data_as(C.c_void_p), 2)
    assert rc == _ffi.VDB_ERR_DIM_MISMATCH and "expected 3, got 2" in _ffi.last_error()


def test_hamming_ties_canonical_order(gpu_required):
    # integer distances tie heavily at rank k: declared canonical order (distance, insertion idx)
    rng = np.random.default_rng(3)
    rows = (rng.random((4000, 48)) > 0.5).astype(np.float32)
    ix = va.HnswIndex(48, DM.Hamming)
    ix.upload(np.arange(4000), rows)
    q = (rng.random(48) > 0.5).astype(np.float32)
    got = ix.search_brute_force(q, 50)
    d = np.array([po.hamming(q, r) for r in rows])
    order = np.lexsort((np.arange(4000), d))[:50]
    assert [i for i, _ in got] == order.tolist()
    assert [s for _, s in got] == d[order].tolist()


@pytest.mark.parametrize("metric", [DM.Hamming, DM.Jaccard])
@pytest.mark.parametrize("n,dim", [(20000, 768), (5000, 48), (1300, 200), (700, 1)])
def test_packed_bit_sweep_batches_exact(gpu_required, metric, n, dim):
    # >= 3 queries: 8 or 32 queries per corpus pass (sweep_topk_bits_batch); integer work, canonical (score, row) order:
    # identical to the oracle and to the per-query kernel, ties at rank k included (dim 48 / 1: almost everything ties)
    rng = np.random.default_rng(n + dim + int(metric))
    rows = rand_rows(rng, n, dim, metric)
    rows[7] = 0.0                                           # empty bit set: Jaccard union can be 0 (score 1.0 vs an empty query)
    ids = np.arange(n, dtype=np.uint64) * 5 + 1
    ix = va.HnswIndex(dim, metric)
    assert ix.upload(ids, rows) == n
    for nq, k in [(30, 10), (64, 10), (100, 1), (300, 17), (32, 200), (9, 64)]:
        Q = rand_rows(rng, nq, dim, metric)
        Q[1] = 0.0
        gid, gsc, gcnt = ix.search_batch_brute_force(Q, k)
        exp = oracle_brute(metric, rows, ids, Q, k)
        for qi in range(nq):
            eid, esc = exp[qi]
            assert gcnt[qi] == len(eid)
            assert np.array_equal(gid[qi, :gcnt[qi]], eid), (metric, nq, k, qi)
            assert np.array_equal(bits(gsc[qi, :gcnt[qi]]), bits(esc))
        one = ix.search_batch_brute_force(Q[:1], k)         # the per-query kernel agrees
        assert np.array_equal(one[0][0, :one[2][0]], gid[0, :gcnt[0]])
    dead = rng.choice(n, n // 7, replace=False)
    for d in dead:
        assert ix.remove(int(ids[d]))
    live = np.ones(n, bool)
    live[dead] = False
    Q = rand_rows(rng, 40, dim, metric)
    gid, gsc, gcnt = ix.search_batch_brute_force(Q, 10)
    exp = oracle_brute(metric, rows, ids, Q, 10, live)
    for qi in range(40):
        assert np.array_equal(gid[qi, :gcnt[qi]], exp[qi][0]) and np.array_equal(bits(gsc[qi, :gcnt[qi]]), bits(exp[qi][1]))
    ix.close()


@pytest.mark.parametrize("metric", [DM.Cosine, DM.Euclidean])
def test_large_planted_neighbours_200k(gpu_required, metric):
    # size-independent property at a larger size: planted near-duplicates of the query must come
    # back first, in planted order; results sorted; and equal to the oracle on the same data.
    rng = np.random.default_rng(42)
    n, d = 200_000, 768
    rows = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    pos = rng.choice(n, 10, replace=False)
    for r, p in enumerate(pos):
        rows[p] = q + np.float32(0.01 * (r + 1)) * rng.standard_normal(d).astype(np.float32)
    ix = va.HnswIndex(d, metric)
    ix.upload(np.arange(n), rows)
    got = ix.search_brute_force(q, 10)
    assert [i for i, _ in got] == pos.tolist()
    eid, esc = po.scan_topk(int(metric), rows, q, 10, sweep_mode(metric, d), nthreads=8)
    assert [i for i, _ in got] == eid[0].tolist()
    assert np.array_equal(bits(np.float32([s for _, s in got])), bits(esc[0]))


@pytest.mark.parametrize("metric", [DM.Hamming, DM.Jaccard])
@pytest.mark.parametrize("n,dim", [(70_077, 768), (66_600, 100), (140_000, 48), (66_100, 1000)])
def test_bit_metric_batches_on_the_matrix_cores_exact(gpu_required, metric, n, dim):
    """Hamming / Jaccard batches of >= 32 queries over >= 65 536 rows run as a four-bit GEMM distance (bits_gemm.hip: the
    dot products on v_mfma_scale_f32_16x16x128_f8f6f4 over E2M1 images, the metric's bound and finish in the selection kernel's epilogue).  Integer
    work: ids, ranks (ties by row: dim 48 ties almost everywhere) and score bits equal the oracle's on sampled queries and the
    vector-ALU kernels' on every query; zero rows / zero queries (empty unions: Jaccard 1.0), duplicated rows, a ragged last
    tile, soft deletes, rows appended after the image was built."""
    rng = np.random.default_rng(n + dim + int(metric))
    rows = rand_rows(rng, n, dim, metric)
    rows[7] = 0.0
    rows[n - 1] = 0.0
    rows[1000:1040] = rows[999]                         # duplicates: exact ties across one tile
    rows[n - 300:n - 290] = rows[5]                      # ... and with rows of another launch / the seed prefix
    ids = np.arange(n, dtype=np.uint64) * 3 + 2
    ix = va.HnswIndex(dim, metric)
    assert ix.upload(ids[:n - 500], rows[:n - 500]) == n - 500
    nthreads = po.host_threads()

    def check(nq, k, live=None, n_now=n):
        Q = rand_rows(rng, nq, dim, metric)
        Q[1] = 0.0
        Q[2] = rows[999]
        Q[3] = rows[5]
        gid, gsc, gcnt = ix.search_batch_brute_force(Q, k)
        assert ix.last_kernels() & va.KERNEL_BITS_GEMM, "the matrix-core path did not serve the batch"
        ix.set_option(va.OPT_SWEEP_ENGINE, 0)            # the vector-ALU kernels on the same batch
        vid, vsc, vcnt = ix.search_batch_brute_force(Q, k)
        assert not (ix.last_kernels() & va.KERNEL_BITS_GEMM)
        ix.set_option(va.OPT_SWEEP_ENGINE, -1)
        assert np.array_equal(gcnt, vcnt) and np.array_equal(gid, vid) and np.array_equal(bits(gsc), bits(vsc)), (metric, n, dim, nq, k)
        sel = np.unique(np.concatenate([[0, 1, 2, 3, nq - 1], rng.choice(nq, 24, replace=False)]))
        lv = np.arange(n_now) if live is None else np.nonzero(live[:n_now])[0]
        r, s = po.scan_topk(int(metric), rows[lv], Q[sel], k, po.MODE_C, nthreads=nthreads)
        for j, qi in enumerate(sel):
            assert gcnt[qi] == k
            assert np.array_equal(gid[qi], ids[lv[r[j].astype(np.int64)]]), (metric, n, dim, nq, k, qi)
            assert np.array_equal(bits(gsc[qi]), bits(s[j]))

    check(256, 10, n_now=n - 500)
    check(300, 1, n_now=n - 500)                         # one full query tile + a partly filled one
    check(40, 10, n_now=n - 500)                         # the smallest batches that take the matrix cores (>= 32 queries)
    assert ix.upload(ids[n - 500:], rows[n - 500:]) == 500   # the image follows the inserts
    check(1024, 3)
    check(700, 10)
    dead = rng.choice(n, n // 9, replace=False)
    for d in dead[:2000]:
        assert ix.remove(int(ids[d]))
    live = np.ones(n, bool)
    live[dead[:2000]] = False
    check(480, 10, live)
    ix.close()
