"""GPU parity tests of the dual-precision search: DualPrecisionHnsw::search_with_config(use_int8_traversal)
(native/dual_precision.rs:223-441) — scalar quantiser (native/quantization.rs:191-252), int8 graph walk with integer
L2^2 distances, exact f32 re-rank.  Integer arithmetic: ids, ranks, exact distances AND the kernel's counters must be
bit-identical to the oracle's restatement."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM = va.DistanceMetric
PO_METRIC = {DM.Cosine: po.COSINE, DM.Euclidean: po.EUCLIDEAN, DM.DotProduct: po.DOT}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", [DM.Cosine, DM.Euclidean, DM.DotProduct])
@pytest.mark.parametrize("n,dim,M,efc", [(3000, 96, 8, 60), (1500, 768, 16, 100), (1200, 37, 6, 40)])
def test_int8_traversal_bit_exact(tmp_path, metric, n, dim, M, efc):
    rng = np.random.default_rng(n + dim)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[:, 0] = 1.5  # a constant dimension: scale must fall back to 1.0 (quantization.rs:219-221)
    g = po.NativeHnsw(dim, PO_METRIC[metric], M, efc, po.MODE_C)
    for v in rows:
        g.insert(v)
    g.file_dump(str(tmp_path), "native_hnsw")
    ix = va.HnswIndex(dim, metric, va.HnswParams(M, efc, n))
    ix.load_reference_files(str(tmp_path), "native_hnsw")
    ix.train_quantizer()                       # first min(1000, n) rows, like DualPrecisionHnsw
    sq = po.ScalarQuantizer(rows[:1000])
    codes = sq.quantize(rows)
    qs = rng.standard_normal((20, dim)).astype(np.float32)
    for k, ef in [(10, 64), (1, 16), (25, 50), (10, 300)]:   # ef < k*4 -> ef = k*4 (dual_precision.rs:334); ef > 192: LDS list
        res = ix.search_batch_int8(qs, k, ef)
        nd_gpu, ne_gpu = ix.last_search_stats()
        nd = ne = 0
        for qi, q in enumerate(qs):
            oid, od, a, b = po.dual_search_int8(g, sq, codes, q, k, ef, 4, po.TIE_CANONICAL)
            nd += a
            ne += b
            osc = np.array([po.transform_score(PO_METRIC[metric], float(x)) for x in od], dtype=np.float32)
            assert [r[0] for r in res[qi]] == oid.tolist(), (metric, k, ef, qi)
            assert np.array_equal(bits([r[1] for r in res[qi]]), bits(osc))
        assert (nd_gpu, ne_gpu) == (nd, ne)


def test_int8_rows_added_after_training_and_recall(tmp_path):
    # rows inserted after the quantiser was trained are encoded with the same quantiser (dual_precision.rs:115-117);
    # recall of int8 + re-rank stays close to the f32 traversal's
    rng = np.random.default_rng(8)
    n, dim = 6000, 64
    proj = rng.standard_normal((12, dim)).astype(np.float32)
    rows = (rng.standard_normal((n, 12)).astype(np.float32) @ proj + 0.1 * rng.standard_normal((n, dim)).astype(np.float32))
    ix = va.HnswIndex(dim, DM.Euclidean, va.HnswParams(12, 100, n))
    ix.insert_batch_parallel([(i, rows[i]) for i in range(2000)], 64)
    ix.train_quantizer()
    ix.insert_batch_parallel([(i, rows[i]) for i in range(2000, n)], 256)
    qs = (rng.standard_normal((50, 12)).astype(np.float32) @ proj).astype(np.float32)
    gt, _, _ = ix.search_batch_brute_force(qs, 10)
    r8 = ix.search_batch_int8(qs, 10, 128)
    r32 = ix.search_batch_parallel(qs, 10, va.SearchQuality.Custom(128))
    rec8 = np.mean([len({x for x, _ in r} & set(gt[i].tolist())) / 10 for i, r in enumerate(r8)])
    rec32 = np.mean([len({x for x, _ in r} & set(gt[i].tolist())) / 10 for i, r in enumerate(r32)])
    assert rec8 >= rec32 - 0.05 and rec8 >= 0.9, (rec8, rec32)
    # re-ranked scores are exact f32 distances: identical to the f32 traversal's score for the same id
    d32 = {i: s for r in r32[:1] for i, s in r}
    for i, s in r8[0]:
        if i in d32:
            assert s == d32[i]


def test_int8_needs_training():
    ix = va.HnswIndex(16, DM.Cosine, va.HnswParams(8, 50, 100))
    for i in range(150):
        ix.insert(i, np.random.default_rng(i).standard_normal(16).astype(np.float32))
    with pytest.raises(va.VelesHipError):
        ix.search_batch_int8(np.zeros((1, 16), np.float32), 5, 64)


def test_search_with_config_applies_the_rule_natively_and_takes_the_ratio_per_call(tmp_path):
    """DualPrecisionHnsw::search_with_config (native/dual_precision.rs:259-278) through vdb_hip_index_search_with_config: the int8
    traversal only with a trained quantiser AND use_int8_traversal AND len >= min_index_size — decided inside the library from the
    handle's own state (is_quantizer_trained is read from the handle too) — with the call's OWN oversampling ratio (the handle
    option VDB_OPT_INT8_OVERSAMPLING is neither read nor written).  Every branch against the oracle: ids + score bits."""
    rng = np.random.default_rng(77)
    n, dim, M, efc = 2500, 48, 8, 60
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    g = po.NativeHnsw(dim, po.EUCLIDEAN, M, efc, po.MODE_C)
    for v in rows:
        g.insert(v)
    g.file_dump(str(tmp_path), "native_hnsw")
    ix = va.HnswIndex(dim, DM.Euclidean, va.HnswParams(M, efc, n))
    ix.load_reference_files(str(tmp_path), "native_hnsw")
    qs = rng.standard_normal((6, dim)).astype(np.float32)
    k, ef = 10, 64

    def f32_walk(q):
        oid, od = g.search(q, k, ef, po.TIE_CANONICAL)
        return oid.tolist(), np.array([po.transform_score(po.EUCLIDEAN, float(x)) for x in od], dtype=np.float32)

    def same(res, oid, osc):
        return [r[0] for r in res] == oid and np.array_equal(bits([r[1] for r in res]), bits(osc))

    assert not ix.is_quantizer_trained()
    cfg_small = va.DualPrecisionConfig(min_index_size=100)
    for q in qs:                                   # no quantiser: the plain f32 graph search whatever the config says
        assert same(ix.search_with_config(q, k, ef, cfg_small), *f32_walk(q))
    ix.train_quantizer()
    assert ix.is_quantizer_trained()
    sq = po.ScalarQuantizer(rows[:1000])
    codes = sq.quantize(rows)
    opt_before = ix.get_option(va.OPT_INT8_OVERSAMPLING)
    for ratio in (1, 4, 7):                        # the call's own ratio: k * ratio of the int8 walk's best are re-scored exactly
        cfg = va.DualPrecisionConfig(oversampling_ratio=ratio, min_index_size=100)
        for q in qs:
            oid, od, _, _ = po.dual_search_int8(g, sq, codes, q, k, ef, ratio, po.TIE_CANONICAL)
            osc = np.array([po.transform_score(po.EUCLIDEAN, float(x)) for x in od], dtype=np.float32)
            assert same(ix.search_with_config(q, k, ef, cfg), oid.tolist(), osc), ratio
    assert ix.get_option(va.OPT_INT8_OVERSAMPLING) == opt_before      # untouched
    for q in qs:                                   # default config: 2 500 < min_index_size 10 000 => f32 search
        assert same(ix.search_with_config(q, k, ef), *f32_walk(q))
        assert same(ix.search_with_config(q, k, ef, va.DualPrecisionConfig(use_int8_traversal=False, min_index_size=0)), *f32_walk(q))
    # the f32 branch is NativeHnsw::search with ef_search AS GIVEN (dual_precision.rs:269,274 -> graph.rs:251-270): with ef_search < k
    # at most ef_search results, ef_search = 0 acts as 1 — not HnswIndex's max(ef, k) / "0 = Balanced" rules
    for ef_small in (4, 1, 0, 9):
        for q in qs:
            oid, od = g.search(q, k, ef_small, po.TIE_CANONICAL)
            assert len(oid) == max(ef_small, 1), "the oracle's NativeHnsw::search returns what search_layer(ef_search) holds"
            osc = np.array([po.transform_score(po.EUCLIDEAN, float(x)) for x in od], dtype=np.float32)
            res = ix.search_with_config(q, k, ef_small)
            assert same(res, oid.tolist(), osc), ef_small
            mid, msc, mcnt = ix.search_multi_entry(q, k, ef_small, 1)   # what the Rust shim's DualPrecisionHnsw::search binds
            assert int(mcnt[0]) == len(oid) and mid[0, :len(oid)].tolist() == oid.tolist() and np.array_equal(bits(msc[0, :len(oid)]), bits(osc))
    ix.close()
