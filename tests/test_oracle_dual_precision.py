"""Pins the oracle's dual-precision restatement (per-dimension scalar quantiser, u8 codes, integer L2^2, int8 graph walk +
exact f32 re-rank) against the reference's OWN tests for it: index/hnsw/native/quantization_tests.rs and dual_precision_tests.rs,
transcribed as data (inputs, expected values / properties) with their file:line.  CPU only — tests/test_gpu_int8.py then compares the
HIP kernels with this oracle bit for bit."""
import numpy as np
import pytest

from oracle import pyoracle as po

F = np.float32


def rows(*vs):
    return np.array(vs, dtype=F)


# ---------------------------------------------------------------- ScalarQuantizer::train (quantization_tests.rs:12-50)
def test_train_computes_min_and_scale():
    sq = po.ScalarQuantizer(rows([0.0, 10.0, -5.0], [5.0, 20.0, 5.0], [2.5, 15.0, 0.0]))   # :13-29
    assert sq.dim == 3
    assert np.all(np.abs(sq.min_vals - rows([0.0, 10.0, -5.0])[0]) < 1e-6)
    assert np.all(np.abs(sq.scales - rows([255.0 / 5.0, 255.0 / 10.0, 255.0 / 10.0])[0]) < 1e-4)
    # the reference computes 255.0 / range in f32 and its inverse as 1.0 / scale (quantization.rs:218-231): bit-exact restatement
    assert np.array_equal(sq.scales, F(255.0) / rows([5.0, 10.0, 10.0])[0])
    assert np.array_equal(sq.inv_scales, F(1.0) / sq.scales)


def test_train_constant_dimension_gets_unit_scale():
    sq = po.ScalarQuantizer(rows([1.0, 5.0, 5.0], [2.0, 5.0, 5.0]))   # :32-41
    assert sq.scales[1] == F(1.0) and sq.scales[2] == F(1.0) and sq.scales[0] == F(255.0)
    # a range below 1e-10 counts as constant (quantization.rs:222)
    sq = po.ScalarQuantizer(rows([0.0, 3.0], [5e-11, 3.0]))
    assert sq.scales[0] == F(1.0)


# ---------------------------------------------------------------- quantize / dequantize (quantization_tests.rs:53-126)
def test_quantize_maps_the_training_range_onto_0_255():
    sq = po.ScalarQuantizer(rows([0.0, 100.0]))                        # :54-63: one vector, every dimension constant
    assert sq.quantize([0.0, 100.0])[0].tolist() == [0, 0]
    sq = po.ScalarQuantizer(rows([0.0, 0.0], [10.0, 100.0]))           # :66-85
    assert sq.quantize([0.0, 0.0])[0].tolist() == [0, 0]
    assert sq.quantize([10.0, 100.0])[0].tolist() == [255, 255]
    mid = sq.quantize([5.0, 50.0])[0].astype(int)
    assert np.all(np.abs(mid - 127) <= 1)
    assert mid.tolist() == [128, 128]                                  # f32::round: 127.5 rounds half AWAY from zero
    sq = po.ScalarQuantizer(rows([0.0], [10.0]))                       # :88-100: clamped outside the training range
    assert sq.quantize([-5.0])[0, 0] == 0 and sq.quantize([20.0])[0, 0] == 255
    assert sq.quantize([np.nan])[0, 0] == 0                            # `as u8` of NaN saturates to 0


def test_dequantize_recovers_within_one_percent_of_the_range():
    lo, hi = rows([0.0, -10.0, 100.0])[0], rows([10.0, 10.0, 200.0])[0]   # :103-126
    sq = po.ScalarQuantizer(np.stack([lo, hi]))
    orig = rows([5.0, 0.0, 150.0])
    rec = sq.dequantize(sq.quantize(orig))
    assert np.all(np.abs(orig - rec) / (hi - lo) < 0.01)


def test_768d_embedding_round_trip():
    i = np.arange(768, dtype=F)                                        # :245-268
    v1, v2 = np.sin(i * F(0.01), dtype=F), np.cos(i * F(0.01), dtype=F)
    sq = po.ScalarQuantizer(np.stack([v1, v2]))
    codes = sq.quantize(v1)
    assert sq.dim == 768 and codes.shape == (1, 768)
    rec = sq.dequantize(codes)[0]
    assert float(np.mean((v1 - rec) ** 2, dtype=F)) < 0.001


# ---------------------------------------------------------------- integer L2^2 between codes (quantization_tests.rs:128-147)
def test_distance_l2_quantized():
    sq = po.ScalarQuantizer(rows([0.0, 0.0], [10.0, 10.0]))
    v = sq.quantize([5.0, 5.0])[0]
    assert sq.distance_l2_quantized(v, v) == 0                         # :129-135
    a, b = sq.quantize([2.0, 3.0])[0], sq.quantize([7.0, 8.0])[0]      # :138-147
    assert sq.distance_l2_quantized(a, b) == sq.distance_l2_quantized(b, a)
    # the value itself: codes round(2 * 25.5) = 51, round(3 * 25.5) = 77 (76.5 away from zero), 179 (178.5), 204
    assert a.tolist() == [51, 77] and b.tolist() == [179, 204]
    assert sq.distance_l2_quantized(a, b) == (179 - 51) ** 2 + (204 - 77) ** 2
    # any length (the reference's 8-wide unrolled form and its remainder loop add the same integers: quantization.rs:42-91)
    rng = np.random.default_rng(3)
    for dim in (1, 7, 8, 9, 63, 64, 65, 768):
        x, y = rng.integers(0, 256, dim, dtype=np.uint8), rng.integers(0, 256, dim, dtype=np.uint8)
        q = po.ScalarQuantizer(np.zeros((1, dim), dtype=F))
        assert q.distance_l2_quantized(x, y) == int(((x.astype(np.int64) - y.astype(np.int64)) ** 2).sum())
    # the largest value a 768-dimensional pair can reach fits the reference's u32 accumulators
    q = po.ScalarQuantizer(np.zeros((1, 768), dtype=F))
    assert q.distance_l2_quantized(np.zeros(768, np.uint8), np.full(768, 255, np.uint8)) == 768 * 255 * 255


# ---------------------------------------------------------------- DualPrecisionHnsw (dual_precision_tests.rs)
def _graph(vectors, M, efc):
    g = po.NativeHnsw(vectors.shape[1], po.EUCLIDEAN, M, efc)
    for v in vectors:
        g.insert(v)
    return g


def _fixture(n, dim, step, fn):
    idx = (np.arange(n, dtype=np.int64)[:, None] * dim + np.arange(dim, dtype=np.int64)[None, :]).astype(F)   # `(i * dim + j) as f32`
    return fn(idx * F(step), dtype=F)


def _brute_l2(vectors, q, k):
    d = np.sqrt(((vectors.astype(np.float64) - q.astype(np.float64)) ** 2).sum(axis=1))
    return np.argsort(d, kind="stable")[:k]


def test_search_with_int8_traversal_200x64():
    """dual_precision_tests.rs:259-288: 200 x 64 sin(0.01 (64 i + j)), Euclidean, M 16, ef_construction 100, quantiser trained on what
    was inserted (force_train_quantizer: :177-181), query sin(0.01 j) = row 0, k 10, ef_search 50, oversampling 4: results non-empty,
    ascending.  (With DualPrecisionConfig::default() the reference itself answers this size from `inner.search` — min_index_size is
    10 000, dual_precision.rs:272-275 — so the fixture is run through BOTH paths here: the int8 traversal proper must satisfy the
    reference's assertions too.)"""
    vectors = _fixture(200, 64, 0.01, np.sin)
    g = _graph(vectors, 16, 100)
    sq = po.ScalarQuantizer(vectors)          # first min(1000, n) inserted vectors
    codes = sq.quantize(vectors)
    q = np.sin(np.arange(64, dtype=F) * F(0.01), dtype=F)
    assert np.array_equal(q, vectors[0])
    ids, ds, n_dist, n_expand = po.dual_search_int8(g, sq, codes, q, 10, 50, 4)
    assert len(ids) == 10 and np.all(np.diff(ds) >= 0)
    assert ids[0] == 0 and ds[0] == 0.0       # the query is row 0: the exact re-rank puts it first at distance 0
    assert n_dist > 0 and n_expand > 0
    fid, fds = g.search(q, 10, 50)            # what search_with_config returns below min_index_size
    assert len(fid) == 10 and np.all(np.diff(fds) >= 0) and fid[0] == 0
    # re-ranked distances are the ENGINE's exact f32 distances (dual_precision.rs:253-282): equal to the f32 search's for shared ids
    f32_of = dict(zip(fid.tolist(), fds.tolist()))
    for i, d in zip(ids.tolist(), ds.tolist()):
        if i in f32_of:
            assert np.float32(d) == np.float32(f32_of[i])


def test_int8_traversal_recall_vs_f32_500x128():
    """dual_precision_tests.rs:290-334 (and :223-256): 500 x 128 cos(0.001 (128 i + j)), M 32, ef_construction 200, query = row 0, k 10,
    ef_search 100: >= 90 % of the f32 results are among the int8-traversal results; node 0 is found."""
    vectors = _fixture(500, 128, 0.001, np.cos)
    g = _graph(vectors, 32, 200)
    sq = po.ScalarQuantizer(vectors)
    codes = sq.quantize(vectors)
    q = vectors[0].copy()
    fid, _ = g.search(q, 10, 100)
    ids, ds, _, _ = po.dual_search_int8(g, sq, codes, q, 10, 100, 4)
    assert len(fid) == 10 and len(ids) == 10
    recall = len(set(fid.tolist()) & set(ids.tolist())) / max(len(fid), 1)
    assert recall >= 0.90, recall
    assert 0 in ids.tolist() and 0 in fid.tolist()
    assert np.all(np.diff(ds) >= 0)
    # against the exact neighbours as well (the fixture's rows are 0.128 apart along one curve: the graph finds them all)
    exact = _brute_l2(vectors, q, 10)
    assert len(set(exact.tolist()) & set(ids.tolist())) >= 9


def test_dual_precision_recall_200x128():
    """dual_precision_tests.rs:122-158: 200 x 128 sin(0.01 (128 i + j)), M 32, ef_construction 200; query sin(0.01 j), k 10, ef 100:
    >= 5 results, ascending — `search` with a trained quantiser = f32 traversal over-fetching max(2 ef, 4 k) + exact re-rank
    (:209-243), i.e. the first k of a longer f32 search."""
    vectors = _fixture(200, 128, 0.01, np.sin)
    g = _graph(vectors, 32, 200)
    q = np.sin(np.arange(128, dtype=F) * F(0.01), dtype=F)
    rerank_k = max(100 * 2, 10 * 4)
    cid, cds = g.search(q, rerank_k, 100)
    order = np.argsort(cds, kind="stable")[:10]      # sort_by total_cmp of exact distances, truncate(k)
    ids, ds = cid[order], cds[order]
    assert len(ids) >= 5 and np.all(np.diff(ds) >= 0) and ids[0] == 0
    # int8 traversal over the same graph: same nearest neighbour, ascending exact distances
    sq = po.ScalarQuantizer(vectors)
    iid, ids_d, _, _ = po.dual_search_int8(g, sq, sq.quantize(vectors), q, 10, 100, 4)
    assert iid[0] == 0 and np.all(np.diff(ids_d) >= 0)


@pytest.mark.parametrize("oversampling", [1, 2, 4, 8])
def test_oversampling_only_widens_the_rerank_pool(oversampling):
    """DualPrecisionConfig::oversampling_ratio (dual_precision.rs:33-35; default 4, dual_precision_tests.rs:337-342): the k *
    oversampling best of the int8 walk are re-scored exactly — a larger pool can only improve the k-th exact distance."""
    vectors = _fixture(300, 64, 0.01, np.sin)
    g = _graph(vectors, 16, 100)
    sq = po.ScalarQuantizer(vectors)
    codes = sq.quantize(vectors)
    rng = np.random.default_rng(11)
    for _ in range(5):
        q = vectors[rng.integers(0, 300)] + rng.standard_normal(64).astype(F) * F(0.05)
        _, d1, _, _ = po.dual_search_int8(g, sq, codes, q, 5, 40, 1)
        _, dk, _, _ = po.dual_search_int8(g, sq, codes, q, 5, 40, oversampling)
        assert len(dk) == 5 and np.all(np.diff(dk) >= 0)
        assert dk[-1] <= d1[-1]
