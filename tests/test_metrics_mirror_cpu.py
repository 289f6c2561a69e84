"""velesdb_amd/metrics.py (recall@k — the definition behind the headline metric —, precision@k, MRR) against the reference's own tests
(crates/velesdb-core/src/metrics_tests.rs, inputs transcribed with their lines).  CPU only; pure host arithmetic."""
import sys

import pytest

from velesdb_amd.metrics import mrr, precision_at_k, recall_at_k

EPS = sys.float_info.epsilon

RECALL = [
    ([1, 2, 3, 4, 5], [1, 2, 3, 4, 5], 1.0, "14-27"),
    ([1, 2, 3, 4, 5], [1, 3, 6, 2, 7], 0.6, "30-43"),
    ([1, 2, 3], [10, 20, 30], 0.0, "46-59"),
    ([], [1, 2, 3], 0.0, "62-75"),                       # empty ground truth: 0.0 by definition
    ([1, 2, 3], [], 0.0, "78-91"),
    (list(range(100)), list(range(100)), 1.0, "343-362"),   # an exact search
    (list(range(10)), [0, 1, 2, 3, 4, 5, 6, 7, 100, 101], 0.8, "365-378"),
    (list(range(100)), list(range(90)) + list(range(200, 210)), 0.9, "381-396"),
]
PRECISION = [
    ([1, 2, 3, 4, 5], [1, 2, 3, 4, 5], 1.0, "98-111"),
    ([1, 2, 3, 4, 5], [1, 3, 6, 2, 7], 0.6, "114-127"),
    ([1, 2, 3], [10, 20, 30], 0.0, "130-143"),
    ([1, 2, 3], [], 0.0, "146-159"),
    ([1, 2, 3], [1, 2, 3, 10, 20, 30, 40, 50, 60, 70], 0.3, "162-175"),
    (list(range(100)), list(range(100)), 1.0, "343-362"),
]
MRR = [
    ([1, 2, 3], [1, 10, 20, 30], 1.0, "182-195"),
    ([1, 2, 3], [10, 1, 20, 30], 0.5, "198-211"),
    ([1, 2, 3], [10, 20, 2, 30], 1.0 / 3.0, "214-228"),
    ([1, 2, 3], [10, 20, 30, 40], 0.0, "231-244"),
    ([1, 2, 3], [], 0.0, "247-260"),
]


@pytest.mark.parametrize("truth,results,exp,src", RECALL, ids=[f"recall@metrics_tests.rs:{c[3]}" for c in RECALL])
def test_recall_at_k(truth, results, exp, src):
    assert abs(recall_at_k(truth, results) - exp) < EPS


@pytest.mark.parametrize("truth,results,exp,src", PRECISION, ids=[f"precision@metrics_tests.rs:{c[3]}" for c in PRECISION])
def test_precision_at_k(truth, results, exp, src):
    assert abs(precision_at_k(truth, results) - exp) < EPS


@pytest.mark.parametrize("truth,results,exp,src", MRR, ids=[f"mrr@metrics_tests.rs:{c[3]}" for c in MRR])
def test_mrr(truth, results, exp, src):
    assert abs(mrr(truth, results) - exp) < EPS


def test_bench_expression_is_recall_at_k():
    """bench.py's inline `len(set(result) & set(truth)) / K` equals recall_at_k whenever a result list holds no id twice (every search
    path returns distinct ids) and the truth has K entries"""
    import numpy as np
    rng = np.random.default_rng(0)
    for _ in range(50):
        k = int(rng.integers(1, 20))
        truth = rng.choice(1000, k, replace=False).tolist()
        results = rng.choice(1000, k, replace=False).tolist()
        assert len(set(results) & set(truth)) / k == recall_at_k(truth, results)


# ---------------------------------------------------------------- crates/velesdb-core/tests/recall_validation.rs
def test_compute_recall_and_thresholds():
    from velesdb_amd.metrics import MIN_RECALL_AT_1, MIN_RECALL_AT_10, MIN_RECALL_AT_100, compute_recall
    assert abs(compute_recall([1, 2, 3, 4, 5], [1, 2, 3, 4, 5], 5) - 1.0) < EPS      # :87-96
    assert abs(compute_recall([1, 2, 3, 4, 5], [1, 2, 6, 7, 8], 5) - 0.4) < EPS      # :99-108
    assert abs(compute_recall([1, 2, 3, 4, 5], [6, 7, 8, 9, 10], 5)) < EPS           # :111-120
    assert compute_recall([], [1, 2], 5) == 0.0 and compute_recall([1, 2, 3], [1, 9], 10) == 0.5   # k = min(k, both lengths)
    assert MIN_RECALL_AT_1 > MIN_RECALL_AT_10 > MIN_RECALL_AT_100                    # :221-230


def test_recall_validation_fixtures_on_the_oracle_graph():
    """recall_validation.rs: the generator ((31 i + 17 d) mod 1000) / 1000 (:48-56), the brute-force ground truth by cosine DISTANCE
    (:59-83), its literal case (:123-137), the 100 x 32 and 1 000 x 128 fixtures (:140-182) — and, where the reference only simulates a
    retrieval (:190-218), the oracle's HNSW index on the 1 000 x 128 fixture held to the reference's declared minimum recalls."""
    import numpy as np
    from oracle import pyoracle as po
    from velesdb_amd.metrics import MIN_RECALL_AT_1, MIN_RECALL_AT_10, MIN_RECALL_AT_100, compute_recall

    def gen(count, dim):
        i = np.arange(count, dtype=np.int64)[:, None]
        d = np.arange(dim, dtype=np.int64)[None, :]
        return (((i * 31 + d * 17) % 1000).astype(np.float32) / np.float32(1000.0)).astype(np.float32)

    def ground_truth(vectors, q, k):
        dist = np.array([po.distance(po.COSINE, q, v, po.MODE_SCALAR) for v in vectors], dtype=np.float32)
        order = np.argsort(dist, kind="stable")[:k]
        return order.tolist(), dist[order]

    lit = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.9, 0.1, 0.0], [0.0, 0.0, 1.0]], dtype=np.float32)
    assert ground_truth(lit, lit[0], 2)[0] == [0, 2]                                  # :123-137
    v = gen(100, 32)
    ids, _ = ground_truth(v, v[50], 10)
    assert 50 in ids and compute_recall(ids, ids, 10) == 1.0                          # :140-160
    v = gen(1000, 128)
    ids, dist = ground_truth(v, v[500], 10)
    assert ids[0] == 500 and np.all(np.diff(dist) >= 0)                               # :163-182
    ix = po.HnswIndex(128, po.COSINE)
    for i, row in enumerate(v):
        ix.insert(i, row)
    tot = {1: 0.0, 10: 0.0, 100: 0.0}
    queries = [v[i * 10] for i in range(100)]                                         # (:192: every 100th of 10 000; here every 10th of 1 000)
    v64 = v.astype(np.float64)
    vn = v64 / np.linalg.norm(v64, axis=1, keepdims=True)
    for q in queries:
        # exact ranking in f64: the fixture holds nearly parallel rows (cosine distance ~1e-7 apart: rows 20 and 891), which an f32
        # ground truth orders by its own rounding — the reference's f32 `cosine_distance` puts row 891 at -1.2e-7 in FRONT of the
        # query itself at 0.0; its tests only ask that the query be IN the truth (:150) or first for query 500 (:181)
        q64 = q.astype(np.float64)
        gt = np.argsort(1.0 - vn @ (q64 / np.linalg.norm(q64)), kind="stable")[:100].tolist()
        for k in tot:
            got, _ = ix.search_with_quality(q, k, po.Q_BALANCED)
            tot[k] += compute_recall(got.tolist(), gt, k)
    assert tot[1] / 100 >= MIN_RECALL_AT_1 and tot[10] / 100 >= MIN_RECALL_AT_10 and tot[100] / 100 >= MIN_RECALL_AT_100, tot
