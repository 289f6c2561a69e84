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
