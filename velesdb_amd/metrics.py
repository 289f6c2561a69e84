"""Retrieval-quality metrics of the reference (crates/velesdb-core/src/metrics.rs), mirrored for the measurement side: recall@k is
the definition behind BASELINE's "QPS @ recall@10" (SURVEY §8d).  Pure host arithmetic — bench.py computes the same recall expression
inline (|truth ∩ result| / k over duplicate-free result lists); the tests use these functions."""
from typing import Hashable, Sequence


def recall_at_k(ground_truth: Sequence[Hashable], results: Sequence[Hashable]) -> float:
    """metrics.rs:46-57: results found in the truth set / |ground truth|; 0.0 for an empty ground truth (every result counts, a
    duplicate in `results` counts twice, as in the reference)."""
    if len(ground_truth) == 0:
        return 0.0
    truth = set(ground_truth)
    return sum(1 for r in results if r in truth) / len(ground_truth)


def precision_at_k(ground_truth: Sequence[Hashable], results: Sequence[Hashable]) -> float:
    """metrics.rs:81-93: relevant results / |results|; 0.0 for no results."""
    if len(results) == 0:
        return 0.0
    truth = set(ground_truth)
    return sum(1 for r in results if r in truth) / len(results)


def mrr(ground_truth: Sequence[Hashable], results: Sequence[Hashable]) -> float:
    """metrics.rs:113-124: reciprocal rank of the first relevant result, 0.0 if there is none."""
    truth = set(ground_truth)
    for rank, r in enumerate(results):
        if r in truth:
            return 1.0 / (rank + 1)
    return 0.0


def compute_recall(retrieved: Sequence[Hashable], ground_truth: Sequence[Hashable], k: int) -> float:
    """crates/velesdb-core/tests/recall_validation.rs:26-39: |top-k(retrieved) ∩ top-k(ground truth)| / k with
    k = min(k, len(retrieved), len(ground_truth)); 0.0 when that k is 0."""
    k = min(k, len(retrieved), len(ground_truth))
    if k == 0:
        return 0.0
    return len(set(retrieved[:k]) & set(ground_truth[:k])) / k


# the minimum acceptable recall values the reference declares (recall_validation.rs:221-230)
MIN_RECALL_AT_1, MIN_RECALL_AT_10, MIN_RECALL_AT_100 = 0.99, 0.95, 0.90
