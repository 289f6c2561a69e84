"""Pins the oracle's (and the Python mirror's) `DistanceMetric` — core/distance.rs: calculate :52-73, higher_is_better :76-82,
sort_results :95-103 — against the reference's own tests for it (distance_tests.rs, transcribed as data with file:line).  sort_results
decides the order of every result list the exact search modes return (search.rs:176-219) and of the merge across shards, so its
direction, its stability and its total order over NaN / signed zeros are stated here as exact expectations.  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po
from velesdb_amd.params import DistanceMetric as DM

F32_MODES = [po.MODE_R, po.MODE_C, po.MODE_SCALAR]

# (fn, a, b, expected, tolerance, distance_tests.rs lines)
CALCULATE = [
    ("cosine", [1.0, 0.0, 0.0], [1.0, 0.0, 0.0], 1.0, 1e-6, "6-11"),
    ("cosine", [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], 0.0, 1e-6, "12-15"),
    ("euclidean", [0.0, 0.0, 0.0], [3.0, 4.0, 0.0], 5.0, 1e-6, "18-23"),
    ("dot", [1.0, 2.0, 3.0], [4.0, 5.0, 6.0], 32.0, 1e-6, "26-31"),
    ("hamming", [1.0, 0.0, 1.0, 0.0], [1.0, 0.0, 1.0, 0.0], 0.0, 0.0, "79-88"),
    ("hamming", [1.0, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0], 4.0, 0.0, "91-100"),
    ("hamming", [1.0, 0.0, 1.0, 0.0], [1.0, 1.0, 0.0, 0.0], 2.0, 0.0, "103-109"),
    ("jaccard", [1.0, 0.0, 1.0, 1.0], [1.0, 0.0, 1.0, 1.0], 1.0, 1e-6, "122-131"),
    ("jaccard", [1.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 1.0], 0.0, 1e-6, "134-143"),
    ("jaccard", [1.0, 1.0, 1.0, 0.0], [1.0, 1.0, 0.0, 1.0], 0.5, 1e-6, "146-155"),
    ("jaccard", [0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0], 1.0, 1e-6, "158-167"),   # both sets empty: defined as 1.0
]


@pytest.mark.parametrize("fn,a,b,exp,tol,src", CALCULATE, ids=[f"{c[0]}@distance_tests.rs:{c[5]}" for c in CALCULATE])
def test_calculate(fn, a, b, exp, tol, src):
    if fn in ("cosine", "euclidean", "dot"):
        for mode in F32_MODES:
            assert abs(getattr(po, fn)(a, b, mode) - exp) <= tol, mode
    else:
        assert abs(float(getattr(po, fn)(a, b)) - exp) <= tol


def test_higher_is_better():
    """distance_tests.rs:34-43,112-115,170-173"""
    want = {po.COSINE: True, po.DOT: True, po.JACCARD: True, po.EUCLIDEAN: False, po.HAMMING: False}
    for m, hib in want.items():
        assert po.higher_is_better(m) is hib
    # the Python mirror of the enum (what tests and bench drive) says the same, and numbers the metrics like the reference's files do
    assert {int(m): m.higher_is_better() for m in DM} == {int(DM.Cosine): True, int(DM.Euclidean): False, int(DM.DotProduct): True,
                                                          int(DM.Hamming): False, int(DM.Jaccard): True}
    assert (po.COSINE, po.EUCLIDEAN, po.DOT, po.HAMMING, po.JACCARD) == tuple(int(m) for m in
                                                                              (DM.Cosine, DM.Euclidean, DM.DotProduct, DM.Hamming, DM.Jaccard))


# (metric, input, expected id order, distance_tests.rs lines)
SORTS = [
    (po.COSINE, [(1, 0.7), (2, 0.9), (3, 0.8)], [2, 3, 1], "180-186"),
    (po.EUCLIDEAN, [(1, 5.0), (2, 2.0), (3, 3.0)], [2, 3, 1], "189-195"),
    (po.DOT, [(1, 10.0), (2, 30.0), (3, 20.0)], [2, 3, 1], "198-202"),
    (po.HAMMING, [(1, 4.0), (2, 1.0), (3, 2.0)], [2, 3, 1], "205-209"),
    (po.JACCARD, [(1, 0.3), (2, 0.9), (3, 0.5)], [2, 3, 1], "212-216"),
]


@pytest.mark.parametrize("metric,inp,order,src", SORTS, ids=[f"sort@distance_tests.rs:{c[3]}" for c in SORTS])
def test_sort_results(metric, inp, order, src):
    assert [i for i, _ in po.sort_results(metric, inp)] == order


def test_sort_results_nan_and_empty():
    """distance_tests.rs:219-230: NaN must not panic (`total_cmp` is a total order: the reference's result IS defined, and stated
    here), an empty list stays empty."""
    out = po.sort_results(po.COSINE, [(1, float("nan")), (2, 0.5), (3, 0.8)])
    assert [i for i, _ in out] == [1, 3, 2]           # descending total order: +NaN is above every number
    out = po.sort_results(po.EUCLIDEAN, [(1, float("nan")), (2, 0.5), (3, 0.8)])
    assert [i for i, _ in out] == [2, 3, 1]           # ascending: +NaN last
    neg_nan = np.uint32(0xFFC00000).view(np.float32)
    out = po.sort_results(po.EUCLIDEAN, [(1, 0.5), (2, float(neg_nan)), (3, float("-inf"))])
    assert [i for i, _ in out] == [2, 3, 1]           # -NaN is below -inf
    assert po.sort_results(po.COSINE, []) == []


def test_sort_results_is_stable_and_orders_signed_zeros():
    """`sort_by` is a stable sort (distance.rs:98,101): equal scores keep their input order — the property the brute-force path's row
    order and the shard merge (global row order among equals) rest on; -0.0 < +0.0 in the total order."""
    inp = [(10, 1.0), (11, 2.0), (12, 1.0), (13, 2.0), (14, 1.0)]
    assert [i for i, _ in po.sort_results(po.EUCLIDEAN, inp)] == [10, 12, 14, 11, 13]
    assert [i for i, _ in po.sort_results(po.COSINE, inp)] == [11, 13, 10, 12, 14]
    zeros = [(1, 0.0), (2, -0.0), (3, 0.0), (4, -0.0)]
    assert [i for i, _ in po.sort_results(po.EUCLIDEAN, zeros)] == [2, 4, 1, 3]
    assert [i for i, _ in po.sort_results(po.DOT, zeros)] == [1, 3, 2, 4]
    # against an independent statement of the same rule on random data with many ties
    rng = np.random.default_rng(4)
    for metric in (po.COSINE, po.EUCLIDEAN, po.HAMMING):
        sc = rng.integers(0, 5, 200).astype(np.float32)
        inp = [(int(i), float(s)) for i, s in enumerate(sc)]
        got = [i for i, _ in po.sort_results(metric, inp)]
        exp = sorted(range(200), key=(lambda i: (-sc[i], i)) if po.higher_is_better(metric) else (lambda i: (sc[i], i)))
        assert got == exp
