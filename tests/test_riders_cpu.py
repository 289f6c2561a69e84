"""Host logic of the riders (no GPU): Collection::search_with_filter's index side (collection/search/vector.rs:164-235) over a stub
`search` — the over-fetch size max(4 k, k + 10), first-k-after-filter, and the metric's stable order."""
import velesdb_amd as va
from velesdb_amd import DistanceMetric as DM


class _Stub(va.HnswIndex):
    """an HnswIndex whose search() returns a canned list (no handle, no device)"""

    def __init__(self, metric, answers):
        self._metric, self._dimension, self._h = metric, 4, None
        self.answers, self.asked = answers, []

    def search(self, query, k):
        self.asked.append(k)
        return self.answers[:k]

    def close(self):
        pass

    __del__ = close


def test_search_filtered_over_fetch_and_cut():
    cand = [(i, 1.0 - 0.01 * i) for i in range(100)]          # cosine: best first
    ix = _Stub(DM.Cosine, cand)
    for k in (1, 3, 10, 25):
        ix.asked.clear()
        got = ix.search_filtered([0, 0, 0, 0], k, lambda i: i % 2 == 1)
        assert ix.asked == [max(4 * k, k + 10)]               # vector.rs:182
        want = [c for c in cand[:max(4 * k, k + 10)] if c[0] % 2 == 1][:k]
        assert got == want and len(got) <= k
    assert ix.search_filtered([0, 0, 0, 0], 5, lambda i: False) == []


def test_search_filtered_orders_by_the_metric_and_is_stable():
    # distance metric: ascending; equal scores keep the order the index returned them in (sort_by + partial_cmp is stable)
    cand = [(7, 0.5), (3, 0.25), (9, 0.5), (1, 0.75), (4, 0.25)]
    assert _Stub(DM.Euclidean, cand).search_filtered([0] * 4, 5, lambda i: True) == [(3, 0.25), (4, 0.25), (7, 0.5), (9, 0.5), (1, 0.75)]
    assert _Stub(DM.DotProduct, cand).search_filtered([0] * 4, 5, lambda i: True) == [(1, 0.75), (7, 0.5), (9, 0.5), (3, 0.25), (4, 0.25)]
    nan = float("nan")                                          # incomparable pairs compare Equal (unwrap_or(Equal)): they stay put
    got = _Stub(DM.Cosine, [(1, 0.5), (2, nan), (3, 0.9)]).search_filtered([0] * 4, 3, lambda i: True)
    assert [i for i, _ in got][0] in (1, 3) and len(got) == 3
