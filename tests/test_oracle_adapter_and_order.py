"""The reference's own tests for `OrderedFloat` (native/ordered_float_tests.rs — the total order every candidate / result heap of the
graph path sorts by) and for the backend adapter (native/backend_adapter_tests.rs — parallel_insert,
search_neighbours, file_dump / file_load; row a24), run against the oracle.  Inputs transcribed as data, lines cited.  (The three
transform_score tests of that file are KATs in tests/golden/reference_kats.json.)  CPU only."""
import os

import numpy as np

from oracle import pyoracle as po

F = np.float32


# ---------------------------------------------------------------- OrderedFloat (ordered_float_tests.rs)
def test_ordered_float_total_order():
    assert po.total_cmp(1.0, 1.0) == 0                          # :7-11 eq, :37-41
    assert po.total_cmp(1.0, 2.0) != 0                          # :14-18 ne
    assert po.total_cmp(1.0, 2.0) == -1                         # :21-26 less
    assert po.total_cmp(3.0, 2.0) == 1                          # :29-34 greater
    assert po.total_cmp(-1.0, 1.0) == -1                        # :44-48
    assert po.total_cmp(0.0, -0.0) == 1 and po.total_cmp(-0.0, 0.0) == -1   # :51-61: -0.0 < +0.0 in the total order
    assert po.total_cmp(0.0, 0.0) == 0 and po.total_cmp(-0.0, -0.0) == 0
    # :79-93 sorting; :64-76 a max-heap pops 3, 2, 1 — through the restated BinaryHeap (`heap_order` = its backing array after the
    # pushes; the root is the maximum) and through the stable sort every result list goes through
    asc = [i for i, _ in po.sort_results(po.EUCLIDEAN, [(0, 3.0), (1, 1.0), (2, 2.0), (3, -1.0)])]
    assert asc == [3, 1, 2, 0]
    assert po.heap_order(np.array([3.0, 1.0, 2.0], dtype=F), np.array([0, 1, 2], dtype=np.uint64), False)[0] == 0   # max-heap root = 3.0
    assert po.heap_order(np.array([3.0, 1.0, 2.0], dtype=F), np.array([0, 1, 2], dtype=np.uint64), True)[0] == 1    # min-heap root = 1.0


# ---------------------------------------------------------------- backend adapter (backend_adapter_tests.rs)
def _graph(vectors, metric=po.EUCLIDEAN, M=16, efc=100):
    g = po.NativeHnsw(vectors.shape[1], metric, M, efc)
    for v in vectors:
        g.insert(v)
    return g


def test_parallel_insert_counts():
    """:35-62 — 10 constant vectors [i; 32] and 50 vectors [0.01 i; 32] (below the reference's 100-vector threshold parallel_insert is
    the sequential insert, backend_adapter.rs:110-123): every vector is in the index"""
    g = _graph(np.stack([np.full(32, float(i), F) for i in range(10)]))
    assert len(g) == 10
    g = _graph(np.stack([np.full(32, F(i) * F(0.01), F) for i in range(50)]))
    assert len(g) == 50
    for node in range(50):                                       # and every node is linked into layer 0
        assert len(g.neighbors(0, node)) > 0


def test_search_neighbours_format():
    """:67-84 — 50 constant vectors [0.1 i; 32], query zeros, k 5, ef 50: <= 5 results, node ids < 50, distances >= 0 (exactly: the
    five smallest i, at distances 0.1 i sqrt(32), ascending)"""
    g = _graph(np.stack([np.full(32, F(i) * F(0.1), F) for i in range(50)]))
    ids, ds = g.search(np.zeros(32, F), 5, 50)
    assert len(ids) <= 5 and np.all(ids < 50) and np.all(ds >= 0.0)
    assert ids.tolist() == [0, 1, 2, 3, 4] and np.all(np.diff(ds) > 0)
    assert np.allclose(ds, np.arange(5) * 0.1 * np.sqrt(32.0), rtol=1e-5)


def test_file_dump_creates_files_and_roundtrips(tmp_path):
    """:121-135 (20 constant vectors: both files exist), :138-172 (30 vectors 0.01 (32 i + j): the loaded index has 30 nodes and answers
    the query vectors[0] with the same list)"""
    g = _graph(np.stack([np.full(32, float(i), F) for i in range(20)]))
    g.file_dump(str(tmp_path), "test_index")
    assert os.path.exists(tmp_path / "test_index.vectors") and os.path.exists(tmp_path / "test_index.graph")
    vectors = ((np.arange(30, dtype=np.int64)[:, None] * 32 + np.arange(32)[None, :]).astype(F) * F(0.01)).astype(F)
    g = _graph(vectors)
    g.file_dump(str(tmp_path), "roundtrip")
    loaded = po.NativeHnsw.file_load(str(tmp_path), "roundtrip", po.EUCLIDEAN)
    assert len(loaded) == 30
    a_ids, a_ds = g.search(vectors[0], 5, 50)
    b_ids, b_ds = loaded.search(vectors[0], 5, 50)
    assert len(a_ids) == len(b_ids) == 5 and a_ids[0] == b_ids[0] == 0
    assert a_ids.tolist() == b_ids.tolist() and np.array_equal(a_ds.view(np.uint32), b_ds.view(np.uint32))
    for layer in range(g.num_layers):                           # the adjacency itself survives the round trip
        for node in range(30):
            assert g.neighbors(layer, node) == loaded.neighbors(layer, node)
