"""The reference's tests of `NativeHnswInner` (index/hnsw/native_inner_tests.rs — SURVEY §8a row a6) and of the second `VectorIndex`
implementor `NativeHnswIndex` (native_index_tests.rs), run against the oracle: graph search over sequentially inserted rows, the score
transform per metric, remove, the exact scan, persistence.  Inputs transcribed as data with their lines.  The GPU side of the same
surface: tests/test_gpu_riders.py (`NativeHnswIndex` mirror), tests/test_gpu_hnsw.py.  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po

F = np.float32


def ramp(n, dim, step):
    """`(i * dim + j) as f32 * step`"""
    return ((np.arange(n, dtype=np.int64)[:, None] * dim + np.arange(dim)[None, :]).astype(F) * F(step)).astype(F)


def graph(vectors, metric, M=16, efc=100):
    g = po.NativeHnsw(vectors.shape[1], metric, M, efc)
    for v in vectors:
        g.insert(v)
    return g


# ---------------------------------------------------------------- NativeHnswInner (native_inner_tests.rs)
@pytest.mark.parametrize("metric", [po.COSINE, po.EUCLIDEAN, po.DOT, po.HAMMING, po.JACCARD])
def test_inner_new_all_metrics(metric):
    g = po.NativeHnsw(32, metric, 16, 100)                       # :8-20
    assert len(g) == 0 and len(g.search(np.zeros(32, F), 5, 50)[0]) == 0


def test_inner_insert_and_search():
    vectors = ramp(20, 32, 0.01)                                 # :23-39
    g = graph(vectors, po.EUCLIDEAN)
    assert len(g) == 20
    ids, ds = g.search(np.arange(32, dtype=F) * F(0.01), 5, 50)
    assert 1 <= len(ids) <= 5 and ids[0] == 0
    assert ids.tolist() == [0, 1, 2, 3, 4] and ds[0] == 0.0     # rows lie on a line, 0.32 sqrt(32) apart


def test_inner_transform_score():
    eps = float(np.finfo(F).eps)
    assert abs(po.transform_score(po.COSINE, 0.3) - 0.7) < eps       # :42-45: similarity = 1 - distance
    assert abs(po.transform_score(po.EUCLIDEAN, 0.5) - 0.5) < eps    # :48-51
    assert abs(po.transform_score(po.DOT, 0.5) + 0.5) < eps          # :54-57: the engine's distance is -dot


def test_inner_persistence_roundtrip(tmp_path):
    vectors = ramp(30, 32, 0.01)                                 # :60-88 (the same shape as backend_adapter_tests.rs:138-172)
    g = graph(vectors, po.EUCLIDEAN)
    g.file_dump(str(tmp_path), "native_hnsw")
    loaded = po.NativeHnsw.file_load(str(tmp_path), "native_hnsw", po.EUCLIDEAN)
    assert len(loaded) == 30
    a, b = g.search(vectors[3], 5, 50), loaded.search(vectors[3], 5, 50)
    assert a[0].tolist() == b[0].tolist() and a[0][0] == 3


# ---------------------------------------------------------------- NativeHnswIndex (native_index_tests.rs)
def test_native_index_insert_search_and_batch_insert():
    vectors = ramp(50, 32, 0.01)                                 # :19-35
    g = graph(vectors, po.EUCLIDEAN)
    assert len(g) == 50
    ids, _ = g.search(np.arange(32, dtype=F) * F(0.01), 5, 128)  # VectorIndex::search = Balanced: ef max(128, 4 k)
    assert 1 <= len(ids) <= 5 and ids[0] == 0
    g = graph(np.stack([np.full(32, F(i) * F(0.01), F) for i in range(50)]), po.EUCLIDEAN)   # :38-46 insert_batch
    assert len(g) == 50


def test_native_index_delete_and_trait_calls():
    ix = po.HnswIndex(32, po.EUCLIDEAN)                          # :70-77
    ix.insert(1, np.full(32, 0.1, F))
    ix.insert(2, np.full(32, 0.2, F))
    assert ix.remove(1) is True and ix.remove(999) is False
    ix = po.HnswIndex(32, po.EUCLIDEAN)                          # :80-89 through the VectorIndex trait
    ix.insert(1, np.full(32, 0.1, F))
    assert len(ix) == 1
    ids, _ = ix.search(np.full(32, 0.1, F), 1)
    assert ids.tolist() == [1]


def test_native_index_brute_force():
    vectors = ramp(20, 32, 0.001)                                # :92-108
    ix = po.HnswIndex(32, po.EUCLIDEAN)
    for i, v in enumerate(vectors):
        ix.insert(i, v)
    ids, ds = ix.search_brute_force(np.arange(32, dtype=F) * F(0.001), 5)
    assert len(ids) == 5 and ids[0] == 0 and np.all(np.diff(ds) >= 0)
    assert ids.tolist() == [0, 1, 2, 3, 4]
    assert len(po.HnswIndex(32, po.EUCLIDEAN).search_brute_force(np.zeros(32, F), 5)[0]) == 0   # :111-116 empty
    ix = po.HnswIndex(32, po.EUCLIDEAN)                          # :119-126 k larger than the index
    ix.insert(1, np.full(32, 0.1, F))
    ix.insert(2, np.full(32, 0.2, F))
    ids, _ = ix.search_brute_force(np.zeros(32, F), 10)
    assert ids.tolist() == [1, 2]


def test_native_index_persistence(tmp_path):
    vectors = np.stack([np.full(32, F(i) * F(0.1), F) for i in range(30)])   # :49-67 (cosine over constant vectors: row 0 is the zero vector)
    g = graph(vectors, po.COSINE)
    g.file_dump(str(tmp_path), "native_hnsw")
    loaded = po.NativeHnsw.file_load(str(tmp_path), "native_hnsw", po.COSINE)
    assert len(loaded) == 30
    ids, ds = loaded.search(np.zeros(32, F), 5, 128)
    assert len(ids) >= 1 and np.all(ds == 1.0)                  # a zero query: every cosine distance is 1.0 (similarity defined as 0)


# ---------------------------------------------------------------- ShardedMappings as seen through the index (sharded_mappings_tests.rs)
def test_id_mapping_semantics_through_the_index():
    """register = dense internal indices in arrival order (:19-33), a duplicate id is refused and changes nothing (:36-42), remove
    frees the id but its index is never handed out again (:61-69, :305-314: next_idx survives), contains / get_idx / get_id as the
    search results show them — the rules the GPU index's `ext_ids` / `alive` arrays and host id map follow (row a22)."""
    ix = po.HnswIndex(4, po.EUCLIDEAN)
    v = lambda x: np.full(4, x, F)
    assert ix.insert(42, v(1.0)) and ix.insert(100, v(2.0)) and ix.insert(999, v(3.0)) and len(ix) == 3
    g = ix.graph
    assert [g.vector(i)[0] for i in range(3)] == [1.0, 2.0, 3.0]          # node 0, 1, 2 in registration order
    assert ix.insert(42, v(9.0)) is False and len(ix) == 3 and len(g) == 3   # duplicate: skipped, nothing stored
    assert ix.search_brute_force(v(1.0), 3)[0].tolist() == [42, 100, 999]
    assert ix.remove(42) is True and len(ix) == 2 and ix.remove(42) is False   # :61-75
    assert ix.search_brute_force(v(1.0), 3)[0].tolist() == [100, 999]     # the removed id is in no result
    assert ix.insert(42, v(0.5)) is True and len(ix) == 3                 # the id is free again ...
    assert len(ix.graph) == 4 and ix.graph.vector(3)[0] == 0.5            # ... and gets a NEW internal index (3), never the old one
    assert ix.search_brute_force(v(0.5), 1)[0].tolist() == [42]


def test_mappings_file_keeps_next_idx(tmp_path):
    """as_parts / from_parts (:258-314) = what native_mappings.bin holds (constructors.rs:262-271): both maps and next_idx; a removed
    entry leaves a hole, and next_idx — not the number of entries — says where registration continues"""
    po.write_index_mappings(str(tmp_path), {0: 42, 1: 100, 2: 999})
    id_to_idx, idx_to_id, next_idx = po.read_index_mappings(str(tmp_path))
    assert id_to_idx == {42: 0, 100: 1, 999: 2} and idx_to_id == {0: 42, 1: 100, 2: 999} and next_idx == 3   # :270-301
    po.write_index_mappings(str(tmp_path), {0: 1, 2: 3}, next_idx=3)       # idx 1 was removed
    id_to_idx, idx_to_id, next_idx = po.read_index_mappings(str(tmp_path))
    assert len(id_to_idx) == 2 and next_idx == 3 and 1 not in idx_to_id
    po.write_index_mappings(str(tmp_path), {})                             # :261-267 empty
    assert po.read_index_mappings(str(tmp_path)) == ({}, {}, 0)
