"""A flushed MmapStorage directory as an upload source for the GPU index (core/storage/mmap.rs): the
C-ABI importer against the oracle's restatement of the store.  Rows arrive in ascending byte offset; exact search over
them is bit-identical to the oracle's scan over the same rows."""
import struct

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM = va.DistanceMetric


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def make_store(path, dim, n, seed):
    rng = np.random.default_rng(seed)
    st = po.MmapVectorStore(str(path), dim)
    ids = rng.choice(1 << 40, n, replace=False).astype(np.uint64)
    for i in ids:
        st.store(int(i), rng.standard_normal(dim).astype(np.float32))
    for i in ids[::7]:
        st.store(int(i), rng.standard_normal(dim).astype(np.float32))    # updates keep their slot
    for i in ids[3::11]:
        st.delete(int(i))
    st.flush()
    st.close()
    return ids


@pytest.mark.parametrize("metric", [DM.Cosine, DM.Euclidean, DM.DotProduct])
def test_upload_vector_store_matches_oracle(tmp_path, metric):
    dim, n = 24, 600
    make_store(tmp_path, dim, n, 5)
    sids, svecs = po.read_vector_store(str(tmp_path), dim)
    assert 0 < len(sids) < n
    ix = va.HnswIndex(dim, metric)
    assert ix.upload_vector_store(str(tmp_path)) == len(sids) == len(ix)
    Q = np.random.default_rng(6).standard_normal((9, dim)).astype(np.float32)
    gid, gsc, gcnt = ix.search_batch_brute_force(Q, 10)
    mode = po.MODE_M if ix.sweep_arith_mode(10) == "M" else po.MODE_C
    eid, esc = po.scan_topk(int(metric), svecs, Q, 10, mode, nthreads=2)
    assert np.all(gcnt == 10)
    assert np.array_equal(gid, sids[eid.astype(np.int64)])     # external ids of the store, rows in offset order
    assert np.array_equal(bits(gsc), bits(esc))
    assert ix.upload_vector_store(str(tmp_path)) == 0          # every id already present: skipped (trait_impl.rs:23-25)
    ix.build_graph()                                           # the uploaded rows can be linked like any others
    res = ix.search(Q[0], 5)
    assert len(res) == 5 and res[0][0] == int(gid[0, 0])
    ix.close()


def test_upload_vector_store_errors(tmp_path):
    ix = va.HnswIndex(24, DM.Cosine)
    with pytest.raises(va.VelesHipError):
        ix.upload_vector_store(str(tmp_path / "missing"))
    make_store(tmp_path, 24, 50, 8)
    wide = va.HnswIndex(48, DM.Cosine)                         # wrong dimension: the 96-byte slots overlap as 192-byte rows
    with pytest.raises(va.VelesHipError):
        wide.upload_vector_store(str(tmp_path))
    bad = tmp_path / "bad"
    bad.mkdir()
    (bad / "vectors.dat").write_bytes(bytes(96 * 2))
    (bad / "vectors.idx").write_bytes(struct.pack("<QQQQQ", 2, 1, 0, 2, 96 * 2))   # second offset past the end
    with pytest.raises(va.VelesHipError):
        ix.upload_vector_store(str(bad))
    (bad / "vectors.idx").write_bytes(struct.pack("<QQQ", 3, 1, 0))                # count does not match the file
    with pytest.raises(va.VelesHipError):
        ix.upload_vector_store(str(bad))
    assert len(ix) == 0
