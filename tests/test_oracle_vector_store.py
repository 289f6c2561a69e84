"""MmapStorage directory (core/storage/mmap.rs) on the oracle restatement: the behaviours the reference's own tests pin
(storage/tests.rs:8-175), plus the byte layout read off the code (vectors.idx = bincode map, vectors.wal records,
vectors.dat sizing).  CPU only."""
import os
import struct

import numpy as np
import pytest

from oracle import pyoracle as po


def test_new_creates_files(tmp_path):                      # tests.rs:8-15
    st = po.MmapVectorStore(str(tmp_path), 3)
    assert (tmp_path / "vectors.dat").exists() and (tmp_path / "vectors.wal").exists()
    assert os.path.getsize(tmp_path / "vectors.dat") == 16 * 1024 * 1024   # INITIAL_SIZE, mmap.rs:76,109-112
    assert len(st) == 0 and st.next_offset == 0
    st.close()


def test_store_retrieve_persistence_delete(tmp_path):      # tests.rs:18-80
    d = str(tmp_path)
    st = po.MmapVectorStore(d, 3)
    st.store(1, [1.0, 2.0, 3.0])
    assert st.retrieve(1).tolist() == [1.0, 2.0, 3.0] and len(st) == 1
    st.flush()
    st.close()
    st = po.MmapVectorStore(d, 3)                          # re-open: index from vectors.idx, data from vectors.dat
    assert st.retrieve(1).tolist() == [1.0, 2.0, 3.0] and len(st) == 1 and st.next_offset == 12
    st.delete(1)
    assert st.retrieve(1) is None and len(st) == 0
    assert st.retrieve(99) is None                         # tests.rs:167-174
    st.close()


def test_multiple_vectors_and_update_in_place(tmp_path):   # tests.rs:127-165
    st = po.MmapVectorStore(str(tmp_path), 4)
    for i in range(10):
        st.store(i, [float(i * 4 + j) for j in range(4)])
    for i in range(10):
        assert st.retrieve(i).tolist() == [float(i * 4 + j) for j in range(4)]
        assert st.index[i] == i * 16                       # offsets from the monotonic counter (:434-436)
    st.store(3, [4.0, 5.0, 6.0, 7.5])
    assert st.retrieve(3).tolist() == [4.0, 5.0, 6.0, 7.5] and st.index[3] == 48 and len(st) == 10
    with pytest.raises(OSError, match="Vector dimension mismatch: expected 4, got 3"):
        st.store(11, [1.0, 2.0, 3.0])                      # :403-412
    st.close()


def test_byte_layout(tmp_path):
    st = po.MmapVectorStore(str(tmp_path), 3)
    st.store(7, [1.0, 2.0, 3.0])
    st.store(9, [4.0, 5.0, 6.0])
    st.delete(7)
    st.flush()
    st.close()
    idx = open(tmp_path / "vectors.idx", "rb").read()
    assert idx == struct.pack("<QQQ", 1, 9, 12)            # bincode map: count, (id, offset)
    wal = open(tmp_path / "vectors.wal", "rb").read()
    rec1 = b"\x01" + struct.pack("<QI", 7, 12) + np.float32([1, 2, 3]).tobytes()
    rec2 = b"\x01" + struct.pack("<QI", 9, 12) + np.float32([4, 5, 6]).tobytes()
    assert wal == rec1 + rec2 + b"\x02" + struct.pack("<Q", 7)     # :414-426, :576-582
    dat = open(tmp_path / "vectors.dat", "rb").read(24)
    assert dat[:12] == bytes(12)                           # deleted slot: hole-punched / zeroed (:590-596)
    assert np.frombuffer(dat[12:], "<f4").tolist() == [4.0, 5.0, 6.0]
    ids, vecs = po.read_vector_store(str(tmp_path), 3)
    assert ids.tolist() == [9] and vecs.tolist() == [[4.0, 5.0, 6.0]]


def test_growth_rule(tmp_path):                            # ensure_capacity, :175-221
    st = po.MmapVectorStore(str(tmp_path), 1024)           # 4 KiB per vector: 4096 fill the initial 16 MiB
    st.next_offset = 16 * 1024 * 1024 - 4096
    st.store(1, np.zeros(1024, np.float32))                # fits exactly: no growth
    assert os.path.getsize(tmp_path / "vectors.dat") == 16 * 1024 * 1024
    st.store(2, np.ones(1024, np.float32))                 # required = 16 MiB + 4 KiB
    want = max(32 << 20, (16 << 20) + 4096 + (64 << 20), (16 << 20) + (64 << 20))
    assert os.path.getsize(tmp_path / "vectors.dat") == want
    assert st.retrieve(2)[0] == 1.0
    st.close()


def test_bad_index_file(tmp_path):
    (tmp_path / "vectors.idx").write_bytes(struct.pack("<QQ", 2, 1))   # claims 2 entries, holds half of one
    with pytest.raises(OSError):
        po.MmapVectorStore(str(tmp_path), 3)
