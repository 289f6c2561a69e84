"""HnswIndex::save / ::load directory format (hnsw/index/constructors.rs:190-287): the bincode 1.3.3 files beside the
graph/vector files.  CPU-only: the oracle's writer/reader against byte-level known answers derived from bincode's
default encoding (fixed-width little-endian integers, u64 lengths, usize = u64, bool = 1 byte)."""
import os

import pytest

from oracle import pyoracle as po


def test_meta_bytes_known_answer(tmp_path):
    po.write_index_meta(str(tmp_path), 768, po.COSINE, True)
    raw = open(tmp_path / "native_meta.bin", "rb").read()
    assert raw == bytes.fromhex("0003000000000000" "00" "01")  # (768usize, 0u8, true)
    assert po.read_index_meta(str(tmp_path)) == (768, po.COSINE, True)
    po.write_index_meta(str(tmp_path), 3, po.JACCARD, False)
    assert open(tmp_path / "native_meta.bin", "rb").read() == bytes.fromhex("0300000000000000" "04" "00")
    # metric discriminants 0..4 only (constructors.rs:204-216)
    open(tmp_path / "native_meta.bin", "wb").write(bytes.fromhex("0300000000000000" "05" "00"))
    with pytest.raises(OSError, match="Unknown distance metric"):
        po.read_index_meta(str(tmp_path))


def test_mappings_bytes_known_answer(tmp_path):
    # two live entries: idx 0 -> id 7, idx 2 -> id 9 (idx 1 was removed), next_idx 3
    po.write_index_mappings(str(tmp_path), {0: 7, 2: 9}, next_idx=3)
    raw = open(tmp_path / "native_mappings.bin", "rb").read()
    u = lambda v: v.to_bytes(8, "little")
    assert raw == (u(2) + u(7) + u(0) + u(9) + u(2)      # id_to_idx: len, (id, idx)*
                   + u(2) + u(0) + u(7) + u(2) + u(9)    # idx_to_id: len, (idx, id)*
                   + u(3))                               # next_idx
    assert po.read_index_mappings(str(tmp_path)) == ({7: 0, 9: 2}, {0: 7, 2: 9}, 3)
    po.write_index_mappings(str(tmp_path), {})
    assert open(tmp_path / "native_mappings.bin", "rb").read() == u(0) + u(0) + u(0)
    assert po.read_index_mappings(str(tmp_path)) == ({}, {}, 0)


def test_reader_accepts_any_entry_order(tmp_path):
    # hash-iteration order of the writer is arbitrary: the reader must not depend on it
    import random
    m = {i: 1000 + 7 * i for i in range(50) if i % 5}
    items = list(m.items())
    random.Random(1).shuffle(items)
    po.write_index_mappings(str(tmp_path), dict(items), next_idx=50)
    a, b, n = po.read_index_mappings(str(tmp_path))
    assert b == m and a == {v: k for k, v in m.items()} and n == 50
