"""The host-side mirror of the reference's `HnswParams` / `SearchQuality` (velesdb_amd/params.py: what tests, bench and a Python
caller construct an index with) against the reference's own tests for them — index/hnsw/params_tests.rs, every expectation
transcribed with its lines.  The oracle's C++ `ef_search` (what the GPU results are checked against) is held to the same table.
CPU only; pure host logic."""
import pytest

from oracle import pyoracle as po
from velesdb_amd.params import DualPrecisionConfig, HnswParams, SearchQuality, StorageMode

# (constructor, args, (max_connections, ef_construction, max_elements or None), params_tests.rs lines)
PRESETS = [
    ("default", (), (32, 400, None), "7-11"),                       # auto(768)
    ("auto", (128,), (24, 300, None), "14-18"),
    ("auto", (1024,), (32, 400, None), "21-25"),
    ("auto", (256,), (24, 300, 100_000), "params.rs:41-57 (0..=256)"),
    ("auto", (257,), (32, 400, 100_000), "params.rs:41-57"),
    ("fast", (), (16, 150, 100_000), "28-33"),
    ("high_recall", (768,), (40, 600, None), "36-40"),
    ("large_dataset", (768,), (128, 2000, 750_000), "43-49"),
    ("for_dataset_size", (768, 5_000), (32, 400, 20_000), "52-57"),
    ("for_dataset_size", (768, 50_000), (128, 1600, 150_000), "60-66"),
    ("for_dataset_size", (768, 300_000), (128, 2000, 750_000), "69-75"),
    ("million_scale", (768,), (128, 1600, 1_500_000), "78-84"),
    ("max_recall", (128,), (32, 500, None), "87-91"),
    ("max_recall", (512,), (48, 800, None), "94-98"),
    ("max_recall", (1024,), (64, 1000, None), "101-105"),
    ("fast_indexing", (768,), (16, 200, None), "108-112"),
    ("custom", (32, 400, 50_000), (32, 400, 50_000), "115-121"),
    ("turbo", (), (12, 100, 100_000), "199-210"),
    ("for_dataset_size", (768, 100_000), (128, 1600, None), "233-238"),
    ("for_dataset_size", (768, 500_000), (128, 2000, None), "241-246"),
    # the small-dimension arms and the boundaries of the ranges (params.rs:72-147)
    ("for_dataset_size", (128, 10_000), (24, 200, 20_000), "params.rs:75-81"),
    ("for_dataset_size", (128, 10_001), (64, 800, 150_000), "params.rs:92-98"),
    ("for_dataset_size", (128, 100_001), (96, 1200, 750_000), "params.rs:109-115"),
    ("for_dataset_size", (128, 500_001), (64, 800, 1_500_000), "params.rs:126-132"),
    ("fast_indexing", (128,), (12, 150, None), "params.rs:237-244"),
]


@pytest.mark.parametrize("ctor,args,exp,src", PRESETS, ids=[f"{p[0]}{p[1]}@{p[3]}" for p in PRESETS])
def test_presets(ctor, args, exp, src):
    p = getattr(HnswParams, ctor)(*args)
    assert (p.max_connections, p.ef_construction) == exp[:2]
    if exp[2] is not None:
        assert p.max_elements == exp[2]
    assert p.storage_mode == StorageMode.Full                        # :115-121, :145-151, :199-210


def test_storage_mode_builders():
    """params_tests.rs:124-151: with_sq8 / with_binary = auto(dimension) + the mode; the default mode is Full"""
    p = HnswParams.with_sq8(768)
    assert p.storage_mode == StorageMode.SQ8 and (p.max_connections, p.ef_construction) == (32, 400)
    p = HnswParams.with_binary(768)
    assert p.storage_mode == StorageMode.Binary and (p.max_connections, p.ef_construction) == (32, 400)
    assert HnswParams.default().storage_mode == StorageMode.Full
    assert (int(StorageMode.Full), int(StorageMode.SQ8), int(StorageMode.Binary)) == (0, 1, 2)


# (quality, k, expected ef_search, params_tests.rs lines)
EF = [
    ("Fast", 10, 64, "154-160"), ("Balanced", 10, 128, "154-160"), ("Accurate", 10, 512, "154-160"), (("Custom", 50), 10, 50, "154-160"),
    ("Perfect", 10, 4096, "163-168"), ("Perfect", 50, 5000, "163-168"), ("Perfect", 100, 10000, "163-168"),
    ("Fast", 100, 200, "171-177"), ("Balanced", 50, 200, "171-177"), ("Accurate", 40, 640, "171-177"), ("Perfect", 50, 5000, "171-177"),
    (("Custom", 5), 10, 10, "params.rs:318: max(ef, k)"),
]
QUAL = {"Fast": po.Q_FAST, "Balanced": po.Q_BALANCED, "Accurate": po.Q_ACCURATE, "Perfect": po.Q_PERFECT}


@pytest.mark.parametrize("quality,k,exp,src", EF, ids=[f"{e[0]}-k{e[1]}@{e[3]}" for e in EF])
def test_ef_search(quality, k, exp, src):
    if isinstance(quality, tuple):
        assert SearchQuality.Custom(quality[1]).ef_search(k) == exp
        assert po.ef_search(po.Q_CUSTOM, k, quality[1]) == exp
    else:
        assert getattr(SearchQuality, quality).ef_search(k) == exp
        assert po.ef_search(QUAL[quality], k) == exp


def test_search_quality_default_and_equality():
    """:193-196 default = Balanced; the enum derives PartialEq / Eq (:180-190, :221-230 compare a value with its round trip)"""
    assert SearchQuality.default() == SearchQuality.Balanced
    assert SearchQuality.Custom(7) == SearchQuality.Custom(7) and SearchQuality.Custom(7) != SearchQuality.Custom(8)
    assert SearchQuality.Perfect != SearchQuality.Accurate and SearchQuality.Custom(128) != SearchQuality.Balanced
    assert len({SearchQuality.Fast, SearchQuality("fast"), SearchQuality.Custom(3), SearchQuality.Custom(3)}) == 2
    assert HnswParams.custom(32, 400, 50_000) == HnswParams(32, 400, 50_000, StorageMode.Full)   # :213-218


def test_dual_precision_config_defaults_and_rule():
    """native/dual_precision_tests.rs:337-342 (defaults 4 / true / 10 000) and the rule of search_with_config
    (dual_precision.rs:259-278) that decides between the int8 traversal and the plain f32 search — which is why the reference's own
    200- and 500-vector tests of "int8 traversal" are answered by `inner.search` (tests/test_oracle_dual_precision.py)"""
    c = DualPrecisionConfig()
    assert (c.oversampling_ratio, c.use_int8_traversal, c.min_index_size, c.debug_timings) == (4, True, 10_000, False)
    assert c.takes_int8_traversal(True, 10_000) and c.takes_int8_traversal(True, 1_000_000)
    assert not c.takes_int8_traversal(True, 9_999) and not c.takes_int8_traversal(True, 200)     # below min_index_size
    assert not c.takes_int8_traversal(False, 1_000_000)                                           # no trained quantiser
    assert not DualPrecisionConfig(use_int8_traversal=False).takes_int8_traversal(True, 1_000_000)
    assert DualPrecisionConfig(min_index_size=0).takes_int8_traversal(True, 1)


def test_search_with_config_passes_the_config_with_the_call(monkeypatch):
    """The rule itself (trained quantiser AND use_int8_traversal AND len >= min_index_size, dual_precision.rs:259-278) is applied
    inside the library from the handle's own state (vdb_hip_index_search_with_config; GPU side: tests/test_gpu_int8.py).  The mirror's
    part is to hand the call's DualPrecisionConfig over unchanged — checked here with the C ABI stubbed out — and the mirror's
    DualPrecisionConfig.takes_int8_traversal states the same rule for callers that want to know beforehand (test above)."""
    import numpy as np
    import velesdb_amd.index as vi

    calls = []

    class FakeLib:
        def vdb_hip_index_search_with_config(self, h, q, nq, k, ef, ratio, use_int8, min_size, ids, sc, cnt):
            calls.append((nq, k, ef, ratio, use_int8, min_size))
            np.ctypeslib.as_array(vi.C.cast(ids, vi.C.POINTER(vi.C.c_uint64)), (k,))[:] = 7
            np.ctypeslib.as_array(vi.C.cast(sc, vi.C.POINTER(vi.C.c_float)), (k,))[:] = 0.5
            np.ctypeslib.as_array(vi.C.cast(cnt, vi.C.POINTER(vi.C.c_uint32)), (1,))[:] = k
            return 0

    monkeypatch.setattr(vi, "lib", lambda: FakeLib())

    class Stub(vi.HnswIndex):
        def __init__(self):               # no handle: nothing below reaches the library
            self._dimension, self._h = 4, None

        def close(self):
            pass

    q = [0.0, 1.0, 0.0, 0.0]
    assert Stub().search_with_config(q, 2, 50) == [(7, 0.5), (7, 0.5)]
    assert calls == [(1, 2, 50, 4, 1, 10_000)]                                                    # DualPrecisionConfig::default
    calls.clear()
    Stub().search_with_config(q, 3, 64, DualPrecisionConfig(oversampling_ratio=8, use_int8_traversal=False, min_index_size=123))
    assert calls == [(1, 3, 64, 8, 0, 123)]
    with pytest.raises(AssertionError, match="dimension mismatch"):
        Stub().search_with_config([0.0, 1.0], 3, 64)
