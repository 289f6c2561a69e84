"""HnswParams presets and SearchQuality — host logic mirrored from
crates/velesdb-core/src/index/hnsw/params.rs (cited per function)."""
from __future__ import annotations

import enum
from dataclasses import dataclass


class DistanceMetric(enum.IntEnum):
    """core/distance.rs:16-38; discriminants = on-disk order (constructors.rs:204-210)."""
    Cosine = 0
    Euclidean = 1
    DotProduct = 2
    Hamming = 3
    Jaccard = 4

    def higher_is_better(self) -> bool:  # core/distance.rs:76-82
        return self in (DistanceMetric.Cosine, DistanceMetric.DotProduct, DistanceMetric.Jaccard)


class StorageMode(enum.IntEnum):
    """core/quantization.rs:17-29"""
    Full = 0
    SQ8 = 1
    Binary = 2


@dataclass(frozen=True)
class HnswParams:
    """params.rs:14-28.  `storage_mode`: in the reference the collection layer acts on it (collection/core/crud.rs:66-82 quantises every
    upserted vector); here the index constructor does (`HnswIndex.__init__` -> set_storage_mode), like the Rust shim."""
    max_connections: int
    ef_construction: int
    max_elements: int = 100_000
    storage_mode: StorageMode = StorageMode.Full

    @staticmethod
    def default() -> "HnswParams":  # params.rs:28-32 (impl Default): auto(768)
        return HnswParams.auto(768)

    @staticmethod
    def auto(dimension: int) -> "HnswParams":  # params.rs:41-57
        return HnswParams(24, 300, 100_000) if dimension <= 256 else HnswParams(32, 400, 100_000)

    @staticmethod
    def for_dataset_size(dimension: int, expected_vectors: int) -> "HnswParams":  # params.rs:72-147
        small = dimension <= 256
        if expected_vectors <= 10_000:
            return HnswParams(24, 200, 20_000) if small else HnswParams(32, 400, 20_000)
        if expected_vectors <= 100_000:
            return HnswParams(64, 800, 150_000) if small else HnswParams(128, 1600, 150_000)
        if expected_vectors <= 500_000:
            return HnswParams(96, 1200, 750_000) if small else HnswParams(128, 2000, 750_000)
        return HnswParams(64, 800, 1_500_000) if small else HnswParams(128, 1600, 1_500_000)

    @staticmethod
    def large_dataset(dimension: int) -> "HnswParams":  # params.rs:149-151
        return HnswParams.for_dataset_size(dimension, 500_000)

    @staticmethod
    def million_scale(dimension: int) -> "HnswParams":  # params.rs:155-157
        return HnswParams.for_dataset_size(dimension, 1_000_000)

    @staticmethod
    def fast() -> "HnswParams":  # params.rs:161-169
        return HnswParams(16, 150, 100_000)

    @staticmethod
    def turbo() -> "HnswParams":  # params.rs:189-197
        return HnswParams(12, 100, 100_000)

    @staticmethod
    def high_recall(dimension: int) -> "HnswParams":  # params.rs:201-208: auto + (8, 200)
        b = HnswParams.auto(dimension)
        return HnswParams(b.max_connections + 8, b.ef_construction + 200, b.max_elements)

    @staticmethod
    def max_recall(dimension: int) -> "HnswParams":  # params.rs:212-233
        if dimension <= 256:
            return HnswParams(32, 500, 100_000)
        return HnswParams(48, 800, 100_000) if dimension <= 768 else HnswParams(64, 1000, 100_000)

    @staticmethod
    def fast_indexing(dimension: int) -> "HnswParams":  # params.rs:237-244: auto halved, M >= 8
        b = HnswParams.auto(dimension)
        return HnswParams(max(b.max_connections // 2, 8), b.ef_construction // 2, b.max_elements)

    @staticmethod
    def with_sq8(dimension: int) -> "HnswParams":  # params.rs:263-268
        b = HnswParams.auto(dimension)
        return HnswParams(b.max_connections, b.ef_construction, b.max_elements, StorageMode.SQ8)

    @staticmethod
    def with_binary(dimension: int) -> "HnswParams":  # params.rs:271-276
        b = HnswParams.auto(dimension)
        return HnswParams(b.max_connections, b.ef_construction, b.max_elements, StorageMode.Binary)

    @staticmethod
    def custom(max_connections: int, ef_construction: int, max_elements: int) -> "HnswParams":  # :248-259
        return HnswParams(max_connections, ef_construction, max_elements)


class SearchQuality:
    """params.rs:287-320.  Use the class attributes or SearchQuality.Custom(ef)."""

    def __init__(self, kind: str, ef: int = 0):
        self.kind, self.ef = kind, ef

    def ef_search(self, k: int) -> int:  # params.rs:309-319
        return {"fast": max(64, k * 2), "balanced": max(128, k * 4), "accurate": max(512, k * 16),
                "perfect": max(4096, k * 100), "custom": max(self.ef, k)}[self.kind]

    @staticmethod
    def Custom(ef: int) -> "SearchQuality":
        return SearchQuality("custom", ef)

    def __repr__(self):
        return f"SearchQuality.{self.kind}" + (f"({self.ef})" if self.kind == "custom" else "")

    def __eq__(self, other):  # #[derive(PartialEq, Eq)]: Custom(a) == Custom(b) iff a == b
        return isinstance(other, SearchQuality) and (self.kind, self.ef if self.kind == "custom" else 0) == \
            (other.kind, other.ef if other.kind == "custom" else 0)

    def __hash__(self):
        return hash((self.kind, self.ef if self.kind == "custom" else 0))

    @staticmethod
    def default() -> "SearchQuality":  # #[default] Balanced (params.rs:289-292)
        return SearchQuality.Balanced


SearchQuality.Fast = SearchQuality("fast")
SearchQuality.Balanced = SearchQuality("balanced")
SearchQuality.Accurate = SearchQuality("accurate")
SearchQuality.Perfect = SearchQuality("perfect")


@dataclass(frozen=True)
class DualPrecisionConfig:
    """native/dual_precision.rs:30-55 (Default: oversampling 4, int8 traversal on, min_index_size 10 000, no timings)."""
    oversampling_ratio: int = 4
    use_int8_traversal: bool = True
    min_index_size: int = 10_000
    debug_timings: bool = False

    def takes_int8_traversal(self, quantizer_trained: bool, index_len: int) -> bool:
        """The rule of DualPrecisionHnsw::search_with_config (:259-278): without a trained quantiser, with int8 traversal switched
        off, or below min_index_size the plain f32 search answers (`inner.search`); otherwise search_int8_traversal."""
        return bool(quantizer_trained) and self.use_int8_traversal and index_len >= self.min_index_size

