"""Host-side mirror of the reference's operator interface for the hot path, over the C ABI.

`HnswIndex` has the surface of crates/velesdb-core/src/index/hnsw/index/*.rs (the concrete
type `Collection` holds, core/collection/types.rs:146) and implements `VectorIndex`
(index/mod.rs:30-83): same method names, argument meaning, result conventions and panics
(as AssertionError with the reference's message).  `HipDistance` mirrors `DistanceEngine`
(native/distance.rs:14-28) and `GpuAccelerator` mirrors gpu/gpu_backend.rs:33-415.

All compute happens in libvelesdb_hip.so; this file only marshals numpy buffers.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from ._ffi import check, lib
from .params import DistanceMetric, DualPrecisionConfig, HnswParams, SearchQuality

MODE_AUTO, MODE_BRUTE, MODE_HNSW, MODE_BRUTE_BF16, MODE_HNSW_INT8, MODE_BRUTE_SQ8, MODE_BRUTE_BINARY = 0, 1, 2, 3, 4, 5, 6
OPT_MAX_QUERY_TILE, OPT_SWEEP_ENGINE, OPT_SELECTOR_LEVEL, OPT_INT8_OVERSAMPLING, OPT_KERNEL_TIMING = 0, 1, 2, 3, 4
# the combining front of the host-pointer search entry points (include/velesdb_hip.h): queries per combined launch (0 = off),
# waiting window of a leader in microseconds, combined batches in flight
OPT_COMBINE_MAX_BATCH, OPT_COMBINE_WINDOW_US, OPT_COMBINE_INFLIGHT = 5, 6, 7
KIND_ENGINE, KIND_RAW = 0, 1
# enum vdb_kernel_bit (HnswIndex.last_kernels)
(KERNEL_SWEEP_VALU, KERNEL_SWEEP_MFMA_F32, KERNEL_GEMM_F32, KERNEL_SWEEP_MFMA_BF16, KERNEL_GEMM_BF16, KERNEL_GEMM_BF16_GLDS,
 KERNEL_SELECT_BF16, KERNEL_SELECT_SPLIT, KERNEL_BITS, KERNEL_SQ8, KERNEL_HNSW, KERNEL_HNSW_INT8, KERNEL_BITS_GEMM) = (1 << i for i in range(13))
SHARD_REPLICA, SHARD_RANGE = 0, 1
COMM_ID_BYTES = 128


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    n = C.c_int32(0)
    check(lib().vdb_hip_device_count(C.byref(n)))
    return int(n.value)


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    check(lib().vdb_hip_device_name(device, buf, 256))
    return buf.value.decode()


def comm_unique_id() -> bytes:
    """The 128-byte id of a new one-process-per-GPU shard group (rank 0 makes it, every rank passes it to join_group)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    check(lib().vdb_hip_comm_unique_id(buf))
    return bytes(buf)


def set_kernel_timing(on: bool) -> None:
    check(lib().vdb_hip_set_kernel_timing(1 if on else 0))


def set_max_query_tile(b: int) -> None:
    """Tuning knob of the exact sweep: queries served per corpus pass (1..32).  Results do not depend on it."""
    check(lib().vdb_hip_set_max_query_tile(b))


def set_sweep_engine(engine: int) -> None:
    """0 = vector-ALU sweep kernels (oracle mode C), 1 = matrix-core kernel for Cosine/Dot (oracle mode M)."""
    check(lib().vdb_hip_set_sweep_engine(engine))


def set_split_selector(level) -> None:
    """Large exact cosine / dot batches: 0 / False = the exact f32 matrix-core kernel for the whole batch; 1 / True =
    split-bf16 selection + exact re-scoring + proof; 2 = plain bf16 selection first (block-local candidate lists for k <= 10, the WIDE
    selection for 10 < k <= 128), level 1 as the fallback level; 3 (the library's default) = level 2 with the WIDE selection at every
    k <= 128.  Results are identical at every level."""
    check(lib().vdb_hip_set_split_selector(int(level)))


class HnswIndex:
    """HNSW index whose vectors, graph and search run on one MI355X."""

    def __init__(self, dimension: int, metric: DistanceMetric, params: Optional[HnswParams] = None,
                 device: int = 0, devices: Optional[Sequence[int]] = None, shard_mode: int = SHARD_REPLICA):
        # HnswIndex::new / with_params — constructors.rs:28-32,117-160.  `devices` (more than one) = one handle over
        # several GPUs: SHARD_RANGE (contiguous row ranges, exact searches merged) or SHARD_REPLICA (query stream split)
        self._h = C.c_void_p()
        self._dimension = int(dimension)
        self._metric = DistanceMetric(metric)
        self.params = params or HnswParams.auto(dimension)
        devs = list(devices) if devices is not None else [int(device)]
        arr = (C.c_int32 * len(devs))(*devs)
        check(lib().vdb_hip_index_create(dimension, int(self._metric), self.params.max_connections,
                                         self.params.ef_construction, self.params.max_elements, arr, len(devs),
                                         int(shard_mode), C.byref(self._h)))
        # HnswParams::storage_mode (params.rs:24-27; with_sq8 / with_binary): SQ8 / Binary collections quantise every stored
        # vector — applied at construction, as the Rust shim does (velesdb-hip/src/lib.rs `with_params`)
        if int(getattr(self.params, "storage_mode", 0)) != 0:
            self.set_storage_mode(self.params.storage_mode)

    def join_group(self, unique_id: bytes, rank: int, world: int) -> None:
        """One process per GPU: this index becomes shard `rank` of `world` (rank order = row order); exact searches
        then return the global top-k on every rank (one RCCL all-gather per batch)."""
        assert len(unique_id) == COMM_ID_BYTES
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        check(lib().vdb_hip_index_join_group(self._h, buf, rank, world))

    def shard_info(self) -> dict:
        v = [C.c_int32(0) for _ in range(5)]
        check(lib().vdb_hip_index_shard_info(self._h, *[C.byref(x) for x in v]))
        return {"n_shards": v[0].value, "shard_mode": v[1].value, "rank": v[2].value, "world": v[3].value,
                "transport": {0: "none", 1: "rccl", 2: "d2d-copy"}[v[4].value]}

    @classmethod
    def with_params(cls, dimension, metric, params, device=0):
        return cls(dimension, metric, params, device)

    @classmethod
    def new_turbo(cls, dimension, metric, device=0):  # constructors.rs:86-91
        p = HnswParams.auto(dimension)
        return cls(dimension, metric, HnswParams(p.max_connections, p.ef_construction * 3 // 2, p.max_elements), device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().vdb_hip_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- VectorIndex ------------------------------------------------------------------
    def insert(self, id: int, vector) -> None:
        """VectorIndex::insert (index/mod.rs:46).  Duplicate ids are skipped silently."""
        v = _f32(vector).reshape(-1)
        assert v.size == self._dimension, \
            f"Vector dimension mismatch: expected {self._dimension}, got {v.size}"  # trait_impl.rs:12-18
        check(lib().vdb_hip_index_insert(self._h, int(id), _ptr(v), v.size))

    def search(self, query, k: int) -> List[Tuple[int, float]]:
        """VectorIndex::search (index/mod.rs:58) = search_with_quality(Balanced) (trait_impl.rs:38-42)."""
        return self.search_with_quality(query, k, SearchQuality.Balanced)

    def remove(self, id: int) -> bool:
        r = C.c_int32(0)
        check(lib().vdb_hip_index_remove(self._h, int(id), C.byref(r)))
        return bool(r.value)

    def len(self) -> int:
        n = C.c_uint64(0)
        check(lib().vdb_hip_index_len(self._h, C.byref(n)))
        return int(n.value)

    __len__ = len

    def is_empty(self) -> bool:
        return self.len() == 0

    def dimension(self) -> int:
        return self._dimension

    def metric(self) -> DistanceMetric:
        return self._metric

    # ---- HnswIndex inherent methods ----------------------------------------------------
    def _validate(self, q: np.ndarray, what="Query"):
        assert q.shape[-1] == self._dimension, \
            f"{what} dimension mismatch: expected {self._dimension}, got {q.shape[-1]}"  # search.rs:16-23

    def _search_raw(self, queries: np.ndarray, k: int, ef: int, mode: int):
        assert queries.ndim == 2 and queries.shape[1] == self._dimension and queries.flags.c_contiguous
        nq = queries.shape[0]
        kk = max(k, 1)
        ids = np.empty((nq, kk), dtype=np.uint64)
        sc = np.empty((nq, kk), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        check(lib().vdb_hip_index_search_batch(self._h, _ptr(queries), nq, k, ef, mode, _ptr(ids), _ptr(sc),
                                               _ptr(cnt)))
        return ids, sc, cnt

    @staticmethod
    def _tuples(ids, sc, n) -> List[Tuple[int, float]]:
        return [(int(ids[i]), float(sc[i])) for i in range(int(n))]

    def search_with_quality(self, query, k: int, quality: SearchQuality) -> List[Tuple[int, float]]:
        """search.rs:59-94."""
        q = _f32(query).reshape(1, -1)
        self._validate(q)
        if quality.kind == "perfect":  # search.rs:68-70
            ids, sc, cnt = self._search_raw(q, k, 0, MODE_BRUTE)
        else:
            ids, sc, cnt = self._search_raw(q, k, quality.ef_search(k), MODE_AUTO)
        return self._tuples(ids[0], sc[0], cnt[0])

    def search_filtered(self, query, k: int, keep) -> List[Tuple[int, float]]:
        """The index side of Collection::search_with_filter (collection/search/vector.rs:164-235): post-filtering over an
        over-fetched candidate list — candidates_k = max(4 k, k + 10) through VectorIndex::search, the ids `keep(id)`
        rejects dropped, the first k survivors kept, then ordered by the metric's rule (a stable sort, as the reference's
        sort_by over partial_cmp).  Payload storage and the Filter type stay the caller's: `keep` stands for
        `filter.matches(payload(id))`."""
        candidates_k = max(k * 4, k + 10)  # vector.rs:182
        out = [(i, s) for i, s in self.search(query, candidates_k) if keep(i)][:k]
        hib = self._metric in (DistanceMetric.Cosine, DistanceMetric.DotProduct, DistanceMetric.Jaccard)  # higher_is_better
        # vector.rs:219-233: stable sort by score (partial_cmp; incomparable pairs compare Equal and keep their order)
        import functools

        def cmp(a, b):
            x, y = (b[1], a[1]) if hib else (a[1], b[1])
            return -1 if x < y else (1 if x > y else 0)
        return sorted(out, key=functools.cmp_to_key(cmp))

    def search_brute_force(self, query, k: int) -> List[Tuple[int, float]]:
        """search.rs:176-219: exact scan, raw scores, metric.sort_results order."""
        q = _f32(query).reshape(1, -1)
        self._validate(q)
        ids, sc, cnt = self._search_raw(q, k, 0, MODE_BRUTE)
        return self._tuples(ids[0], sc[0], cnt[0])

    search_brute_force_buffered = search_brute_force  # search.rs:367-370
    brute_force_search_parallel = search_brute_force  # batch.rs:223-244 (same result contract)

    def search_brute_force_gpu(self, query, k: int) -> Optional[List[Tuple[int, float]]]:
        """search.rs:229-279: None when no GPU is available."""
        if device_count() == 0:
            return None
        return self.search_brute_force(query, k)

    def search_batch_parallel(self, queries, k: int, quality: SearchQuality) -> List[List[Tuple[int, float]]]:
        """batch.rs:159-197: always the graph, one launch for the whole batch."""
        qs = _f32(queries)
        if qs.ndim == 1:
            qs = qs.reshape(1, -1)
        if qs.shape[0] == 0:
            return []
        for i in range(qs.shape[0]):
            assert qs.shape[1] == self._dimension, \
                f"Query {i} dimension mismatch: expected {self._dimension}, got {qs.shape[1]}"
        ids, sc, cnt = self._search_raw(qs, k, quality.ef_search(k), MODE_HNSW)
        return [self._tuples(ids[i], sc[i], cnt[i]) for i in range(qs.shape[0])]

    def search_with_rerank(self, query, k: int, rerank_k: int) -> List[Tuple[int, float]]:
        """search.rs:118-160: HNSW candidates (Accurate) re-scored exactly; raw scores, metric order."""
        return self.search_with_rerank_quality(query, k, rerank_k, SearchQuality.Accurate)

    def search_with_rerank_quality(self, query, k: int, rerank_k: int, initial_quality: SearchQuality):
        """search.rs:297-350 (Perfect is replaced by Accurate to avoid recursion, :305-310)."""
        q = _f32(query).reshape(1, -1)
        self._validate(q)
        if initial_quality.kind == "perfect":
            initial_quality = SearchQuality.Accurate
        ef = initial_quality.ef_search(rerank_k)
        kk = max(k, 1)
        ids = np.empty((1, kk), dtype=np.uint64)
        sc = np.empty((1, kk), dtype=np.float32)
        cnt = np.zeros(1, dtype=np.uint32)
        check(lib().vdb_hip_index_search_rerank(self._h, _ptr(q), 1, k, rerank_k, ef, _ptr(ids), _ptr(sc), _ptr(cnt)))
        return self._tuples(ids[0], sc[0], cnt[0])

    def search_multi_entry(self, queries, k: int, ef_search: int, num_probes: int):
        """NativeHnsw::search_multi_entry (native/graph.rs:288-348) for a batch: the descent's result plus up to
        min(num_probes, 4) - 1 nodes drawn from the graph's own xorshift stream as entry points of one layer-0 search.  Like the
        reference it advances that stream (query i takes the draws nq sequential calls would give it).  numpy outputs."""
        qs = _f32(queries)
        if qs.ndim == 1:
            qs = qs.reshape(1, -1)
        self._validate(qs)
        nq, kk = qs.shape[0], max(k, 1)
        ids = np.empty((nq, kk), dtype=np.uint64)
        sc = np.empty((nq, kk), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        check(lib().vdb_hip_index_search_multi_entry(self._h, _ptr(qs), nq, k, ef_search, num_probes, _ptr(ids), _ptr(sc), _ptr(cnt)))
        return ids, sc, cnt

    def train_quantizer(self, sample_rows: int = 0) -> None:
        """ScalarQuantizer::train on the first sample_rows rows (0 = min(1000, rows)) + u8 codes of every row."""
        check(lib().vdb_hip_index_train_quantizer(self._h, sample_rows))

    def is_quantizer_trained(self) -> bool:
        """DualPrecisionHnsw::is_quantizer_trained (native/dual_precision.rs:117-120), read from the handle."""
        t = C.c_int32(0)
        check(lib().vdb_hip_index_quantizer_trained(self._h, C.byref(t)))
        return bool(t.value)

    def search_with_config(self, query, k: int, ef_search: int, config: Optional[DualPrecisionConfig] = None) -> List[Tuple[int, float]]:
        """DualPrecisionHnsw::search_with_config (native/dual_precision.rs:259-278): the int8 traversal only with a trained
        quantiser, `use_int8_traversal` and at least `min_index_size` (default 10 000) vectors; otherwise the plain f32 graph
        search.  The config travels with the call (vdb_hip_index_search_with_config): the rule is applied inside the library from
        the handle's own state, `oversampling_ratio` (k * ratio of the int8 walk's best are re-scored exactly) is this call's own —
        no handle option is touched, concurrent callers may use different ratios."""
        cfg = config or DualPrecisionConfig()
        q = _f32(query).reshape(1, -1)
        self._validate(q)
        kk = max(k, 1)
        ids = np.empty((1, kk), dtype=np.uint64)
        sc = np.empty((1, kk), dtype=np.float32)
        cnt = np.zeros(1, dtype=np.uint32)
        check(lib().vdb_hip_index_search_with_config(self._h, _ptr(q), 1, k, ef_search, max(int(cfg.oversampling_ratio), 1),
                                                     1 if cfg.use_int8_traversal else 0, int(cfg.min_index_size), _ptr(ids), _ptr(sc), _ptr(cnt)))
        return self._tuples(ids[0], sc[0], cnt[0])

    def search_batch_int8(self, queries, k: int, ef_search: int):
        """DualPrecisionHnsw::search_with_config(use_int8_traversal): int8 graph walk + exact f32 re-rank."""
        qs = _f32(queries)
        if qs.ndim == 1:
            qs = qs.reshape(1, -1)
        self._validate(qs)
        ids, sc, cnt = self._search_raw(qs, k, ef_search, MODE_HNSW_INT8)
        return [self._tuples(ids[i], sc[i], cnt[i]) for i in range(qs.shape[0])]

    def enable_bf16(self) -> None:
        """Keeps a bf16 (round-to-nearest-even) copy of the rows for search_batch_brute_force_bf16."""
        check(lib().vdb_hip_index_enable_bf16(self._h))

    def search_batch_brute_force_bf16(self, queries, k: int):
        """Exact scan with bf16 rows and queries, f32 accumulation on the matrix cores
        (half_precision.rs:199-255 semantics); numpy outputs like search_batch_brute_force."""
        qs = _f32(queries)
        if qs.ndim == 1:
            qs = qs.reshape(1, -1)
        self._validate(qs)
        return self._search_raw(qs, k, 0, MODE_BRUTE_BF16)

    # ---- storage modes (core/quantization.rs) -----------------------------------------------
    def set_storage_mode(self, mode) -> None:
        """StorageMode of the owning collection (quantization.rs:17-29): the index keeps the SQ8 / binary code of every
        vector, as collection/core/crud.rs:66-82 does on upsert."""
        check(lib().vdb_hip_index_set_storage_mode(self._h, int(mode)))

    def get_quantized_bytes(self, id: int) -> bytes:
        """QuantizedVector::to_bytes / BinaryQuantizedVector::to_bytes of the stored code of `id`."""
        n = C.c_size_t(0)
        buf = (C.c_uint8 * (self._dimension + 16))()
        check(lib().vdb_hip_index_get_quantized(self._h, id, buf, len(buf), C.byref(n)))
        return bytes(buf[:n.value])

    def search_batch_sq8(self, queries, k: int):
        """Exact scan over the SQ8 codes with the reference's asymmetric `_simd` distances (quantization.rs:410-554):
        cosine / dot similarity (best = largest) or SQUARED Euclidean distance (best = smallest)."""
        qs = _f32(queries)
        if qs.ndim == 1:
            qs = qs.reshape(1, -1)
        self._validate(qs)
        return self._search_raw(qs, k, 0, MODE_BRUTE_SQ8)

    def search_batch_binary(self, queries, k: int):
        """Exact scan by Hamming distance between sign-bit codes (BinaryQuantizedVector::hamming_distance)."""
        qs = _f32(queries)
        if qs.ndim == 1:
            qs = qs.reshape(1, -1)
        self._validate(qs)
        return self._search_raw(qs, k, 0, MODE_BRUTE_BINARY)

    def search_batch_brute_force(self, queries, k: int):
        """Batched exact search (one corpus pass per tile of queries); numpy outputs."""
        qs = _f32(queries)
        if qs.ndim == 1:
            qs = qs.reshape(1, -1)
        self._validate(qs)
        return self._search_raw(qs, k, 0, MODE_BRUTE)

    def insert_batch_sequential(self, vectors: Iterable[Tuple[int, Sequence[float]]]) -> int:
        """batch.rs:128-149."""
        items = list(vectors)
        if not items:
            return 0
        ids = np.ascontiguousarray([i for i, _ in items], dtype=np.uint64)
        for _, v in items:
            assert len(v) == self._dimension, \
                f"Vector dimension mismatch: expected {self._dimension}, got {len(v)}"
        vecs = _f32([v for _, v in items])
        n = C.c_uint64(0)
        check(lib().vdb_hip_index_insert_batch(self._h, _ptr(ids), _ptr(vecs), len(items), C.byref(n)))
        return int(n.value)

    def insert_batch_parallel(self, vectors: Iterable[Tuple[int, Sequence[float]]], max_batch: int = 0) -> int:
        """batch.rs:83-108.  Batch-synchronous and deterministic here (rayon, non-deterministic, there)."""
        items = list(vectors)
        if not items:
            return 0
        ids = np.ascontiguousarray([i for i, _ in items], dtype=np.uint64)
        for _, v in items:
            assert len(v) == self._dimension, \
                f"Vector dimension mismatch: expected {self._dimension}, got {len(v)}"
        vecs = _f32([v for _, v in items])
        n = C.c_uint64(0)
        check(lib().vdb_hip_index_insert_batch_parallel(self._h, _ptr(ids), _ptr(vecs), len(items), max_batch,
                                                        C.byref(n)))
        return int(n.value)

    def build_graph(self, max_batch: int = 0) -> None:
        """Links every uploaded row that is not in the graph yet (batched GPU construction)."""
        check(lib().vdb_hip_index_build_graph(self._h, max_batch))

    def upload(self, ids, vectors) -> int:
        """Bulk upload without graph construction (exact search only until a graph exists)."""
        ids = np.ascontiguousarray(ids, dtype=np.uint64).reshape(-1)
        vecs = _f32(vectors)
        assert vecs.ndim == 2 and vecs.shape[1] == self._dimension, \
            f"Vector dimension mismatch: expected {self._dimension}, got {vecs.shape[-1]}"
        assert ids.shape[0] == vecs.shape[0], f"{ids.shape[0]} ids for {vecs.shape[0]} vectors"
        n = C.c_uint64(0)
        check(lib().vdb_hip_index_upload(self._h, _ptr(ids), _ptr(vecs), vecs.shape[0], C.byref(n)))
        return int(n.value)

    def upload_vector_store(self, directory: str) -> int:
        """Upload every vector of a flushed MmapStorage directory (core/storage/mmap.rs: vectors.idx + vectors.dat),
        in the order the store first saw the ids; no graph (as `upload`).  Returns the number of rows added."""
        n = C.c_uint64(0)
        check(lib().vdb_hip_index_upload_vector_store(self._h, directory.encode(), C.byref(n)))
        return int(n.value)

    # ---- maintenance (index/hnsw/index/vacuum.rs) ------------------------------------------
    def tombstone_count(self) -> int:
        n = C.c_uint64(0)
        check(lib().vdb_hip_index_tombstone_count(self._h, C.byref(n)))
        return int(n.value)

    def tombstone_ratio(self) -> float:  # vacuum.rs:60-67
        total = self.node_count()
        return 0.0 if total == 0 else self.tombstone_count() / total

    def needs_vacuum(self) -> bool:  # vacuum.rs:74-76
        return self.tombstone_ratio() > 0.2

    def vacuum(self) -> int:
        """Rebuild without tombstones (vacuum.rs:110-184); returns the number of vectors in the rebuilt index."""
        n = C.c_uint64(0)
        check(lib().vdb_hip_index_vacuum(self._h, C.byref(n)))
        self.params = HnswParams.auto(self._dimension)
        return int(n.value)

    def set_searching_mode(self) -> None:  # search.rs:380-384: no-op for the native engine
        pass

    def save(self, directory: str, basename: Optional[str] = None) -> None:
        """HnswIndex::save (constructors.rs:255-287): native_hnsw.{vectors,graph} + native_mappings.bin +
        native_meta.bin in `directory`.  With a `basename`: only the graph + vector files under that name
        (NativeHnsw::file_dump, backend_adapter.rs:184-261)."""
        os.makedirs(directory, exist_ok=True)
        if basename is None:
            check(lib().vdb_hip_index_save_dir(self._h, directory.encode()))
        else:
            check(lib().vdb_hip_index_save_reference_files(self._h, directory.encode(), basename.encode()))

    @classmethod
    def load(cls, directory: str, dimension: int = 0, metric: Optional[DistanceMetric] = None, device: int = 0):
        """HnswIndex::load (constructors.rs:190-253); dimension / metric are read from the metadata file, the
        arguments exist for API compatibility like the reference's."""
        h = C.c_void_p()
        check(lib().vdb_hip_index_load_dir(directory.encode(), device, C.byref(h)))
        self = cls.__new__(cls)
        self._h = h
        d, m = C.c_uint32(0), C.c_int32(0)
        check(lib().vdb_hip_index_dimension(h, C.byref(d)))
        check(lib().vdb_hip_index_metric(h, C.byref(m)))
        self._dimension, self._metric = int(d.value), DistanceMetric(m.value)
        self.params = HnswParams.auto(self._dimension)
        return self

    def load_reference_files(self, directory: str, basename: str = "native_hnsw") -> None:
        check(lib().vdb_hip_index_load_reference_files(self._h, directory.encode(), basename.encode()))

    def sweep_arith_mode(self, k: int) -> str:
        """'M' if exact searches with this k run on the matrix-core kernel (oracle mode M), else 'C'."""
        m = C.c_int32(0)
        check(lib().vdb_hip_index_sweep_arith_mode(self._h, k, C.byref(m)))
        return "M" if m.value else "C"

    # ---- introspection ------------------------------------------------------------------
    def node_count(self) -> int:
        n = C.c_uint64(0)
        check(lib().vdb_hip_index_node_count(self._h, C.byref(n)))
        return int(n.value)

    def neighbors(self, layer: int, node: int) -> List[int]:
        buf = np.empty(1024, dtype=np.uint32)
        n = C.c_uint32(0)
        check(lib().vdb_hip_index_get_neighbors(self._h, layer, node, _ptr(buf), buf.size, C.byref(n)))
        return buf[: n.value].tolist()

    def graph_info(self):
        nl, ml, ep = C.c_uint32(0), C.c_uint32(0), C.c_int64(-1)
        check(lib().vdb_hip_index_graph_info(self._h, C.byref(nl), C.byref(ml), C.byref(ep)))
        return int(nl.value), int(ml.value), int(ep.value)

    def last_search_stats(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(lib().vdb_hip_index_last_search_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def build_stats(self):
        """Construction counters, cumulative since the handle was created: (rows whose distance was evaluated by the insert kernel's
        search_layer / select_neighbors phases, distance phases = dependent memory round trips, nodes inserted, the rows of the first
        count that were select_neighbors evaluations — re-reads of a node's <= ef_construction candidate rows, cache hits)."""
        a, b, c, d = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(lib().vdb_hip_index_build_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return int(a.value), int(b.value), int(c.value), int(d.value)

    def last_prefetch_hits(self):
        """Expansions of the last graph search batch whose neighbour ids had been requested one pop ahead (the walk's prediction)."""
        a = C.c_uint64(0)
        check(lib().vdb_hip_index_last_prefetch_hits(self._h, C.byref(a)))
        return int(a.value)

    def last_split_stats(self):
        """(queries, unproven) of the last split-selector batch: unproven ones were answered by the exact fallback kernel."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        check(lib().vdb_hip_index_last_split_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def set_option(self, option: int, value: int) -> None:
        """Per-handle tuning option (OPT_*); value < 0 = follow the process-wide default again.  Results never depend on it."""
        check(lib().vdb_hip_index_set_option(self._h, int(option), int(value)))

    def get_option(self, option: int) -> int:
        v = C.c_int64(0)
        check(lib().vdb_hip_index_get_option(self._h, int(option), C.byref(v)))
        return int(v.value)

    def combine_stats(self):
        """Counters of the combining front: (launches, calls, queries, largest batch) since the handle was created."""
        v = [C.c_uint64(0) for _ in range(4)]
        check(lib().vdb_hip_index_combine_stats(self._h, *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def last_select_level(self) -> int:
        """Selection level the last exact batch of this handle ran at: 0 = exact kernels, 1 = split-bf16, 2 = plain bf16 (Cosine: normalised
        images), 3 = the SQ8 storage mode's, 4 = the WIDE selection of 10 < k <= 128 (csrc/sweep_wide.hip)."""
        v = C.c_int32(0)
        check(lib().vdb_hip_index_last_select_level(self._h, C.byref(v)))
        return int(v.value)

    def last_kernels(self) -> int:
        """Bit set (KERNEL_*) of the kernel families that served the last search call."""
        v = C.c_uint32(0)
        check(lib().vdb_hip_index_last_kernels(self._h, C.byref(v)))
        return int(v.value)

    def last_selection_ms(self):
        """(total ms, launches) of the selection kernel in the last search call (kernel timing on)."""
        ms, n = C.c_float(0), C.c_uint32(0)
        check(lib().vdb_hip_index_last_selection_ms(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def last_kernel_ms(self):
        ms, n = C.c_float(0), C.c_uint32(0)
        check(lib().vdb_hip_index_last_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    # ---- device-pointer entry points (torch / raw HIP pointers) --------------------------
    def upload_dev(self, id_base: int, d_ptr: int, n: int, stream: int = 0) -> None:
        check(lib().vdb_hip_index_upload_dev(self._h, id_base, C.c_void_p(d_ptr), n, C.c_void_p(stream)))

    def search_batch_dev(self, d_queries: int, nq: int, k: int, ef: int, mode: int, d_ids: int, d_scores: int,
                         d_n: int, stream: int = 0) -> None:
        check(lib().vdb_hip_index_search_batch_dev(self._h, C.c_void_p(d_queries), nq, k, ef, mode,
                                                   C.c_void_p(d_ids), C.c_void_p(d_scores), C.c_void_p(d_n),
                                                   C.c_void_p(stream)))


class NativeHnswIndex(HnswIndex):
    """index/hnsw/native_index.rs: the reference's second `impl VectorIndex` (:403-427) over the same NativeHnsw graph.  What
    differs from HnswIndex is the search entry point: `search_with_quality` ALWAYS walks the graph with
    ef = quality.ef_search(k) (:230-249) — no exact-scan shortcut for Perfect or for indexes of <= 100 vectors, scores always
    through transform_score; removed ids are dropped after the cut (:241-247, mappings.get_id).  Deviation kept from
    HnswIndex: a duplicate id is skipped (the reference re-inserts the vector under the existing internal index, :256-263,
    which links one node twice)."""

    def search_with_quality(self, query, k: int, quality: SearchQuality) -> List[Tuple[int, float]]:
        q = _f32(query).reshape(1, -1)
        self._validate(q)
        ids, sc, cnt = self._search_raw(q, k, quality.ef_search(k), MODE_HNSW)
        return self._tuples(ids[0], sc[0], cnt[0])

    def insert_batch(self, items) -> None:  # native_index.rs:275-295
        self.insert_batch_parallel(items)


class HipDistance:
    """DistanceEngine (native/distance.rs:14-28) whose batch_distance runs on the GPU."""

    def __init__(self, metric: DistanceMetric, device: int = 0):
        self._metric = DistanceMetric(metric)
        self.device = device

    def metric(self) -> DistanceMetric:
        return self._metric

    def batch_distance(self, query, candidates) -> np.ndarray:
        q = _f32(query).reshape(-1)
        c = _f32(candidates)
        if c.size == 0:
            return np.empty(0, dtype=np.float32)
        assert c.ndim == 2 and c.shape[1] == q.size, "Vector dimensions must match"
        out = np.empty(c.shape[0], dtype=np.float32)
        check(lib().vdb_hip_batch_distance(self.device, int(self._metric), KIND_ENGINE, _ptr(q), _ptr(c),
                                           c.shape[0], q.size, _ptr(out)))
        return out

    def distance(self, a, b) -> float:
        return float(self.batch_distance(a, _f32(b).reshape(1, -1))[0])


class GpuAccelerator:
    """gpu/gpu_backend.rs:33-415: new() -> None without a GPU; batch_* return raw similarities."""

    def __init__(self, device: int = 0):
        self.device = device

    @staticmethod
    def new(device: int = 0) -> Optional["GpuAccelerator"]:
        return GpuAccelerator(device) if device_count() > 0 else None

    @staticmethod
    def is_available() -> bool:
        return device_count() > 0

    def _batch(self, metric, vectors, query, dimension) -> np.ndarray:
        v = _f32(vectors).reshape(-1)
        q = _f32(query).reshape(-1)
        if dimension == 0 or v.size == 0:
            return np.empty(0, dtype=np.float32)  # gpu_backend.rs:163-169
        assert q.size == dimension, f"Query dimension mismatch: expected {dimension}, got {q.size}"
        n = v.size // dimension
        if n == 0:
            return np.empty(0, dtype=np.float32)
        out = np.empty(n, dtype=np.float32)
        check(lib().vdb_hip_batch_distance(self.device, int(metric), KIND_RAW, _ptr(q), _ptr(v), n, dimension,
                                           _ptr(out)))
        return out

    def batch_cosine_similarity(self, vectors, query, dimension) -> np.ndarray:
        return self._batch(DistanceMetric.Cosine, vectors, query, dimension)

    def batch_euclidean_distance(self, vectors, query, dimension) -> np.ndarray:
        return self._batch(DistanceMetric.Euclidean, vectors, query, dimension)

    def batch_dot_product(self, vectors, query, dimension) -> np.ndarray:
        return self._batch(DistanceMetric.DotProduct, vectors, query, dimension)
