"""ctypes binding of the CPU oracle (oracle/vdb_oracle.cpp).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke().  The product package (velesdb_amd/) never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvdb_oracle.so")

COSINE, EUCLIDEAN, DOT, HAMMING, JACCARD = 0, 1, 2, 3, 4
METRICS = {"cosine": COSINE, "euclidean": EUCLIDEAN, "dot": DOT, "hamming": HAMMING, "jaccard": JACCARD}
MODE_R, MODE_C, MODE_SCALAR, MODE_NATIVE, MODE_R_NOFMA, MODE_M = 0, 1, 2, 3, 4, 5
TIE_REFERENCE, TIE_CANONICAL = 0, 1
Q_FAST, Q_BALANCED, Q_ACCURATE, Q_PERFECT, Q_CUSTOM = 0, 1, 2, 3, 4


def build(force: bool = False) -> str:
    """Compile the oracle with oracle/Makefile (g++); returns the .so path."""
    src = os.path.join(_HERE, "vdb_oracle.cpp")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "vdb_oracle.h")))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libvdb_oracle.so"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        # VDB_ORACLE_SO: another build of the same sources (oracle/Makefile's `sanitize` target; tools/oracle_sanitize.sh)
        _lib = C.CDLL(os.environ.get("VDB_ORACLE_SO") or build())
        _declare(_lib)
    return _lib


_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def _declare(L):
    sz = C.c_size_t
    for name in ("vo_dot", "vo_sql2", "vo_euclidean", "vo_cosine"):
        f = getattr(L, name)
        f.restype, f.argtypes = C.c_float, [C.c_int, _f32p, _f32p, sz]
    L.vo_norm_sq.restype, L.vo_norm_sq.argtypes = C.c_float, [C.c_int, _f32p, sz]
    L.vo_norm.restype, L.vo_norm.argtypes = C.c_float, [_f32p, sz]
    for name in ("vo_hamming", "vo_jaccard", "vo_dot_simd8", "vo_sql2_simd8", "vo_cosine_simd8", "vo_dot_c_plain", "vo_sql2_c_plain"):
        f = getattr(L, name)
        f.restype, f.argtypes = C.c_float, [_f32p, _f32p, sz]
    L.vo_hamming_binary.restype, L.vo_hamming_binary.argtypes = C.c_uint32, [_u64p, _u64p, sz]
    for name in ("vo_distance", "vo_compute_distance"):
        f = getattr(L, name)
        f.restype, f.argtypes = C.c_float, [C.c_int, C.c_int, _f32p, _f32p, sz]
    for name in ("vo_batch_distance", "vo_batch_compute_distance"):
        f = getattr(L, name)
        f.restype, f.argtypes = None, [C.c_int, C.c_int, _f32p, _f32p, sz, sz, _f32p]
    L.vo_transform_score.restype, L.vo_transform_score.argtypes = C.c_float, [C.c_int, C.c_float]
    L.vo_higher_is_better.restype, L.vo_higher_is_better.argtypes = C.c_int, [C.c_int]
    L.vo_ef_search.restype, L.vo_ef_search.argtypes = C.c_uint64, [C.c_int, C.c_uint64, C.c_uint64]
    L.vo_total_cmp.restype, L.vo_total_cmp.argtypes = C.c_int, [C.c_float, C.c_float]
    L.vo_sort_results.restype, L.vo_sort_results.argtypes = None, [C.c_int, _u64p, _f32p, C.c_uint64]
    L.vo_xorshift64_next.restype, L.vo_xorshift64_next.argtypes = C.c_uint64, [C.POINTER(C.c_uint64)]
    L.vo_random_layer.restype, L.vo_random_layer.argtypes = C.c_uint32, [C.POINTER(C.c_uint64), C.c_double]
    L.vo_heap_order_after_pushes.restype = None
    L.vo_heap_order_after_pushes.argtypes = [_f32p, _u64p, sz, C.c_int, _u64p]
    vp = C.c_void_p
    L.vo_hnsw_new.restype, L.vo_hnsw_new.argtypes = vp, [C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32]
    L.vo_hnsw_free.restype, L.vo_hnsw_free.argtypes = None, [vp]
    L.vo_hnsw_set_alpha.restype, L.vo_hnsw_set_alpha.argtypes = None, [vp, C.c_float]
    L.vo_hnsw_insert.restype, L.vo_hnsw_insert.argtypes = C.c_uint64, [vp, _f32p]
    L.vo_hnsw_len.restype, L.vo_hnsw_len.argtypes = C.c_uint64, [vp]
    L.vo_hnsw_set_build_tie.restype, L.vo_hnsw_set_build_tie.argtypes = None, [vp, C.c_int]
    L.vo_hnsw_set_build_threads.restype, L.vo_hnsw_set_build_threads.argtypes = None, [vp, C.c_uint32]
    L.vo_hnsw_insert_batch_sync.restype, L.vo_hnsw_insert_batch_sync.argtypes = None, [vp, _f32p, C.c_uint64]
    L.vo_build_batch_size.restype, L.vo_build_batch_size.argtypes = C.c_uint32, [C.c_uint64, C.c_uint32]
    L.vo_hnsw_build_batched.restype = None
    L.vo_hnsw_build_batched.argtypes = [vp, _f32p, C.c_uint64, C.c_uint32]
    L.vo_hnsw_max_layer.restype, L.vo_hnsw_max_layer.argtypes = C.c_uint32, [vp]
    L.vo_hnsw_entry_point.restype, L.vo_hnsw_entry_point.argtypes = C.c_int64, [vp]
    L.vo_hnsw_num_layers.restype, L.vo_hnsw_num_layers.argtypes = C.c_uint32, [vp]
    L.vo_hnsw_neighbors.restype = C.c_uint32
    L.vo_hnsw_neighbors.argtypes = [vp, C.c_uint32, C.c_uint64, _u64p, C.c_uint32]
    L.vo_hnsw_vector.restype, L.vo_hnsw_vector.argtypes = C.POINTER(C.c_float), [vp, C.c_uint64]
    L.vo_hnsw_search.restype = C.c_uint32
    L.vo_hnsw_search.argtypes = [vp, _f32p, C.c_uint32, C.c_uint32, C.c_int, _u64p, _f32p]
    L.vo_hnsw_search_multi_entry.restype = C.c_uint32
    L.vo_hnsw_search_multi_entry.argtypes = [vp, _f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _u64p, _f32p]
    L.vo_hnsw_rng_state.restype = C.c_uint64
    L.vo_hnsw_rng_state.argtypes = [vp]
    L.vo_hnsw_search_batch.restype = None
    L.vo_hnsw_search_batch.argtypes = [vp, _f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, _u64p,
                                       _f32p, _u32p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.vo_hnsw_last_stats.restype = None
    L.vo_hnsw_last_stats.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.vo_hnsw_search_layer_single.restype = C.c_uint64
    L.vo_hnsw_search_layer_single.argtypes = [vp, _f32p, C.c_uint64, C.c_uint32]
    L.vo_hnsw_search_layer.restype = C.c_uint32
    L.vo_hnsw_search_layer.argtypes = [vp, _f32p, _u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _u64p,
                                       _f32p, C.c_uint32]
    L.vo_hnsw_select_neighbors.restype = C.c_uint32
    L.vo_hnsw_select_neighbors.argtypes = [vp, _u64p, _f32p, C.c_uint32, C.c_uint32, _u64p]
    L.vo_hnsw_file_dump.restype, L.vo_hnsw_file_dump.argtypes = C.c_int, [vp, C.c_char_p, C.c_char_p]
    L.vo_hnsw_file_load.restype = vp
    L.vo_hnsw_file_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.vo_index_new.restype, L.vo_index_new.argtypes = vp, [C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32]
    L.vo_index_new_auto.restype, L.vo_index_new_auto.argtypes = vp, [C.c_uint32, C.c_int, C.c_int]
    L.vo_index_free.restype, L.vo_index_free.argtypes = None, [vp]
    L.vo_index_insert.restype, L.vo_index_insert.argtypes = C.c_int, [vp, C.c_uint64, _f32p]
    L.vo_index_remove.restype, L.vo_index_remove.argtypes = C.c_int, [vp, C.c_uint64]
    L.vo_index_len.restype, L.vo_index_len.argtypes = C.c_uint64, [vp]
    L.vo_index_graph.restype, L.vo_index_graph.argtypes = vp, [vp]
    L.vo_index_search_with_quality.restype = C.c_uint32
    L.vo_index_search_with_quality.argtypes = [vp, _f32p, C.c_uint32, C.c_int, C.c_uint32, C.c_int, _u64p, _f32p]
    L.vo_index_search_brute_force.restype = C.c_uint32
    L.vo_index_search_brute_force.argtypes = [vp, _f32p, C.c_uint32, _u64p, _f32p]
    L.vo_index_search_with_rerank.restype = C.c_uint32
    L.vo_index_search_with_rerank.argtypes = [vp, _f32p, C.c_uint32, C.c_uint32, _u64p, _f32p]
    L.vo_index_search_with_rerank_quality.restype = C.c_uint32
    L.vo_index_search_with_rerank_quality.argtypes = [vp, _f32p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int,
                                                      _u64p, _f32p]
    L.vo_index_search_batch.restype = None
    L.vo_index_search_batch.argtypes = [vp, _f32p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int,
                                        C.c_uint32, _u64p, _f32p, _u32p]
    L.vo_scan_topk.restype = None
    L.vo_scan_topk.argtypes = [C.c_int, C.c_int, _f32p, C.c_uint64, C.c_uint32, _f32p, C.c_uint32, C.c_uint32,
                               C.c_uint32, _u64p, _f32p]
    _u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    L.vo_sq_train.restype, L.vo_sq_train.argtypes = None, [_f32p, C.c_uint64, C.c_uint32, _f32p, _f32p, _f32p]
    L.vo_sq_quantize.restype, L.vo_sq_quantize.argtypes = None, [_f32p, C.c_uint64, C.c_uint32, _f32p, _f32p, _u8p]
    L.vo_sq_l2.restype, L.vo_sq_l2.argtypes = C.c_uint32, [_u8p, _u8p, C.c_uint32]
    L.vo_dual_search_int8.restype = C.c_uint32
    L.vo_dual_search_int8.argtypes = [vp, _u8p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _u64p,
                                      _f32p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.vo_round_bf16.restype, L.vo_round_bf16.argtypes = None, [_f32p, _f32p, C.c_uint64]
    _u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
    L.vo_binary_quantize.restype, L.vo_binary_quantize.argtypes = None, [_f32p, C.c_uint32, _u8p]
    L.vo_binary_hamming.restype, L.vo_binary_hamming.argtypes = C.c_uint32, [_u8p, _u8p, C.c_uint32]
    L.vo_sq8_quantize.restype = None
    L.vo_sq8_quantize.argtypes = [_f32p, C.c_uint32, _u8p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.vo_sq8_dequantize.restype, L.vo_sq8_dequantize.argtypes = None, [_u8p, C.c_float, C.c_float, C.c_uint32, _f32p]
    L.vo_sq8_dot.restype, L.vo_sq8_dot.argtypes = C.c_float, [_f32p, _u8p, C.c_float, C.c_float, C.c_uint32]
    L.vo_sq8_l2sq.restype, L.vo_sq8_l2sq.argtypes = C.c_float, [_f32p, _u8p, C.c_float, C.c_float, C.c_uint32, C.c_int]
    L.vo_sq8_cosine.restype, L.vo_sq8_cosine.argtypes = C.c_float, [_f32p, _u8p, C.c_float, C.c_float, C.c_uint32, C.c_int]
    L.vo_sq8_norm_sq.restype, L.vo_sq8_norm_sq.argtypes = C.c_float, [_u8p, C.c_float, C.c_float, C.c_uint32]
    L.vo_scan_topk_sq8.restype = None
    L.vo_scan_topk_sq8.argtypes = [C.c_int, _f32p, C.c_uint64, C.c_uint32, _f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                   np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS"), _f32p]
    L.vo_scan_topk_binary.restype = None
    L.vo_scan_topk_binary.argtypes = [_f32p, C.c_uint64, C.c_uint32, _f32p, C.c_uint32, C.c_uint32,
                                      np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS"), _f32p]
    L.vo_scan_topk_bf16.restype = None
    L.vo_scan_topk_bf16.argtypes = [C.c_int, _f32p, C.c_uint64, C.c_uint32, _f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                    _u64p, _f32p]
    L.vo_cpu_has_avx512f.restype = C.c_int
    L.vo_alloc_spread.restype = C.c_void_p
    L.vo_alloc_spread.argtypes = [_f32p, C.c_uint64, C.c_uint32, C.c_uint32]
    L.vo_free_spread.restype = None
    L.vo_free_spread.argtypes = [C.c_void_p]
    L.vo_hnsw_spread.restype = None
    L.vo_hnsw_spread.argtypes = [C.c_void_p, C.c_uint32]
    L.vo_merge_shard_records.restype = None
    L.vo_merge_shard_records.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS"), C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_int, _u64p, _f32p,
                                         np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")]
    L.vo_build_info.restype = C.c_char_p


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- scalar kernels --------------------------------------------------------
def dot(a, b, mode=MODE_R):
    a, b = _f(a), _f(b)
    assert a.shape == b.shape, "Vector dimensions must match"
    return float(lib().vo_dot(mode, a, b, a.size))


def sql2(a, b, mode=MODE_R):
    a, b = _f(a), _f(b)
    assert a.shape == b.shape, "Vector dimensions must match"
    return float(lib().vo_sql2(mode, a, b, a.size))


def euclidean(a, b, mode=MODE_R):
    a, b = _f(a), _f(b)
    assert a.shape == b.shape, "Vector dimensions must match"
    return float(lib().vo_euclidean(mode, a, b, a.size))


def cosine(a, b, mode=MODE_R):
    a, b = _f(a), _f(b)
    assert a.shape == b.shape, "Vector dimensions must match"
    return float(lib().vo_cosine(mode, a, b, a.size))


def norm(a):
    a = _f(a)
    return float(lib().vo_norm(a, a.size))


def norm_sq(a, mode=MODE_R):
    a = _f(a)
    return float(lib().vo_norm_sq(mode, a, a.size))


def hamming(a, b):
    a, b = _f(a), _f(b)
    assert a.shape == b.shape, "Vector dimensions must match"
    return float(lib().vo_hamming(a, b, a.size))


def jaccard(a, b):
    a, b = _f(a), _f(b)
    assert a.shape == b.shape, "Vector dimensions must match"
    return float(lib().vo_jaccard(a, b, a.size))


def hamming_binary(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    return int(lib().vo_hamming_binary(a, b, a.size))


def dot_c_plain(a, b):
    """Mode C's dot product by the plain per-element loop (what the vectorised reduction of `dot(..., MODE_C)` must reproduce)."""
    a, b = _f(a), _f(b)
    return float(lib().vo_dot_c_plain(a, b, a.size))


def sql2_c_plain(a, b):
    a, b = _f(a), _f(b)
    return float(lib().vo_sql2_c_plain(a, b, a.size))


def dot_simd8(a, b):
    a, b = _f(a), _f(b)
    return float(lib().vo_dot_simd8(a, b, a.size))


def sql2_simd8(a, b):
    a, b = _f(a), _f(b)
    return float(lib().vo_sql2_simd8(a, b, a.size))


def cosine_simd8(a, b):
    a, b = _f(a), _f(b)
    return float(lib().vo_cosine_simd8(a, b, a.size))


def distance(metric, a, b, mode=MODE_R):
    a, b = _f(a), _f(b)
    return float(lib().vo_distance(metric, mode, a, b, a.size))


def compute_distance(metric, a, b, mode=MODE_R):
    a, b = _f(a), _f(b)
    return float(lib().vo_compute_distance(metric, mode, a, b, a.size))


def batch_distance(metric, q, rows, mode=MODE_R):
    q, rows = _f(q), _f(rows)
    out = np.empty(rows.shape[0], dtype=np.float32)
    lib().vo_batch_distance(metric, mode, q, rows, rows.shape[0], rows.shape[1], out)
    return out


def batch_compute_distance(metric, q, rows, mode=MODE_R):
    q, rows = _f(q), _f(rows)
    out = np.empty(rows.shape[0], dtype=np.float32)
    lib().vo_batch_compute_distance(metric, mode, q, rows, rows.shape[0], rows.shape[1], out)
    return out


def transform_score(metric, d):
    return float(lib().vo_transform_score(metric, float(d)))


def ef_search(quality, k, custom=0):
    return int(lib().vo_ef_search(quality, custom, k))


def total_cmp(a, b):
    return int(lib().vo_total_cmp(float(a), float(b)))


def sort_results(metric, results):
    """DistanceMetric::sort_results (core/distance.rs:95-103) on a list of (id, score): stable, direction by metric"""
    ids = np.array([r[0] for r in results], dtype=np.uint64)
    sc = np.array([r[1] for r in results], dtype=np.float32)
    lib().vo_sort_results(metric, ids, sc, len(results))
    return list(zip(ids.tolist(), sc.tolist()))


def higher_is_better(metric) -> bool:
    return bool(lib().vo_higher_is_better(metric))


def random_layers(n, M, seed=0x5DEECE66D1A4B5B5):
    import math
    st = C.c_uint64(seed)
    lm = 1.0 / math.log(M)
    return [int(lib().vo_random_layer(C.byref(st), lm)) for _ in range(n)]


def xorshift_stream(n, seed=0x5DEECE66D1A4B5B5):
    st = C.c_uint64(seed)
    return [int(lib().vo_xorshift64_next(C.byref(st))) for _ in range(n)]


def heap_order(dists, nodes, min_heap=False):
    d = _f(dists)
    n = np.ascontiguousarray(nodes, dtype=np.uint64)
    out = np.empty_like(n)
    lib().vo_heap_order_after_pushes(d, n, n.size, 1 if min_heap else 0, out)
    return out.tolist()


def scan_topk(metric, rows, queries, k, mode=MODE_R, nthreads=1):
    rows, queries = _f(rows), _f(queries)
    if queries.ndim == 1:
        queries = queries[None, :]
    nq = queries.shape[0]
    ids = np.empty((nq, k), dtype=np.uint64)
    sc = np.empty((nq, k), dtype=np.float32)
    lib().vo_scan_topk(metric, mode, rows, rows.shape[0], rows.shape[1], queries, nq, k, nthreads, ids, sc)
    return ids, sc


class SpreadRows:
    """A copy of a row-major f32 corpus whose pages were first touched by the pool threads that scan_topk(…, nthreads)
    reads them with (NUMA placement of the CPU baseline); `.array` is a numpy view, freed with the object."""

    def __init__(self, rows, nthreads):
        rows = _f(rows)
        self.shape = rows.shape
        self._p = lib().vo_alloc_spread(rows, rows.shape[0], rows.shape[1], nthreads)
        buf = (C.c_float * (rows.shape[0] * rows.shape[1])).from_address(self._p)
        self.array = np.frombuffer(buf, dtype=np.float32).reshape(rows.shape)

    def __del__(self):
        try:
            if getattr(self, "_p", None):
                self.array = None
                lib().vo_free_spread(self._p)
                self._p = None
        except Exception:
            pass


class ScalarQuantizer:
    """quantization.rs:191-260 — trained on `train_rows`, then used to encode any rows"""

    def __init__(self, train_rows):
        t = _f(train_rows)
        self.dim = t.shape[1]
        self.min_vals = np.empty(self.dim, np.float32)
        self.scales = np.empty(self.dim, np.float32)
        self.inv_scales = np.empty(self.dim, np.float32)
        lib().vo_sq_train(t, t.shape[0], self.dim, self.min_vals, self.scales, self.inv_scales)

    def quantize(self, rows):
        r = _f(rows).reshape(-1, self.dim)
        out = np.empty(r.shape, dtype=np.uint8)
        lib().vo_sq_quantize(r, r.shape[0], self.dim, self.min_vals, self.scales, out)
        return out


    def dequantize(self, codes):
        """quantization.rs:255-268: code * inv_scale + min"""
        c = np.ascontiguousarray(codes, dtype=np.uint8).reshape(-1, self.dim)
        return c.astype(np.float32) * self.inv_scales[None, :] + self.min_vals[None, :]

    def distance_l2_quantized(self, a, b) -> int:
        """quantization.rs:42-91 (scalar and SIMD forms agree: integer arithmetic): sum of squared code differences"""
        a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
        b = np.ascontiguousarray(b, dtype=np.uint8).reshape(-1)
        assert a.size == self.dim and b.size == self.dim
        return int(lib().vo_sq_l2(a, b, self.dim))


def dual_search_int8(graph, sq, codes, q, k, ef, oversampling=4, tie=TIE_CANONICAL):
    """DualPrecisionHnsw::search_with_config(use_int8_traversal=true) on `graph` (a NativeHnsw) ->
    (nodes, exact engine distances, n_dist_int8, n_expand)"""
    ids = np.empty(max(k, 1), dtype=np.uint64)
    ds = np.empty(max(k, 1), dtype=np.float32)
    a, b = C.c_uint64(0), C.c_uint64(0)
    n = lib().vo_dual_search_int8(graph._h, np.ascontiguousarray(codes, dtype=np.uint8), sq.min_vals, sq.scales, _f(q), k,
                                  ef, oversampling, tie, ids, ds, C.byref(a), C.byref(b))
    return ids[:n].copy(), ds[:n].copy(), int(a.value), int(b.value)


def round_bf16(a):
    """f32 -> bf16 (round to nearest even) -> f32, like half::bf16::from_f32(x).to_f32()"""
    a = _f(a)
    out = np.empty_like(a)
    lib().vo_round_bf16(a.reshape(-1), out.reshape(-1), a.size)
    return out


def scan_topk_bf16(metric, rows, queries, k, nthreads=1):
    """half_precision.rs BF16 semantics: exact top-k over bf16-rounded rows and queries, f32 sequential sums"""
    rows, queries = _f(rows), _f(queries)
    if queries.ndim == 1:
        queries = queries.reshape(1, -1)
    nq = queries.shape[0]
    ids = np.zeros((nq, k), dtype=np.uint64)
    sc = np.zeros((nq, k), dtype=np.float32)
    lib().vo_scan_topk_bf16(metric, rows, rows.shape[0], rows.shape[1], queries, nq, k, nthreads, ids, sc)
    return ids, sc


# ---- NativeHnsw -------------------------------------------------------------
class NativeHnsw:
    """Mirror of NativeHnsw<D> (native/graph.rs)."""

    def __init__(self, dim, metric, M, ef_construction, mode=MODE_R, _handle=None):
        self.dim, self.metric, self.mode = dim, metric, mode
        self._owned = _handle is None
        self._h = _handle if _handle is not None else lib().vo_hnsw_new(dim, metric, mode, M, ef_construction)

    def __del__(self):
        if getattr(self, "_owned", False) and self._h:
            lib().vo_hnsw_free(self._h)
            self._h = None

    def set_alpha(self, a):
        lib().vo_hnsw_set_alpha(self._h, a)

    def insert(self, v):
        v = _f(v)
        assert v.size == self.dim
        return int(lib().vo_hnsw_insert(self._h, v))

    def set_build_tie(self, tie):
        lib().vo_hnsw_set_build_tie(self._h, tie)

    def set_build_threads(self, nthreads):
        """Host threads of the batch-synchronous build's search phase; the graph does not depend on it."""
        lib().vo_hnsw_set_build_threads(self._h, int(nthreads))

    def insert_batch_sync(self, vecs):
        vecs = _f(vecs).reshape(-1, self.dim)
        lib().vo_hnsw_insert_batch_sync(self._h, vecs, vecs.shape[0])

    def build_batched(self, vecs, max_batch):
        vecs = _f(vecs).reshape(-1, self.dim)
        lib().vo_hnsw_build_batched(self._h, vecs, vecs.shape[0], max_batch)

    def __len__(self):
        return int(lib().vo_hnsw_len(self._h))

    @property
    def max_layer(self):
        return int(lib().vo_hnsw_max_layer(self._h))

    @property
    def entry_point(self):
        return int(lib().vo_hnsw_entry_point(self._h))

    @property
    def num_layers(self):
        return int(lib().vo_hnsw_num_layers(self._h))

    def neighbors(self, layer, node):
        buf = np.empty(4096, dtype=np.uint64)
        n = lib().vo_hnsw_neighbors(self._h, layer, node, buf, buf.size)
        return buf[:n].tolist()

    def vector(self, node):
        p = lib().vo_hnsw_vector(self._h, node)
        return np.ctypeslib.as_array(p, shape=(self.dim,)).copy()

    def search(self, q, k, ef, tie=TIE_REFERENCE):
        q = _f(q)
        ids = np.empty(max(k, 1), dtype=np.uint64)
        ds = np.empty(max(k, 1), dtype=np.float32)
        n = lib().vo_hnsw_search(self._h, q, k, ef, tie, ids, ds)
        return ids[:n].copy(), ds[:n].copy()

    def search_multi_entry(self, q, k, ef, num_probes, tie=TIE_REFERENCE):
        """NativeHnsw::search_multi_entry (graph.rs:288-348); advances the graph's level RNG like the reference."""
        q = _f(q)
        ids = np.empty(max(k, 1), dtype=np.uint64)
        ds = np.empty(max(k, 1), dtype=np.float32)
        n = lib().vo_hnsw_search_multi_entry(self._h, q, k, ef, num_probes, tie, ids, ds)
        return ids[:n].copy(), ds[:n].copy()

    def rng_state(self):
        return int(lib().vo_hnsw_rng_state(self._h))

    def spread(self, nthreads):
        """re-place the vectors round-robin over the pool's threads (many-thread baselines on NUMA hosts)"""
        lib().vo_hnsw_spread(self._h, nthreads)

    def search_batch(self, queries, k, ef, tie=TIE_REFERENCE, nthreads=1):
        """-> (nodes [nq,k], dist [nq,k], count [nq], total n_dist, total n_expand)"""
        queries = _f(queries)
        nq = queries.shape[0]
        ids = np.zeros((nq, k), dtype=np.uint64)
        ds = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        a, b = C.c_uint64(0), C.c_uint64(0)
        lib().vo_hnsw_search_batch(self._h, queries, nq, k, ef, tie, nthreads, ids, ds, cnt, C.byref(a), C.byref(b))
        return ids, ds, cnt, int(a.value), int(b.value)

    @staticmethod
    def last_stats():
        a, b = C.c_uint64(0), C.c_uint64(0)
        lib().vo_hnsw_last_stats(C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def search_layer_single(self, q, entry, layer):
        return int(lib().vo_hnsw_search_layer_single(self._h, _f(q), entry, layer))

    def search_layer(self, q, eps, ef, layer, tie=TIE_REFERENCE):
        eps = np.ascontiguousarray(eps, dtype=np.uint64)
        cap = ef + len(eps) + 8
        ids = np.empty(cap, dtype=np.uint64)
        ds = np.empty(cap, dtype=np.float32)
        n = lib().vo_hnsw_search_layer(self._h, _f(q), eps, eps.size, ef, layer, tie, ids, ds, cap)
        return ids[:n].copy(), ds[:n].copy()

    def select_neighbors(self, cand, max_neighbors):
        ids = np.ascontiguousarray([c[0] for c in cand], dtype=np.uint64)
        ds = _f([c[1] for c in cand])
        out = np.empty(max(len(cand), 1), dtype=np.uint64)
        n = lib().vo_hnsw_select_neighbors(self._h, ids, ds, len(cand), max_neighbors, out)
        return out[:n].tolist()

    def file_dump(self, directory, basename):
        rc = lib().vo_hnsw_file_dump(self._h, directory.encode(), basename.encode())
        if rc != 0:
            raise OSError("file_dump failed")

    @classmethod
    def file_load(cls, directory, basename, metric, mode=MODE_R):
        h = lib().vo_hnsw_file_load(directory.encode(), basename.encode(), metric, mode)
        if not h:
            raise OSError("file_load failed")
        obj = cls.__new__(cls)
        obj._h, obj._owned, obj.metric, obj.mode = h, True, metric, mode
        p = lib().vo_hnsw_vector(h, 0)
        obj.dim = None
        return obj


# ---- HnswIndex ---------------------------------------------------------------
class HnswIndex:
    """Mirror of HnswIndex (hnsw/index/*.rs) on the oracle."""

    def __init__(self, dim, metric, mode=MODE_R, M=None, ef_construction=None):
        self.dim, self.metric, self.mode = dim, metric, mode
        if M is None:
            self._h = lib().vo_index_new_auto(dim, metric, mode)
        else:
            self._h = lib().vo_index_new(dim, metric, mode, M, ef_construction)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().vo_index_free(self._h)
            self._h = None

    @property
    def graph(self):
        return NativeHnsw(self.dim, self.metric, 0, 0, self.mode, _handle=lib().vo_index_graph(self._h))

    def insert(self, id_, v):
        v = _f(v)
        assert v.size == self.dim, f"Vector dimension mismatch: expected {self.dim}, got {v.size}"
        return bool(lib().vo_index_insert(self._h, id_, v))

    def remove(self, id_):
        return bool(lib().vo_index_remove(self._h, id_))

    def __len__(self):
        return int(lib().vo_index_len(self._h))

    def search_with_quality(self, q, k, quality=Q_BALANCED, custom_ef=0, tie=TIE_REFERENCE):
        q = _f(q)
        assert q.size == self.dim, f"Query dimension mismatch: expected {self.dim}, got {q.size}"
        ids = np.empty(max(k, 1), dtype=np.uint64)
        sc = np.empty(max(k, 1), dtype=np.float32)
        n = lib().vo_index_search_with_quality(self._h, q, k, quality, custom_ef, tie, ids, sc)
        return ids[:n].copy(), sc[:n].copy()

    def search(self, q, k, tie=TIE_REFERENCE):
        return self.search_with_quality(q, k, Q_BALANCED, 0, tie)

    def search_brute_force(self, q, k):
        q = _f(q)
        ids = np.empty(max(k, 1), dtype=np.uint64)
        sc = np.empty(max(k, 1), dtype=np.float32)
        n = lib().vo_index_search_brute_force(self._h, q, k, ids, sc)
        return ids[:n].copy(), sc[:n].copy()

    def search_with_rerank(self, q, k, rerank_k):
        q = _f(q)
        ids = np.empty(max(k, 1), dtype=np.uint64)
        sc = np.empty(max(k, 1), dtype=np.float32)
        n = lib().vo_index_search_with_rerank(self._h, q, k, rerank_k, ids, sc)
        return ids[:n].copy(), sc[:n].copy()

    def search_with_rerank_quality(self, q, k, rerank_k, quality=Q_ACCURATE, custom_ef=0, tie=TIE_REFERENCE):
        q = _f(q)
        ids = np.empty(max(k, 1), dtype=np.uint64)
        sc = np.empty(max(k, 1), dtype=np.float32)
        n = lib().vo_index_search_with_rerank_quality(self._h, q, k, rerank_k, quality, custom_ef, tie, ids, sc)
        return ids[:n].copy(), sc[:n].copy()

    def search_batch(self, queries, k, quality=Q_BALANCED, custom_ef=0, tie=TIE_REFERENCE, nthreads=1):
        queries = _f(queries)
        nq = queries.shape[0]
        ids = np.zeros((nq, k), dtype=np.uint64)
        sc = np.zeros((nq, k), dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        lib().vo_index_search_batch(self._h, queries, nq, k, quality, custom_ef, tie, nthreads, ids, sc, cnt)
        return ids, sc, cnt


def host_threads() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (cpu.max).  The GPU boxes of
    this project expose 256 hardware threads to a container that is allowed 16 CPUs: 256 threads then time-slice through
    16, and every fork-join of the baseline pays for it."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts and parts[0] != "max":
                    n = min(n, max(1, -(-int(parts[0]) // int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, -(-q // int(f2.read().split()[0]))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def build_info():
    return lib().vo_build_info().decode()


def cpu_has_avx512f():
    return bool(lib().vo_cpu_has_avx512f())


# ---- HnswIndex::save / ::load directory (hnsw/index/constructors.rs:190-287) ---------------------------------
# native_hnsw.{vectors,graph} come from NativeHnsw.file_dump above; the two small files beside them are bincode
# 1.3.3 (Cargo.lock:393-396) with its default options: fixed-width little-endian integers, u64 lengths, usize as u64,
# bool as one byte.  Pure-Python struct code: these files are tiny.
def write_index_meta(directory, dim, metric, enable_vector_storage=True):
    """(usize dimension, u8 metric, bool enable_vector_storage) — constructors.rs:273-283"""
    import struct
    with open(os.path.join(directory, "native_meta.bin"), "wb") as f:
        f.write(struct.pack("<QBB", dim, metric, 1 if enable_vector_storage else 0))


def read_index_meta(directory):
    import struct
    raw = open(os.path.join(directory, "native_meta.bin"), "rb").read()
    dim, metric, storage = struct.unpack("<QBB", raw[:10])
    if metric > 4:
        raise OSError("Unknown distance metric")  # constructors.rs:211-216
    return dim, metric, bool(storage)


def write_index_mappings(directory, idx_to_id, next_idx=None):
    """(HashMap<u64,usize> id_to_idx, HashMap<usize,u64> idx_to_id, usize next_idx) — constructors.rs:262-271.
    `idx_to_id`: dict internal index -> external id of the LIVE entries (removed ids are absent from both maps,
    sharded_mappings.rs:115-122); entries are written in dict order (any order is valid: hash-iteration order)."""
    import struct
    items = list(idx_to_id.items())
    if next_idx is None:
        next_idx = (max(idx_to_id) + 1) if idx_to_id else 0
    with open(os.path.join(directory, "native_mappings.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(items)))
        for idx, id_ in items:
            f.write(struct.pack("<QQ", id_, idx))
        f.write(struct.pack("<Q", len(items)))
        for idx, id_ in items:
            f.write(struct.pack("<QQ", idx, id_))
        f.write(struct.pack("<Q", next_idx))


def read_index_mappings(directory):
    import struct
    raw = open(os.path.join(directory, "native_mappings.bin"), "rb").read()
    off = 0

    def u64():
        nonlocal off
        v = struct.unpack_from("<Q", raw, off)[0]
        off += 8
        return v

    id_to_idx = {}
    for _ in range(u64()):
        id_ = u64()
        id_to_idx[id_] = u64()
    idx_to_id = {}
    for _ in range(u64()):
        idx = u64()
        idx_to_id[idx] = u64()
    next_idx = u64()
    if off != len(raw):
        raise OSError("trailing bytes in native_mappings.bin")
    return id_to_idx, idx_to_id, next_idx


# ---- storage modes (core/quantization.rs) ---------------------------------------------------------------------
class BinaryQuantizedVector:
    """quantization.rs:48-202 — sign bits (x >= 0.0), LSB-first in bytes"""

    def __init__(self, data: np.ndarray, dimension: int):
        self.data, self.dimension = np.ascontiguousarray(data, dtype=np.uint8), int(dimension)

    @classmethod
    def from_f32(cls, vector):
        v = _f(vector).reshape(-1)
        assert v.size > 0, "Cannot quantize empty vector"
        out = np.empty((v.size + 7) // 8, dtype=np.uint8)
        lib().vo_binary_quantize(v, v.size, out)
        return cls(out, v.size)

    def memory_size(self):
        return int(self.data.size)

    def get_bits(self):
        return [bool((self.data[i // 8] >> (i % 8)) & 1) for i in range(self.dimension)]

    def hamming_distance(self, other):
        return int(lib().vo_binary_hamming(self.data, other.data, self.data.size))

    def hamming_similarity(self, other):
        return float(np.float32(1.0) - np.float32(self.hamming_distance(other)) / np.float32(self.dimension))

    def to_bytes(self):  # u32 LE dimension + data (:155-169)
        return int(self.dimension).to_bytes(4, "little") + self.data.tobytes()

    @classmethod
    def from_bytes(cls, raw):
        if len(raw) < 4:
            raise OSError("Not enough bytes for BinaryQuantizedVector header")
        dim = int.from_bytes(raw[:4], "little")
        n = (dim + 7) // 8
        if len(raw) < 4 + n:
            raise OSError("Not enough bytes for BinaryQuantizedVector data")
        return cls(np.frombuffer(raw[4:4 + n], dtype=np.uint8).copy(), dim)


class QuantizedVector:
    """quantization.rs:204-316 — SQ8 with per-vector min / max"""

    def __init__(self, data, mn, mx):
        self.data = np.ascontiguousarray(data, dtype=np.uint8)
        self.min, self.max = np.float32(mn), np.float32(mx)

    @classmethod
    def from_f32(cls, vector):
        v = _f(vector).reshape(-1)
        assert v.size > 0, "Cannot quantize empty vector"
        data = np.empty(v.size, dtype=np.uint8)
        mn, mx = C.c_float(0), C.c_float(0)
        lib().vo_sq8_quantize(v, v.size, data, C.byref(mn), C.byref(mx))
        return cls(data, mn.value, mx.value)

    def dimension(self):
        return int(self.data.size)

    def memory_size(self):
        return int(self.data.size) + 8

    def to_f32(self):
        out = np.empty(self.data.size, dtype=np.float32)
        lib().vo_sq8_dequantize(self.data, self.min, self.max, self.data.size, out)
        return out

    def to_bytes(self):  # min, max f32 LE + data (:289-295)
        return np.float32(self.min).tobytes() + np.float32(self.max).tobytes() + self.data.tobytes()

    @classmethod
    def from_bytes(cls, raw):
        if len(raw) < 8:
            raise OSError("Not enough bytes for QuantizedVector header")
        mn, mx = np.frombuffer(raw[:8], dtype=np.float32)
        return cls(np.frombuffer(raw[8:], dtype=np.uint8).copy(), mn, mx)


def dot_product_quantized(q, qv, simd=False):  # :322-345 / :410-469 (one summation order)
    return float(lib().vo_sq8_dot(_f(q), qv.data, qv.min, qv.max, qv.data.size))


def euclidean_squared_quantized(q, qv, simd=False):  # :349-374 / :473-518
    return float(lib().vo_sq8_l2sq(_f(q), qv.data, qv.min, qv.max, qv.data.size, 1 if simd else 0))


def cosine_similarity_quantized(q, qv, simd=False):  # :380-395 / :524-554
    return float(lib().vo_sq8_cosine(_f(q), qv.data, qv.min, qv.max, qv.data.size, 1 if simd else 0))


def sq8_norm_sq(qv):
    return float(lib().vo_sq8_norm_sq(qv.data, qv.min, qv.max, qv.data.size))


def scan_topk_sq8(metric, rows, queries, k, nthreads=1):
    """exact top-k of f32 queries over the SQ8 codes of `rows` with the *_simd distance functions"""
    rows, queries = _f(rows), _f(queries)
    if queries.ndim == 1:
        queries = queries.reshape(1, -1)
    nq = queries.shape[0]
    ids = np.zeros((nq, k), dtype=np.uint64)
    sc = np.zeros((nq, k), dtype=np.float32)
    lib().vo_scan_topk_sq8(metric, rows, rows.shape[0], rows.shape[1], queries, nq, k, nthreads, ids, sc)
    return ids, sc


def scan_topk_binary(rows, queries, k):
    """exact top-k by Hamming distance between the sign-bit codes"""
    rows, queries = _f(rows), _f(queries)
    if queries.ndim == 1:
        queries = queries.reshape(1, -1)
    nq = queries.shape[0]
    ids = np.zeros((nq, k), dtype=np.uint64)
    sc = np.zeros((nq, k), dtype=np.float32)
    lib().vo_scan_topk_binary(rows, rows.shape[0], rows.shape[1], queries, nq, k, ids, sc)
    return ids, sc


def merge_shard_records(rec, k, higher_is_better):
    """rec: uint32 [S][nq][k][3] wire records of the per-shard top-k lists -> (ids [nq,k], scores [nq,k], counts [nq])"""
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    S, nq = rec.shape[0], rec.shape[1]
    ids = np.zeros((nq, k), dtype=np.uint64)
    sc = np.zeros((nq, k), dtype=np.float32)
    cnt = np.zeros(nq, dtype=np.uint32)
    lib().vo_merge_shard_records(rec.reshape(-1), S, nq, k, 1 if higher_is_better else 0, ids, sc, cnt)
    return ids, sc, cnt


# ---- MmapStorage directory (core/storage/mmap.rs), restated for the upload-source hand-off ----
# Byte layout: read off the code (no reference test pins bytes, only behaviour — storage/tests.rs:18-166 — which the
# tests of this restatement repeat): vectors.dat = raw LE f32 at byte offsets handed out by a monotonic counter,
# pre-sized to 16 MiB and grown by ensure_capacity; vectors.idx = bincode 1.3.3 FxHashMap<u64, usize>
# (u64 count, count x (u64 id, u64 offset), hash-iteration order = unspecified); vectors.wal = append-only op log
# (1 | id | len u32 | bytes for a store, 2 | id for a delete) that MmapStorage::new does not replay.
class MmapVectorStore:
    INITIAL_SIZE = 16 * 1024 * 1024   # mmap.rs:76
    MIN_GROWTH = 64 * 1024 * 1024     # :80
    GROWTH_FACTOR = 2                 # :84

    def __init__(self, path, dimension):
        """MmapStorage::new (mmap.rs:96-160): create / open the three files, load the index if it was flushed."""
        import struct
        self.path, self.dimension = path, dimension
        os.makedirs(path, exist_ok=True)
        self._dat = os.path.join(path, "vectors.dat")
        if not os.path.exists(self._dat) or os.path.getsize(self._dat) == 0:
            with open(self._dat, "wb") as f:
                f.truncate(self.INITIAL_SIZE)
        self._wal = open(os.path.join(path, "vectors.wal"), "ab")
        self.index = {}
        ip = os.path.join(path, "vectors.idx")
        if os.path.exists(ip):
            raw = open(ip, "rb").read()
            if len(raw) < 8:
                raise OSError("invalid data: vectors.idx")
            (n,) = struct.unpack_from("<Q", raw, 0)
            if len(raw) != 8 + 16 * n:
                raise OSError("invalid data: vectors.idx")
            for i in range(n):
                id_, off = struct.unpack_from("<QQ", raw, 8 + 16 * i)
                self.index[id_] = off
        # next_offset = max offset + one vector (:137-143)
        self.next_offset = (max(self.index.values()) + dimension * 4) if self.index else 0

    def _ensure_capacity(self, required):  # :175-221
        cur = os.path.getsize(self._dat)
        if cur < required:
            new_len = max(cur * self.GROWTH_FACTOR, required + self.MIN_GROWTH, cur + self.MIN_GROWTH, required)
            with open(self._dat, "r+b") as f:
                f.truncate(new_len)

    def store(self, id_, vector):  # :402-455
        import struct
        v = np.ascontiguousarray(vector, dtype="<f4")
        if v.size != self.dimension:
            raise OSError(f"Vector dimension mismatch: expected {self.dimension}, got {v.size}")
        b = v.tobytes()
        self._wal.write(b"\x01" + struct.pack("<QI", id_, len(b)) + b)
        if id_ in self.index:
            off = self.index[id_]           # an update rewrites its slot in place
        else:
            off = self.next_offset
            self.next_offset += len(b)
        self._ensure_capacity(off + len(b))
        with open(self._dat, "r+b") as f:
            f.seek(off)
            f.write(b)
        self.index.setdefault(id_, off)

    def delete(self, id_):  # :575-599: WAL record, index entry removed, slot hole-punched (reads back as zeros)
        import struct
        self._wal.write(b"\x02" + struct.pack("<Q", id_))
        off = self.index.pop(id_, None)
        if off is not None:
            with open(self._dat, "r+b") as f:
                f.seek(off)
                f.write(bytes(self.dimension * 4))

    def retrieve(self, id_):  # :554-573
        off = self.index.get(id_)
        if off is None:
            return None
        n = self.dimension * 4
        if off + n > os.path.getsize(self._dat):
            raise OSError("Offset out of bounds")
        with open(self._dat, "rb") as f:
            f.seek(off)
            return np.frombuffer(f.read(n), dtype="<f4").copy()

    def __len__(self):
        return len(self.index)

    def ids(self):
        return list(self.index)

    def flush(self):  # :602-626
        import struct
        self._wal.flush()
        with open(os.path.join(self.path, "vectors.idx"), "wb") as f:
            f.write(struct.pack("<Q", len(self.index)))
            for id_, off in self.index.items():
                f.write(struct.pack("<QQ", id_, off))

    def close(self):
        self._wal.close()


def read_vector_store(directory, dimension):
    """(ids, vectors) of a flushed store in ascending byte offset — the order vdb_hip_index_upload_vector_store uses."""
    st = MmapVectorStore(directory, dimension)
    items = sorted(st.index.items(), key=lambda kv: kv[1])
    ids = np.array([k for k, _ in items], dtype=np.uint64)
    vecs = np.zeros((len(items), dimension), dtype=np.float32)
    for i, (k, _) in enumerate(items):
        vecs[i] = st.retrieve(k)
    st.close()
    return ids, vecs
