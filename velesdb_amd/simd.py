"""Host-side mirror of the reference's free SIMD functions that sit on the hot path,
in their natural GPU form: one call for n vectors.  Every function runs in libvelesdb_hip.so (vec_utils.hip,
score_rows); there is no CPU path."""
from __future__ import annotations

import numpy as np

from ._ffi import check, lib
from .index import KIND_RAW, _f32, _ptr
from .params import DistanceMetric

KIND_SQUARED = 2


def batch_norm(vectors, device: int = 0) -> np.ndarray:
    """simd::norm / norm_simd (simd.rs:240-242) of every row."""
    v = np.ascontiguousarray(_f32(vectors))
    v = v.reshape(1, -1) if v.ndim == 1 else v
    out = np.empty(v.shape[0], dtype=np.float32)
    check(lib().vdb_hip_batch_norm(device, _ptr(v), v.shape[0], v.shape[1], _ptr(out)))
    return out


def normalize_rows(vectors, device: int = 0) -> np.ndarray:
    """simd::normalize_inplace (simd.rs:217-219) applied to every row; returns the normalised copy."""
    v = np.array(_f32(vectors), dtype=np.float32, copy=True, order="C")
    v2 = v.reshape(1, -1) if v.ndim == 1 else v
    check(lib().vdb_hip_normalize_rows(device, _ptr(v2), v2.shape[0], v2.shape[1]))
    return v


def batch_squared_l2(query, vectors, device: int = 0) -> np.ndarray:
    """simd::squared_l2_distance (simd.rs:207-211) of one query against n rows."""
    q, v = _f32(query).reshape(-1), np.ascontiguousarray(_f32(vectors))
    out = np.empty(v.shape[0], dtype=np.float32)
    check(lib().vdb_hip_batch_distance(device, int(DistanceMetric.Euclidean), KIND_SQUARED, _ptr(q), _ptr(v), v.shape[0],
                                       v.shape[1], _ptr(out)))
    return out


def batch_cosine_normalized(candidates, query, device: int = 0) -> np.ndarray:
    """simd::batch_cosine_normalized (simd_avx512.rs:405-422): pre-normalised vectors, i.e. a dot product."""
    q, v = _f32(query).reshape(-1), np.ascontiguousarray(_f32(candidates))
    out = np.empty(v.shape[0], dtype=np.float32)
    check(lib().vdb_hip_batch_distance(device, int(DistanceMetric.DotProduct), KIND_RAW, _ptr(q), _ptr(v), v.shape[0],
                                       v.shape[1], _ptr(out)))
    return out


def batch_dot_product(queries, vectors, device: int = 0) -> np.ndarray:
    """simd_explicit::batch_dot_product (simd_explicit.rs:519-560): [len(queries), len(vectors)] matrix."""
    q, v = np.ascontiguousarray(_f32(queries)), np.ascontiguousarray(_f32(vectors))
    if q.shape[0] == 0:
        return np.empty((0, 0), dtype=np.float32)
    if v.shape[0] == 0:
        return np.empty((q.shape[0], 0), dtype=np.float32)
    assert q.shape[1] == v.shape[1], f"Vector 0 dimension mismatch: expected {q.shape[1]}, got {v.shape[1]}"
    out = np.empty((q.shape[0], v.shape[0]), dtype=np.float32)
    check(lib().vdb_hip_batch_dot_product(device, _ptr(q), q.shape[0], _ptr(v), v.shape[0], v.shape[1], _ptr(out)))
    return out


def batch_hamming_binary(query_words, rows_words, device: int = 0) -> np.ndarray:
    """hamming_distance_binary(_fast) (simd_explicit.rs:308-360) of one packed-u64 query against n packed rows."""
    q = np.ascontiguousarray(query_words, dtype=np.uint64).reshape(-1)
    r = np.ascontiguousarray(rows_words, dtype=np.uint64).reshape(-1, q.size)
    out = np.empty(r.shape[0], dtype=np.uint32)
    check(lib().vdb_hip_batch_hamming_binary(device, _ptr(q), _ptr(r), r.shape[0], q.size, _ptr(out)))
    return out


def batch_jaccard_binary(query_words, rows_words, device: int = 0) -> np.ndarray:
    """jaccard_similarity_binary (simd_explicit.rs:457-500)."""
    q = np.ascontiguousarray(query_words, dtype=np.uint64).reshape(-1)
    r = np.ascontiguousarray(rows_words, dtype=np.uint64).reshape(-1, q.size)
    out = np.empty(r.shape[0], dtype=np.float32)
    check(lib().vdb_hip_batch_jaccard_binary(device, _ptr(q), _ptr(r), r.shape[0], q.size, _ptr(out)))
    return out
