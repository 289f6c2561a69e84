"""Storage modes of core/quantization.rs on the oracle, pinned by the reference's own test literals
(quantization_tests.rs): SQ8 (per-vector min/max) and binary (sign bits) codes, their byte formats, the asymmetric
f32 x SQ8 distances (scalar and `_simd` flavours).  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po

EPS = float(np.finfo(np.float32).eps)


# ---- QuantizedVector (quantization_tests.rs:69-190) ----
def test_quantize_simple_negative_constant():
    q = po.QuantizedVector.from_f32([0.0, 0.5, 1.0])
    assert q.dimension() == 3 and abs(q.min) < EPS and abs(q.max - 1.0) < EPS
    assert q.data.tolist() == [0, 128, 255]                       # :81-83
    q = po.QuantizedVector.from_f32([-1.0, 0.0, 1.0])
    assert abs(q.min + 1.0) < EPS and abs(q.max - 1.0) < EPS and q.data.tolist() == [0, 128, 255]   # :95-99
    q = po.QuantizedVector.from_f32([0.5, 0.5, 0.5])
    assert q.data.tolist() == [128, 128, 128]                     # :111-115
    assert np.array_equal(q.to_f32(), np.float32([0.5, 0.5, 0.5]))  # to_f32 of a constant vector = min


def test_dequantize_roundtrip_and_sizes():
    orig = np.float32([0.1, 0.5, 0.9, -0.3, 0.0])
    q = po.QuantizedVector.from_f32(orig)
    rec = q.to_f32()
    step = (float(q.max) - float(q.min)) / 255.0
    assert np.all(np.abs(rec - orig) <= step)                     # :129-137: within one quantisation step
    v768 = np.arange(768, dtype=np.float32) / 768
    assert po.QuantizedVector.from_f32(v768).memory_size() == 776  # :156-157
    b = q.to_bytes()
    assert len(b) == 8 + 5 and b[:4] == np.float32(q.min).tobytes() and b[4:8] == np.float32(q.max).tobytes()
    d = po.QuantizedVector.from_bytes(b)
    assert d.min == q.min and d.max == q.max and np.array_equal(d.data, q.data)   # :165-177
    with pytest.raises(OSError):
        po.QuantizedVector.from_bytes(bytes(4))                   # :181-189


# ---- asymmetric distances (quantization_tests.rs:9-66,196-289) ----
def test_sq8_distances_reference_cases():
    for simd in (False, True):
        q1 = po.QuantizedVector.from_f32([1.0, 0.0, 0.0])
        assert abs(po.dot_product_quantized([1.0, 0.0, 0.0], q1, simd) - 1.0) < 0.1
        assert abs(po.dot_product_quantized([1.0, 0.0, 0.0], po.QuantizedVector.from_f32([0.0, 1.0, 0.0]), simd)) < 0.1
        assert abs(po.euclidean_squared_quantized([0.0, 0.0, 0.0], q1, simd) - 1.0) < 0.1
        c = [0.5, 0.5, 0.5]
        assert po.euclidean_squared_quantized(c, po.QuantizedVector.from_f32(c), simd) < 0.01
        assert abs(po.cosine_similarity_quantized([1.0, 2.0, 3.0], po.QuantizedVector.from_f32([1.0, 2.0, 3.0]), simd) - 1.0) < 0.05
        assert abs(po.cosine_similarity_quantized([1.0, 0.0, 0.0], po.QuantizedVector.from_f32([-1.0, 0.0, 0.0]), simd) + 1.0) < 0.1
    v8 = [1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0]
    assert abs(po.cosine_similarity_quantized(v8, po.QuantizedVector.from_f32(v8), True) - 1.0) < 0.05   # :54-63


def test_sq8_simd_vs_scalar_768d():
    # quantization_tests.rs:23-51: same generators, relative error < 0.01 between the two flavours
    query = (np.arange(768, dtype=np.float32) / np.float32(1000.0))
    v1 = po.QuantizedVector.from_f32(query)
    v2 = po.QuantizedVector.from_f32((np.arange(768, dtype=np.float32) + 10) / np.float32(1000.0))
    s, v = po.dot_product_quantized(query, v1), po.dot_product_quantized(query, v1, True)
    assert s == v                                              # identical summation order
    s, v = po.euclidean_squared_quantized(query, v2), po.euclidean_squared_quantized(query, v2, True)
    assert abs((s - v) / s) < 0.01
    s, v = po.cosine_similarity_quantized(query, v1), po.cosine_similarity_quantized(query, v1, True)
    assert abs(s - v) < 1e-5


def test_sq8_restatement_matches_a_direct_numpy_f32_evaluation():
    # independent re-derivation of the `_simd` formulas with numpy float32 scalars (one rounding per operation)
    rng = np.random.default_rng(5)
    for dim in (3, 8, 17, 100):
        v = rng.standard_normal(dim).astype(np.float32)
        q = rng.standard_normal(dim).astype(np.float32)
        qv = po.QuantizedVector.from_f32(v)
        scale = np.float32(qv.max - qv.min) / np.float32(255.0)
        deq = [np.float32(np.float32(int(c)) * scale) + qv.min for c in qv.data]
        dot = np.float32(0)
        for i in range(dim):
            dot = np.float32(dot + np.float32(q[i] * deq[i]))
        assert np.float32(po.dot_product_quantized(q, qv, True)) == dot
        l2 = np.float32(0)
        for c in range(dim // 4):
            f = [np.float32(q[4 * c + j] - deq[4 * c + j]) for j in range(4)]
            t = np.float32(np.float32(np.float32(f[0] * f[0]) + np.float32(f[1] * f[1])) + np.float32(f[2] * f[2]))
            t = np.float32(t + np.float32(f[3] * f[3]))
            l2 = np.float32(l2 + t)
        for i in range(dim // 4 * 4, dim):
            f = np.float32(q[i] - deq[i])
            l2 = np.float32(l2 + np.float32(f * f))
        assert np.float32(po.euclidean_squared_quantized(q, qv, True)) == l2
        qs = np.float32(0)
        vs = np.float32(0)
        for i in range(dim):
            qs = np.float32(qs + np.float32(q[i] * q[i]))
            vs = np.float32(vs + np.float32(deq[i] * deq[i]))
        cos = np.float32(dot / np.sqrt(np.float32(qs * vs)))
        assert np.float32(po.cosine_similarity_quantized(q, qv, True)) == cos
        assert np.float32(po.sq8_norm_sq(qv)) == vs


def test_sq8_constant_vector_branches():
    # range < EPSILON: dot = sum(q) * value, l2 = sum((q - value)^2), cosine_simd uses scale 0 (quantization.rs:329-334,
    # 356-361,531-537)
    q = np.float32([0.25, -1.5, 2.0, 0.125, 3.0])
    qv = po.QuantizedVector.from_f32([0.75] * 5)
    qsum = np.float32(0)
    for x in q:
        qsum = np.float32(qsum + x)
    assert po.dot_product_quantized(q, qv, True) == float(np.float32(qsum * np.float32(0.75)))
    assert po.euclidean_squared_quantized(q, qv, True) == po.euclidean_squared_quantized(q, qv, False)
    assert abs(po.cosine_similarity_quantized(q, qv, True) - po.cosine_similarity_quantized(q, qv, False)) < 1e-6
    z = po.QuantizedVector.from_f32([0.0] * 5)
    assert po.cosine_similarity_quantized(q, z, True) == 0.0 and po.cosine_similarity_quantized(q, z, False) == 0.0


# ---- BinaryQuantizedVector (quantization_tests.rs:378-525) ----
def test_binary_codes_reference_cases():
    b = po.BinaryQuantizedVector.from_f32([-1.0, 0.5, -0.5, 1.0])
    assert b.dimension == 4 and b.data.tolist() == [0b1010]      # bit i = vec[i] >= 0 (:379-393)
    v = [0.5 if i % 2 == 0 else -0.5 for i in range(768)]
    b = po.BinaryQuantizedVector.from_f32(v)
    assert b.dimension == 768 and b.data.size == 96 and b.memory_size() == 96 and set(b.data.tolist()) == {0x55}
    bits = po.BinaryQuantizedVector.from_f32([0.0, 0.001, -0.001, EPS]).get_bits()
    assert bits == [True, True, False, True]                      # :422-436 (>= 0.0; -0.0 >= 0.0 is true too)
    assert po.BinaryQuantizedVector.from_f32([-0.0]).get_bits() == [True]
    assert po.BinaryQuantizedVector.from_f32([float("nan")]).get_bits() == [False]
    alt = po.BinaryQuantizedVector.from_f32([0.5, -0.5] * 4)
    assert alt.hamming_distance(alt) == 0                         # :439-449
    ones, neg = po.BinaryQuantizedVector.from_f32([1.0] * 8), po.BinaryQuantizedVector.from_f32([-1.0] * 8)
    assert ones.hamming_distance(neg) == 8                        # :452-464
    b1 = po.BinaryQuantizedVector.from_f32([1.0, 1.0, 1.0, 1.0, -1.0, -1.0, -1.0, -1.0])
    b2 = po.BinaryQuantizedVector.from_f32([1.0, 1.0, -1.0, -1.0, 1.0, 1.0, -1.0, -1.0])
    assert b1.hamming_distance(b2) == 4 and b1.hamming_similarity(b2) == 0.5   # :467-479, :142-145


def test_binary_serialization():
    v = [0.5 if i % 3 == 0 else -0.5 for i in range(768)]
    b = po.BinaryQuantizedVector.from_f32(v)
    raw = b.to_bytes()
    assert raw[:4] == (768).to_bytes(4, "little") and len(raw) == 4 + 96
    d = po.BinaryQuantizedVector.from_bytes(raw)
    assert d.dimension == 768 and np.array_equal(d.data, b.data) and d.hamming_distance(b) == 0   # :482-497
    with pytest.raises(OSError):
        po.BinaryQuantizedVector.from_bytes(bytes(3))             # :500-509
    assert po.BinaryQuantizedVector.from_bytes(po.BinaryQuantizedVector.from_f32([0.5, -0.5] * 512).to_bytes()).dimension == 1024
    with pytest.raises(OSError):
        po.BinaryQuantizedVector.from_bytes((16).to_bytes(4, "little") + bytes(1))   # data shorter than ceil(dim/8)


def test_scan_topk_storage_modes_small():
    rng = np.random.default_rng(8)
    rows = rng.standard_normal((200, 24)).astype(np.float32)
    Q = rng.standard_normal((3, 24)).astype(np.float32)
    codes = [po.QuantizedVector.from_f32(r) for r in rows]
    for metric, fn, best in ((po.COSINE, po.cosine_similarity_quantized, max), (po.DOT, po.dot_product_quantized, max),
                             (po.EUCLIDEAN, po.euclidean_squared_quantized, min)):
        ids, sc = po.scan_topk_sq8(metric, rows, Q, 5)
        for qi in range(3):
            all_s = np.float32([fn(Q[qi], c, True) for c in codes])
            order = np.lexsort((np.arange(200), -all_s if best is max else all_s))[:5]
            assert ids[qi].tolist() == order.tolist() and np.array_equal(sc[qi], all_s[order])
    ids, sc = po.scan_topk_binary(rows, Q, 7)
    bc = [po.BinaryQuantizedVector.from_f32(r) for r in rows]
    for qi in range(3):
        qb = po.BinaryQuantizedVector.from_f32(Q[qi])
        d = np.array([qb.hamming_distance(c) for c in bc])
        order = np.lexsort((np.arange(200), d))[:7]
        assert ids[qi].tolist() == order.tolist() and sc[qi].tolist() == d[order].astype(np.float32).tolist()


def test_recall_accuracy_high_dimension():
    """quantization_tests.rs:296-355 — 100 x 768 vectors ((7 i + 13 j) mod 1000) / 1000 * 2 - 1, query = vector 0, dot product: the
    top-10 over the SQ8 codes overlaps the f32 top-10 in >= 8 ids (here through the scan the GPU's SQ8 sweep is bit-compared with)"""
    i = np.arange(100, dtype=np.int64)[:, None]
    j = np.arange(768, dtype=np.int64)[None, :]
    vectors = (((i * 7 + j * 13) % 1000).astype(np.float32) / np.float32(1000.0) * np.float32(2.0) - np.float32(1.0)).astype(np.float32)
    q = vectors[0]
    f32_scores = np.array([po.dot(q, v, po.MODE_SCALAR) for v in vectors], dtype=np.float64)
    f32_top = set(np.argsort(-f32_scores, kind="stable")[:10].tolist())
    ids, sc = po.scan_topk_sq8(po.DOT, vectors, q, 10)
    recall = len(f32_top & set(ids[0].tolist())) / 10.0
    assert recall >= 0.8, recall
    assert np.all(np.diff(sc[0]) <= 0) and ids[0, 0] == 0     # the query's own code scores highest
    # the per-pair function the reference's test calls (dot_product_quantized, quantization.rs:322-345) gives the same ranking
    pair = np.float32([po.dot_product_quantized(q, po.QuantizedVector.from_f32(v)) for v in vectors])
    assert len(f32_top & set(np.argsort(-pair.astype(np.float64), kind="stable")[:10].tolist())) >= 8
