"""GPU parity for the free SIMD functions of SURVEY 8a rows a16 / a18 (vec_utils.hip): norm, normalize, squared L2,
normalised cosine, dot matrix — bit-exact against the oracle's canonical mode C, within the reference's own 1e-5 of
its scalar definitions — and the packed-u64 Hamming / Jaccard, integer-exact, with the reference's KATs
(simd_explicit_tests.rs:94-108,229-274)."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
from velesdb_amd import simd as vs  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("dim", [1, 3, 8, 17, 64, 100, 768, 1001])
def test_norm_normalize_squared_l2(gpu_required, dim):
    rng = np.random.default_rng(dim)
    rows = rng.standard_normal((50, dim)).astype(np.float32)
    rows[7] = 0.0
    q = rng.standard_normal(dim).astype(np.float32)
    n = vs.batch_norm(rows)
    exp = np.float32([np.sqrt(np.float32(po.norm_sq(r, po.MODE_C))) for r in rows])
    assert np.array_equal(bits(n), bits(exp))
    assert np.allclose(n, [po.norm(r) for r in rows], rtol=1e-5, atol=0)      # simd::norm, the scalar definition
    u = vs.normalize_rows(rows)
    for i in range(50):
        if n[i] == 0.0:
            assert np.array_equal(u[i], rows[i])                                  # zero vector unchanged
        else:
            inv = np.float32(1.0) / n[i]
            assert np.array_equal(bits(u[i]), bits(rows[i] * inv))
    assert np.allclose(vs.batch_norm(u)[n > 0], 1.0, atol=1e-5)                  # simd_explicit_tests.rs:94-102
    d2 = vs.batch_squared_l2(q, rows)
    assert np.array_equal(bits(d2), bits(np.float32([po.sql2(q, r, po.MODE_C) for r in rows])))
    assert np.allclose(np.sqrt(d2), va.HipDistance(va.DistanceMetric.Euclidean).batch_distance(q, rows), rtol=1e-6)
    cn = vs.batch_cosine_normalized(u, q)
    assert np.array_equal(bits(cn), bits(np.float32([po.dot(q, r, po.MODE_C) for r in u])))


def test_reference_normalize_kats(gpu_required):
    u = vs.normalize_rows(np.float32([[3.0, 4.0, 0, 0, 0, 0, 0, 0]]))
    assert abs(u[0, 0] - 0.6) < 1e-5 and abs(u[0, 1] - 0.8) < 1e-5            # simd_explicit_tests.rs:94-102
    z = vs.normalize_rows(np.zeros((1, 16), np.float32))
    assert np.all(z == 0.0)                                                      # :105-109


def test_batch_dot_product_matrix(gpu_required):
    rng = np.random.default_rng(2)
    Q = rng.standard_normal((7, 96)).astype(np.float32)
    V = rng.standard_normal((33, 96)).astype(np.float32)
    m = vs.batch_dot_product(Q, V)
    exp = np.float32([[po.dot(q, v, po.MODE_C) for v in V] for q in Q])
    assert m.shape == (7, 33) and np.array_equal(bits(m), bits(exp))
    assert np.allclose(m, Q.astype(np.float64) @ V.astype(np.float64).T, rtol=1e-4, atol=1e-4)
    assert vs.batch_dot_product(np.empty((0, 96), np.float32), V).shape == (0, 0)        # simd_explicit.rs:521-526
    assert vs.batch_dot_product(Q, np.empty((0, 96), np.float32)).shape == (7, 0)


def test_packed_binary_reference_kats_and_random(gpu_required):
    full = np.full(16, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    assert vs.batch_hamming_binary(full, full[None, :])[0] == 0                   # :229-233, :252-256
    assert vs.batch_hamming_binary(np.zeros(1, np.uint64), np.uint64([[0xFFFFFFFFFFFFFFFF]]))[0] == 64   # :236-241
    assert vs.batch_hamming_binary(np.uint64([0b10101010]), np.uint64([[0b01010101]]))[0] == 8           # :244-249
    assert vs.batch_hamming_binary(np.zeros(16, np.uint64), full[None, :])[0] == 64 * 16                 # :259-264
    a = (np.arange(24, dtype=np.uint64) * np.uint64(0x12345678))
    b = (np.arange(24, dtype=np.uint64) * np.uint64(0x87654321))
    assert vs.batch_hamming_binary(a, b[None, :])[0] == po.hamming_binary(a, b)   # :267-275
    rng = np.random.default_rng(4)
    q = rng.integers(0, 2**63, size=12, dtype=np.uint64)
    rows = rng.integers(0, 2**63, size=(500, 12), dtype=np.uint64)
    rows[3] = 0
    h = vs.batch_hamming_binary(q, rows)
    assert h.tolist() == [po.hamming_binary(q, r) for r in rows]
    j = vs.batch_jaccard_binary(q, rows)
    exp = []
    for r in rows:
        inter = int(sum(bin(int(x) & int(y)).count("1") for x, y in zip(q, r)))
        uni = int(sum(bin(int(x) | int(y)).count("1") for x, y in zip(q, r)))
        exp.append(np.float32(1.0) if uni == 0 else np.float32(inter) / np.float32(uni))
    assert np.array_equal(bits(j), bits(np.float32(exp)))
    assert vs.batch_jaccard_binary(np.zeros(2, np.uint64), np.zeros((1, 2), np.uint64))[0] == 1.0   # J(empty, empty) = 1
