"""Pins the oracle's restatement of core/simd_explicit.rs (the `wide` f32x8 kernels :50-189, threshold Hamming :234-287, packed
popcount :290-366, Jaccard :372-443) against the reference's own tests for that module (simd_explicit_tests.rs, transcribed as data
with their lines).  The three f32 kernels are checked in every arithmetic order the repo declares (the 8-lane form of this module, the
16-lane production form, the GPU's modes C and the scalar loop): the reference's tolerances are what all of them must meet.  CPU only."""
import math

import numpy as np
import pytest

from oracle import pyoracle as po

F = np.float32
EPS = 1e-5   # simd_explicit_tests.rs:7
F32 = {"explicit8": None, "wide16": po.MODE_R, "gpu-C": po.MODE_C, "scalar": po.MODE_SCALAR}


def gen(dim, seed):
    """generate_test_vector (:9-12): sin(seed + 0.1 i) in f32"""
    return np.sin(F(seed) + np.arange(dim, dtype=F) * F(0.1), dtype=F)


def dot(a, b, order):
    return po.dot_simd8(a, b) if F32[order] is None else po.dot(a, b, F32[order])


def euclid(a, b, order):
    return math.sqrt(po.sql2_simd8(a, b)) if F32[order] is None else po.euclidean(a, b, F32[order])


def cos(a, b, order):
    return po.cosine_simd8(a, b) if F32[order] is None else po.cosine(a, b, F32[order])


@pytest.mark.parametrize("order", list(F32))
def test_f32_kernels(order):
    assert abs(dot([1, 2, 3, 4, 5, 6, 7, 8], [1] * 8, order) - 36.0) < EPS                        # :19-24
    a, b = gen(768, 0.0), gen(768, 1.0)                                                             # :27-36
    scalar = F(0.0)
    for x, y in zip(a, b):
        scalar = F(scalar + F(x * y))
    assert abs(dot(a, b, order) - float(scalar)) / max(abs(float(scalar)), 1.0) < 1e-4
    assert abs(euclid(a, a, order)) < EPS                                                           # :39-46
    assert abs(euclid([0.0] * 8, [3.0, 4.0, 0, 0, 0, 0, 0, 0], order) - 5.0) < EPS                  # :49-57
    assert abs(cos(a, a, order) - 1.0) < EPS                                                        # :60-67
    e0, e1 = np.zeros(16, F), np.zeros(16, F)                                                       # :70-80
    e0[0] = e1[1] = 1.0
    assert abs(cos(e0, e1, order)) < EPS
    assert abs(cos(a, -a, order) + 1.0) < EPS                                                       # :83-91
    a5, b5 = [1.0, 2.0, 3.0, 4.0, 5.0], [5.0, 4.0, 3.0, 2.0, 1.0]                                   # :149-156 (not a multiple of 8)
    assert abs(dot(a5, b5, order) - 35.0) < EPS
    assert abs(dot([3.0], [4.0], order) - 12.0) < EPS                                               # :159-164


def test_consistency_between_the_orders():
    """:116-143 — the 8-lane kernels against the `*_fast` entry points (the production 16-lane form): dot and Euclidean within 1e-3,
    cosine within 1e-5; held here between every pair of orders"""
    a, b = gen(768, 0.0), gen(768, 1.0)
    for f, tol in ((dot, 1e-3), (euclid, 1e-3), (cos, 1e-5)):
        vals = [f(a, b, o) for o in F32]
        assert max(vals) - min(vals) < tol, (f.__name__, vals)


def test_dimension_mismatch_panics():
    """:167-172 should_panic "Vector dimensions must match\""""
    with pytest.raises(AssertionError, match="Vector dimensions must match"):
        po.dot([1.0, 2.0, 3.0], [1.0, 2.0])


# ---------------------------------------------------------------- threshold Hamming (f32 in, > 0.5 = set)
ALT, ALT_INV = [1.0, 0.0] * 4, [0.0, 1.0] * 4
H_A, H_B = [1.0, 1.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0], [1.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0, 1.0]


def test_hamming_threshold_form():
    assert po.hamming(ALT, ALT) == 0                      # :178-185, :338-342 (the u32 form returns the same count)
    assert po.hamming(ALT, ALT_INV) == 8                  # :188-193, :345-350
    assert po.hamming(H_A, H_B) == 4                      # :196-202, :353-358: positions 1, 3, 5, 7
    i = np.arange(768)
    a, b = (i % 3 == 0).astype(F), (i % 2 == 0).astype(F)  # :205-222, :361-377
    assert po.hamming(a, b) == int(np.count_nonzero((i % 3 == 0) != (i % 2 == 0))) == 384
    assert po.distance(po.HAMMING, a, b, po.MODE_R) == 384.0


# ---------------------------------------------------------------- packed popcount (u64 words)
def test_hamming_binary_popcount():
    ones = np.full(16, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    assert po.hamming_binary(ones, ones) == 0                                            # :229-233, :252-256
    assert po.hamming_binary(np.zeros(1, np.uint64), ones[:1]) == 64                     # :236-241
    assert po.hamming_binary(np.array([0b10101010], np.uint64), np.array([0b01010101], np.uint64)) == 8   # :244-249
    assert po.hamming_binary(np.zeros(16, np.uint64), ones) == 64 * 16                   # :259-264
    i = np.arange(24, dtype=np.uint64)                                                   # :267-275: the 4-word unrolled form = the plain one
    a, b = i * np.uint64(0x12345678), i * np.uint64(0x87654321)
    assert po.hamming_binary(a, b) == sum(bin(int(x) ^ int(y)).count("1") for x, y in zip(a, b))


# ---------------------------------------------------------------- Jaccard over > 0.5 sets
def test_jaccard_threshold_form():
    assert abs(po.jaccard(ALT, ALT) - 1.0) < EPS                                         # :282-286
    assert abs(po.jaccard([1.0, 0, 1.0, 0, 0, 0, 0, 0], [0, 1.0, 0, 1.0, 0, 0, 0, 0])) < EPS       # :289-294
    assert abs(po.jaccard([1.0, 1.0, 0, 0, 0, 0, 0, 0], [1.0, 0, 1.0, 0, 0, 0, 0, 0]) - 1.0 / 3.0) < EPS   # :297-303
    assert abs(po.jaccard([0.0] * 16, [0.0] * 16) - 1.0) < EPS                           # :306-311: both empty = 1.0
    i = np.arange(768)
    sa, sb = i % 3 == 0, i % 2 == 0                                                      # :314-331
    exact = np.count_nonzero(sa & sb) / np.count_nonzero(sa | sb)
    assert abs(po.jaccard(sa.astype(F), sb.astype(F)) - exact) < 1e-4
    # the count form the GPU's packed-bit path uses (|a & b| / |a | b| in f32) is the same number to the last bit
    assert po.jaccard(sa.astype(F), sb.astype(F)) == float(F(np.count_nonzero(sa & sb)) / F(np.count_nonzero(sa | sb)))


def test_jaccard_fast_aligned_unaligned_and_every_length():
    """simd_tests.rs:326-412 (`jaccard_similarity_fast`, the entry point the engine calls): 768-d patterns stay within [0, 1]; an
    8-aligned and a 67-long vector give 32 / 48 and 30 / 40; the modular patterns at 13 lengths (7 .. 768) equal the counted ratio —
    exactly: numerator and denominator are small integers, the quotient is one f32 division"""
    i = np.arange(768)
    r = po.jaccard((i % 2 == 0).astype(F), (i % 3 == 0).astype(F))              # :326-339
    assert 0.0 <= r <= 1.0 and r == float(F(128) / F(512))
    i = np.arange(64)
    assert abs(po.jaccard((i < 32).astype(F), (i < 48).astype(F)) - 32.0 / 48.0) < EPS    # :342-355
    i = np.arange(67)
    assert abs(po.jaccard((i < 30).astype(F), (i < 40).astype(F)) - 30.0 / 40.0) < EPS    # :358-371 (remainder handling)
    for dim in (7, 8, 15, 16, 31, 32, 63, 64, 127, 128, 255, 256, 768):         # :374-412
        i = np.arange(dim)
        sa, sb = (i * 7) % 11 < 6, (i * 5) % 9 < 5
        inter, union = np.count_nonzero(sa & sb), np.count_nonzero(sa | sb)
        exp = 1.0 if union == 0 else float(F(inter) / F(union))
        assert po.jaccard(sa.astype(F), sb.astype(F)) == exp, dim
        assert po.distance(po.JACCARD, sa.astype(F), sb.astype(F), po.MODE_R) == float(F(1.0) - F(exp))   # engine distance = 1 - similarity

