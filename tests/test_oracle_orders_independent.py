"""A SECOND, independent restatement of the reference's production dot product — `dot_product_auto` -> `dot_product_wide16`
(core/simd_avx512.rs:87-97,149-205) over `wide` 0.7.33's f32x8 (fused `mul_add` under avx2 + fma, `reduce_add` =
((l0 + l4) + (l2 + l6)) + ((l1 + l5) + (l3 + l7))) — written from the reference's text with EXACT rational arithmetic and
one correct rounding per operation, and compared bit for bit with the oracle's mode R on random (non-integer) data, where the order
of the additions shows.  (Integer-valued inputs, as in the mode-C definition test, cannot tell two orders apart.)  Mode R is what the
GPU's declared orders are held to within the north-star tolerance, and the CPU baseline of bench.py.  CPU only."""
from fractions import Fraction

import numpy as np
import pytest

from oracle import pyoracle as po

F = np.float32


def rnd(x: Fraction) -> np.float32:
    """round a rational to the nearest f32, ties to even (inputs stay far inside the normal range)"""
    if x == 0:
        return F(0.0)
    c = F(float(x))                      # Fraction -> f64 is correctly rounded; f64 -> f32 may double-round: repair below
    best = c
    for cand in (np.nextafter(c, F(-np.inf)), np.nextafter(c, F(np.inf))):
        dc, db = abs(Fraction(float(cand)) - x), abs(Fraction(float(best)) - x)
        if dc < db or (dc == db and (cand.view(np.uint32) & 1) == 0 and (best.view(np.uint32) & 1) == 1):
            best = cand
    return best


def fr(v) -> Fraction:
    return Fraction(float(v))


def add(a, b):
    return rnd(fr(a) + fr(b))


def fma8(a8, b8, acc8):
    return [rnd(fr(x) * fr(y) + fr(c)) for x, y, c in zip(a8, b8, acc8)]


def reduce_add(v):
    return add(add(add(v[0], v[4]), add(v[2], v[6])), add(add(v[1], v[5]), add(v[3], v[7])))


def dot_wide16(a, b):
    n, simd = len(a), len(a) // 32
    s = [[F(0.0)] * 8 for _ in range(4)]
    for i in range(simd):                                            # :161-180 four accumulators, 32 floats per iteration
        for j in range(4):
            o = i * 32 + 8 * j
            s[j] = fma8(a[o:o + 8], b[o:o + 8], s[j])
    c01 = [add(x, y) for x, y in zip(s[0], s[1])]                    # :183-185 (sum0 + sum1) + (sum2 + sum3), then reduce_add
    c23 = [add(x, y) for x, y in zip(s[2], s[3])]
    result = reduce_add([add(x, y) for x, y in zip(c01, c23)])
    pos = simd * 32
    while pos + 8 <= n:                                              # :191-196 chunks of 8: mul_add onto ZERO, reduce, add
        result = add(result, reduce_add(fma8(a[pos:pos + 8], b[pos:pos + 8], [F(0.0)] * 8)))
        pos += 8
    while pos < n:                                                   # :199-202 scalar tail: a rounded product, then a rounded add
        result = add(result, rnd(fr(a[pos]) * fr(b[pos])))
        pos += 1
    return result


@pytest.mark.parametrize("n", [16, 17, 24, 31, 32, 33, 40, 47, 63, 64, 65, 100, 128, 255, 768])
def test_dot_product_wide16_bit_for_bit(n):
    rng = np.random.default_rng(1000 + n)
    for _ in range(3):
        a, b = rng.standard_normal(n).astype(F), rng.standard_normal(n).astype(F)
        exp = dot_wide16(list(a), list(b))
        got = F(po.dot(a, b, po.MODE_R))
        assert got.view(np.uint32) == exp.view(np.uint32), (n, float(got), float(exp))


def test_the_restatement_can_tell_orders_apart():
    """the point of non-integer data: on the same vectors the other declared orders give DIFFERENT bits (so equality above is not
    vacuous), while all stay within the reference's tolerance of each other"""
    rng = np.random.default_rng(7)
    differ = 0
    for _ in range(20):
        a, b = rng.standard_normal(768).astype(F), rng.standard_normal(768).astype(F)
        r, c, s = po.dot(a, b, po.MODE_R), po.dot(a, b, po.MODE_C), po.dot(a, b, po.MODE_SCALAR)
        differ += (r != c) + (r != s)
        assert abs(r - c) <= 1e-4 * max(1.0, abs(r)) and abs(r - s) <= 1e-4 * max(1.0, abs(r))
    assert differ >= 20


def wide16(a, b, terms):
    """the common shape of dot_product_wide16 / squared_l2_wide16 / cosine_similarity_wide16 for one accumulated quantity:
    terms(x, y) -> (u, v) such that the quantity accumulates u * v (a fused multiply-add per lane / a rounded product in the tail)"""
    n, simd = len(a), len(a) // 32
    s = [[F(0.0)] * 8 for _ in range(4)]
    for i in range(simd):
        for j in range(4):
            o = i * 32 + 8 * j
            u, v = zip(*[terms(x, y) for x, y in zip(a[o:o + 8], b[o:o + 8])])
            s[j] = fma8(u, v, s[j])
    c01 = [add(x, y) for x, y in zip(s[0], s[1])]
    c23 = [add(x, y) for x, y in zip(s[2], s[3])]
    result = reduce_add([add(x, y) for x, y in zip(c01, c23)])
    pos = simd * 32
    while pos + 8 <= n:
        u, v = zip(*[terms(x, y) for x, y in zip(a[pos:pos + 8], b[pos:pos + 8])])
        result = add(result, reduce_add(fma8(u, v, [F(0.0)] * 8)))
        pos += 8
    while pos < n:
        u, v = terms(a[pos], b[pos])
        result = add(result, rnd(fr(u) * fr(v)))
        pos += 1
    return result


def sub(x, y):
    return rnd(fr(x) - fr(y))


@pytest.mark.parametrize("n", [16, 23, 32, 41, 64, 100, 768])
def test_squared_l2_and_cosine_wide16_bit_for_bit(n):
    """squared_l2_wide16 (simd_avx512.rs:207-260: a rounded difference per lane, then diff.mul_add(diff, sum)) and
    cosine_similarity_wide16 (:263-352: three quantities in the same shape, f32 sqrt of each squared norm, 0.0 for a zero norm,
    dot / (norm_a * norm_b))"""
    rng = np.random.default_rng(2000 + n)
    for _ in range(2):
        a, b = rng.standard_normal(n).astype(F), rng.standard_normal(n).astype(F)
        la, lb = list(a), list(b)
        l2 = wide16(la, lb, lambda x, y: (sub(x, y), sub(x, y)))
        assert F(po.sql2(a, b, po.MODE_R)).view(np.uint32) == l2.view(np.uint32)
        assert F(po.euclidean(a, b, po.MODE_R)).view(np.uint32) == np.sqrt(l2, dtype=F).view(np.uint32)      # euclidean_auto = sqrt
        dot = wide16(la, lb, lambda x, y: (x, y))
        na, nb = wide16(la, lb, lambda x, y: (x, x)), wide16(la, lb, lambda x, y: (y, y))
        denom = rnd(fr(np.sqrt(na, dtype=F)) * fr(np.sqrt(nb, dtype=F)))
        cos = rnd(fr(dot) / fr(denom))
        assert F(po.cosine(a, b, po.MODE_R)).view(np.uint32) == cos.view(np.uint32), (n, po.cosine(a, b, po.MODE_R), float(cos))


# ---------------------------------------------------------------- the two orders the GPU declares, from DESIGN §2's definitions
def dot_mode_m(a, b):
    """mode M: ONE fused chain per pair over k = 128 U + 16 m + 4 kk + c, loops U, m, c, kk (what a sequence of
    v_mfma_f32_16x16x4_f32 does to one accumulator), vectors zero-padded to a multiple of 128"""
    pad = (-len(a)) % 128
    a, b = list(a) + [F(0.0)] * pad, list(b) + [F(0.0)] * pad
    acc = F(0.0)
    for U in range(len(a) // 128):
        for m in range(8):
            for c in range(4):
                for kk in range(4):
                    k = 128 * U + 16 * m + 4 * kk + c
                    acc = rnd(fr(a[k]) * fr(b[k]) + fr(acc))
    return acc


def dot_mode_c(a, b):
    """mode C: element i -> float4 chunk i / 4 -> lane (i / 4) mod 64; each lane one fused chain from +0 over its elements in
    increasing i; lanes combined by the xor butterfly 32, 16, 8, 4, 2, 1"""
    t = [F(0.0)] * 64
    for i in range(len(a)):
        lane = (i // 4) % 64
        t[lane] = rnd(fr(a[i]) * fr(b[i]) + fr(t[lane]))
    for s in (32, 16, 8, 4, 2, 1):
        t = [add(t[l], t[l ^ s]) for l in range(64)]
    return t[0]


@pytest.mark.parametrize("n", [1, 5, 64, 127, 128, 129, 300, 768, 1000])
def test_declared_gpu_orders_bit_for_bit(n):
    """the oracle's modes C and M — what every GPU kernel is bit-compared with — against these definitions, on random data"""
    rng = np.random.default_rng(3000 + n)
    a, b = rng.standard_normal(n).astype(F), rng.standard_normal(n).astype(F)
    assert F(po.dot(a, b, po.MODE_M)).view(np.uint32) == dot_mode_m(a, b).view(np.uint32)
    assert F(po.dot(a, b, po.MODE_C)).view(np.uint32) == dot_mode_c(list(a), list(b)).view(np.uint32)


@pytest.mark.parametrize("n", [3, 64, 300, 768])
def test_mode_c_cosine_and_euclidean_bit_for_bit(n):
    """DESIGN §2, mode C: cosine = dot / (sqrt(sum a^2) * sqrt(sum b^2)) with every sum in the lane-chain order, 0 if a norm is 0;
    L2 = sqrt of the same chain over rounded differences (diff * diff fused onto the lane's sum)"""
    rng = np.random.default_rng(4000 + n)
    a, b = rng.standard_normal(n).astype(F), rng.standard_normal(n).astype(F)
    la, lb = list(a), list(b)
    dot, na, nb = dot_mode_c(la, lb), dot_mode_c(la, la), dot_mode_c(lb, lb)
    cos = rnd(fr(dot) / fr(rnd(fr(np.sqrt(na, dtype=F)) * fr(np.sqrt(nb, dtype=F)))))
    assert F(po.cosine(a, b, po.MODE_C)).view(np.uint32) == cos.view(np.uint32)
    assert F(po.norm_sq(a, po.MODE_C)).view(np.uint32) == na.view(np.uint32)
    d = [sub(x, y) for x, y in zip(la, lb)]
    l2sq = dot_mode_c(d, d)
    assert F(po.sql2(a, b, po.MODE_C)).view(np.uint32) == l2sq.view(np.uint32)
    assert F(po.euclidean(a, b, po.MODE_C)).view(np.uint32) == np.sqrt(l2sq, dtype=F).view(np.uint32)
    assert po.cosine(np.zeros(n, F), a, po.MODE_C) == 0.0


def dot_simd8(a, b):
    """simd_explicit.rs:50-78 dot_product_simd: ONE f32x8 accumulator (fused), reduce_add, then the remainder as rounded products added
    one by one — what `dot_product_auto` falls back to below 16 elements (simd_avx512.rs:90-96)"""
    s = [F(0.0)] * 8
    simd = len(a) // 8
    for i in range(simd):
        s = fma8(a[8 * i:8 * i + 8], b[8 * i:8 * i + 8], s)
    result = reduce_add(s)
    for i in range(8 * simd, len(a)):
        result = add(result, rnd(fr(a[i]) * fr(b[i])))
    return result


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 15, 24, 100, 768])
def test_eight_lane_form_and_the_small_vector_fallback(n):
    rng = np.random.default_rng(5000 + n)
    a, b = rng.standard_normal(n).astype(F), rng.standard_normal(n).astype(F)
    exp = dot_simd8(list(a), list(b))
    assert F(po.dot_simd8(a, b)).view(np.uint32) == exp.view(np.uint32)
    if n < 16:                                       # below 16 elements the production entry point IS this form
        assert F(po.dot(a, b, po.MODE_R)).view(np.uint32) == exp.view(np.uint32)
