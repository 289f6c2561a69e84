"""The edge cases the reference's own tests hold for the f32 distance kernels (simd_avx512_tests.rs:276-560; simd_native_tests.rs
:134-275), transcribed as data with their lines and held against EVERY arithmetic order the repo declares: the oracle's restatement of
the production engine (mode R), the intrinsics engine (NATIVE), the scalar loop — and the two orders the GPU kernels are bit-compared
with (modes C and M): zero vectors, opposite signs, values whose squares are denormal or near overflow, 1 to 1 000 000 dimensions,
long sums of ones.  CPU only."""
import math

import numpy as np
import pytest

from oracle import pyoracle as po

F = np.float32
EPS = 1e-5
MODES = {"R": po.MODE_R, "native": po.MODE_NATIVE, "scalar": po.MODE_SCALAR, "gpu-C": po.MODE_C, "gpu-M": po.MODE_M}


def gen(dim, seed):
    return np.sin(F(seed) + np.arange(dim, dtype=F) * F(0.1), dtype=F)


def unit(v):
    n = F(0.0)
    for x in v:                      # `iter().map(|x| x * x).sum::<f32>().sqrt()`: a sequential f32 sum
        n = F(n + F(x * x))
    return (v / np.sqrt(n, dtype=F)).astype(F)


@pytest.mark.parametrize("mode", list(MODES.values()), ids=list(MODES))
def test_zero_vectors(mode):
    z, v = np.zeros(768, F), gen(768, 0.0)
    assert abs(po.dot(z, z, mode)) < EPS                    # simd_avx512_tests.rs:280-285
    assert abs(po.euclidean(z, z, mode)) < EPS              # :288-293
    assert po.cosine(z, z, mode) == 0.0                     # :296-301: defined as 0
    assert po.cosine(v, z, mode) == 0.0                     # :304-309
    assert po.cosine(z, v, mode) == 0.0
    assert po.dot(z, z, mode) == 0.0                        # simd_native_tests.rs:134-139
    assert po.dot(np.ones(768, F), np.ones(768, F), mode) == 768.0   # :142-147


@pytest.mark.parametrize("mode", list(MODES.values()), ids=list(MODES))
def test_signs_and_magnitudes(mode):
    i = np.arange(768, dtype=F)
    a, b = -(i * F(0.01)), i * F(0.01)                      # :317-328
    assert po.dot(a, b, mode) < 0.0 and po.euclidean(a, b, mode) > 0.0 and po.cosine(a, b, mode) < 0.0
    assert abs(po.cosine(a, b, mode) + 1.0) < 1e-4
    tiny = np.full(768, 1e-20, F)                           # :335-353: squares are denormal
    assert math.isfinite(po.dot(tiny, tiny, mode)) and math.isfinite(po.euclidean(tiny, tiny, mode))
    # The reference allows 1 + 1e-5 here ("cosine can slightly exceed 1.0").  Mode M — the GPU's matrix-core order: ONE 768-term chain
    # for the dot product, norms in mode C's order — rounds a sum of EQUAL terms the same way at every step, so a constant
    # vector shows the order at the 1e-6 level (1.0000066 at 1e-10, within the north-star's 1e-5); where the products are SUBNORMAL
    # (1e-40: ~16 significant bits) it reaches 1.0000110, 1e-6 outside the reference's band.  Stated, not hidden: the one place the
    # declared order leaves the reference's tolerance is |x| < 1e-19.
    band = 2e-5 if mode == po.MODE_M else EPS
    assert -1.0 - band <= po.cosine(tiny, tiny, mode) <= 1.0 + band
    # (the scalar engine multiplies the two squared norms before the root, native/distance.rs:171: it leaves f32 at 1e-18 and 1e10)
    for scale in ((1e-3, 1.0, 1e3) if mode == po.MODE_SCALAR else (1e-18, 1e-16, 1e-10, 1e-3, 1.0, 1e3, 1e10)):   # squares normal: within 1e-5
        c = np.full(768, scale, F)
        assert abs(po.cosine(c, c, mode) - 1.0) <= EPS, (scale, po.cosine(c, c, mode))
    large = np.full(32, 1e18, F)                            # :360-372: squares near overflow
    if mode != po.MODE_SCALAR:
        assert abs(po.cosine(large, large, mode) - 1.0) < 1e-4
    else:
        # CpuDistance's cosine takes sqrt(norm_a * norm_b) (native/distance.rs:171): the PRODUCT of the two squared norms overflows
        # here (3.2e37 ^ 2), the quotient is 0 — the reference's test addresses the production engine only; restated as it is
        assert po.cosine(large, large, mode) == 0.0


@pytest.mark.parametrize("mode", list(MODES.values()), ids=list(MODES))
@pytest.mark.parametrize("dim", [384, 1536, 4096])          # simd_native_tests.rs:261-281; simd_avx512_tests.rs:379-391,410-426
def test_large_dimensions(mode, dim):
    a, b = gen(dim, 0.0), gen(dim, 1.0)
    dot, dist, cos = po.dot(a, b, mode), po.euclidean(a, b, mode), po.cosine(a, b, mode)
    assert math.isfinite(dot) and math.isfinite(dist) and dist >= 0.0 and -1.0 <= cos <= 1.0
    ref = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
    assert abs(dot - ref) <= 1e-4 * max(abs(ref), 1.0)      # the tolerance the reference holds its kernels to against a scalar sum


@pytest.mark.parametrize("mode", list(MODES.values()), ids=list(MODES))
def test_million_dimensions_and_long_sums(mode):
    i = np.arange(1_000_000, dtype=F)                       # :394-403
    a, b = np.sin(i * F(0.001), dtype=F), np.cos(i * F(0.002), dtype=F)
    assert math.isfinite(po.dot(a, b, mode))
    ones = np.ones(10000, F)                                # :433-444: |sum - 10000| < 1 (every order here is exact on it)
    assert po.dot(ones, ones, mode) == 10000.0


@pytest.mark.parametrize("mode", list(MODES.values()), ids=list(MODES))
def test_unit_vectors(mode):
    a, b = unit(gen(768, 0.0)), unit(gen(768, 1.0))         # :448-466, :475-523
    assert -1.0 <= po.cosine(a, b, mode) <= 1.0
    # cosine_similarity_normalized = the plain dot product of unit vectors: 1.0 for identical, 0 for orthogonal, = cosine within 1e-4
    assert abs(po.dot(a, a, mode) - 1.0) < EPS
    e0, e1 = np.zeros(768, F), np.zeros(768, F)
    e0[0] = e1[1] = 1.0
    assert abs(po.dot(e0, e1, mode)) < EPS
    assert abs(po.dot(a, b, mode) - po.cosine(a, b, mode)) < 1e-4
    q = unit(gen(768, 100.0))                               # :528-556 batch_cosine_normalized: ten unit vectors, all within [-1, 1]
    for s in range(10):
        assert -1.0 - EPS <= po.dot(unit(gen(768, float(s))), q, mode) <= 1.0 + EPS


@pytest.mark.parametrize("mode", list(MODES.values()), ids=list(MODES))
@pytest.mark.parametrize("n", [0, 1, 7, 8, 15, 16, 17, 31, 32, 33])   # simd_native_tests.rs:150-157,236-258,373-420 (remainders)
def test_remainder_lengths(mode, n):
    rng = np.random.default_rng(n)
    a, b = rng.standard_normal(n).astype(F), rng.standard_normal(n).astype(F)
    ref = float(np.dot(a.astype(np.float64), b.astype(np.float64)))
    assert abs(po.dot(a, b, mode) - ref) <= 1e-5 * max(1.0, abs(ref)) + 1e-6
    ref2 = float(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum())
    assert abs(po.sql2(a, b, mode) - ref2) <= 1e-5 * max(1.0, ref2) + 1e-6
    if n == 0:
        assert po.dot(a, b, mode) == 0.0 and po.euclidean(a, b, mode) == 0.0     # :236-241 empty vectors


def test_declared_orders_against_the_production_order_by_data_family():
    """north_star: "distances within 1e-5 relative for f32 on synthetic random-normal 768-D vectors".  How far the two orders the GPU
    declares (C: 64 short lane chains + a butterfly; M: one 768-term chain, the matrix instruction's order) sit from the restated
    production order (R) on the data families around that sentence — measured here so that the claim has its boundary written down:
    random-normal, ramp and the bench generator's vectors agree to <= 1e-6 (C) / 2e-6 (M); vectors whose components are all EQUAL (or
    take two values) drive a single long chain to round the same way at every step — mode M leaves 1e-5 there (1.04e-5 measured), the
    scalar loop of the reference's own CpuDistance has the same shape, and its tests allow 1e-4 between the two (distance.rs:245-259)."""
    rng = np.random.default_rng(0)
    i = np.arange(768, dtype=F)
    fam = {
        "normal": [(rng.standard_normal(768).astype(F), rng.standard_normal(768).astype(F)) for _ in range(40)],
        "normal-self": [(v, v) for v in (rng.standard_normal(768).astype(F) for _ in range(40))],
        "ramp": [((i + F(s)) * F(0.01), i * F(0.013) + F(s)) for s in range(40)],
        "bench-sin": [(((np.sin(F(0.1 * s) + i * F(0.01)) + 1) / 2).astype(F), ((np.sin(F(0.1 * (s + 7)) + i * F(0.01)) + 1) / 2).astype(F))
                      for s in range(40)],
        "constant": [(np.full(768, a, F), np.full(768, b, F)) for a, b in rng.uniform(1e-3, 10, (40, 2)).astype(F)],
        "two-valued": [((rng.random(768) < 0.5).astype(F) * F(a) + F(b), (rng.random(768) < 0.5).astype(F) * F(a) + F(b))
                       for a, b in rng.uniform(0.01, 3, (40, 2))],
    }
    worst = {}
    for name, pairs in fam.items():
        for mode, tag in ((po.MODE_C, "C"), (po.MODE_M, "M")):
            worst[name, tag] = max(abs(po.cosine(a, b, mode) - po.cosine(a, b, po.MODE_R)) for a, b in pairs)
    for name in ("normal", "normal-self", "ramp", "bench-sin"):
        assert worst[name, "C"] <= 1e-6 and worst[name, "M"] <= 2e-6, (name, worst[name, "C"], worst[name, "M"])
    for name in ("constant", "two-valued"):
        assert worst[name, "C"] <= 1e-6, (name, worst[name, "C"])          # the lane-chain order stays at rounding level
        assert worst[name, "M"] <= 2e-5, (name, worst[name, "M"])          # the one-chain order: up to ~1e-5, written down
    assert worst["normal", "M"] <= 1e-7                                    # the north-star's own data class: 4e-8
