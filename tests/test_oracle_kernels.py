"""Pins the CPU oracle's distance kernels against the reference's own known-answer tests
(tests/golden/reference_kats.json) and tolerance tests (SIMD vs naive scalar on the
reference's closed-form sin vectors, SIMD-boundary sizes).  CPU only."""
import json
import math
import os

import numpy as np
import pytest

from oracle import pyoracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))
F32_MODES = [po.MODE_R, po.MODE_C, po.MODE_NATIVE, po.MODE_R_NOFMA, po.MODE_SCALAR]
QUAL = {"fast": po.Q_FAST, "balanced": po.Q_BALANCED, "accurate": po.Q_ACCURATE, "perfect": po.Q_PERFECT}


def gen_test_vector(dim, seed):
    # simd_tests.rs:17-19 — (seed + i as f32 * 0.1).sin() in f32
    i = np.arange(dim, dtype=np.float32)
    return np.sin((np.float32(seed) + i * np.float32(0.1)).astype(np.float32)).astype(np.float32)


def naive_dot(a, b):
    s = np.float32(0)
    for x, y in zip(a, b):
        s = np.float32(s + np.float32(x * y))
    return float(s)


@pytest.mark.parametrize("kat", GOLD["kats"], ids=lambda k: f"{k['fn']}@{k['src'].split('/')[-1]}")
def test_reference_kat(kat):
    fn, args, exp, tol = kat["fn"], kat["args"], kat["expect"], kat["tol"]
    if fn in ("cosine", "euclidean", "dot", "sql2"):
        for mode in F32_MODES:
            got = getattr(po, fn)(args[0], args[1], mode)
            assert abs(got - exp) <= tol, (mode, got, exp)
    elif fn == "norm":
        assert abs(po.norm(args[0]) - exp) <= tol
    elif fn == "hamming":
        assert po.hamming(args[0], args[1]) == exp
        # the packed-bit re-encoding the HIP path uses is an exact restatement
        a = np.asarray(args[0], dtype=np.float32) > 0.5
        b = np.asarray(args[1], dtype=np.float32) > 0.5
        assert int(np.count_nonzero(a != b)) == exp
    elif fn == "jaccard":
        assert abs(po.jaccard(args[0], args[1]) - exp) <= tol
    elif fn == "engine_distance":
        mode = po.MODE_SCALAR if kat["engine"] == "scalar" else po.MODE_R
        got = po.distance(po.METRICS[args[0]], args[1], args[2], mode)
        assert abs(got - exp) <= tol
    elif fn == "transform_score":
        assert abs(po.transform_score(po.METRICS[args[0]], args[1]) - exp) <= tol
    elif fn == "ef_search":
        q = args[0]
        if q.startswith("custom:"):
            assert po.ef_search(po.Q_CUSTOM, args[1], int(q.split(":")[1])) == exp
        else:
            assert po.ef_search(QUAL[q], args[1]) == exp
    else:
        raise AssertionError(fn)


@pytest.mark.parametrize("mode", F32_MODES)
@pytest.mark.parametrize("size", GOLD["datasets"]["boundary_sizes"]["sizes"] + [768, 1536, 4096])
def test_simd_boundary_sizes_vs_scalar(mode, size):
    # simd_avx512_tests.rs:225-277, simd_tests.rs:85-125: rel < 1e-4 vs naive scalar
    a, b = gen_test_vector(size, 0.0), gen_test_vector(size, 1.0)
    sc = naive_dot(a, b)
    got = po.dot(a, b, mode)
    assert abs(got - sc) / max(abs(sc), 1.0) < 1e-4
    d = (a - b).astype(np.float32)
    sc2 = naive_dot(d, d)
    got2 = po.sql2(a, b, mode)
    assert abs(got2 - sc2) / max(abs(sc2), 1.0) < 1e-4
    # auto vs explicit cosine (abs < 1e-4)
    assert abs(po.cosine(a, b, mode) - po.cosine_simd8(a, b)) < 1e-4


@pytest.mark.parametrize("mode", F32_MODES)
def test_768d_consistency_with_baseline(mode):
    # simd_tests.rs:164-181 — fused cosine vs 3-pass baseline, abs < 1e-5
    a, b = gen_test_vector(768, 0.0), gen_test_vector(768, 1.0)
    base = naive_dot(a, b) / (math.sqrt(naive_dot(a, a)) * math.sqrt(naive_dot(b, b)))
    assert abs(po.cosine(a, b, mode) - base) < 1e-5
    # simd_tests.rs:85-103 — euclidean vs naive, abs < 1e-5
    d = (a - b).astype(np.float32)
    assert abs(po.euclidean(a, b, mode) - math.sqrt(naive_dot(d, d))) < 1e-5
    # simd_avx512_tests.rs:67-78,109-121 — auto vs explicit
    assert abs(po.dot(a, b, mode) - po.dot_simd8(a, b)) < 1e-3
    assert abs(po.sql2(a, b, mode) - po.sql2_simd8(a, b)) < 1e-2


def test_simd_matches_scalar_engine():
    # native/distance.rs:245-259 — |cpu - simd| < 1e-4 on 768-D
    i = np.arange(768, dtype=np.float32)
    a = np.sin(i * np.float32(0.01)).astype(np.float32)
    b = np.cos(i * np.float32(0.02)).astype(np.float32)
    cpu = po.distance(po.COSINE, a, b, po.MODE_SCALAR)
    for mode in (po.MODE_R, po.MODE_C, po.MODE_NATIVE):
        assert abs(cpu - po.distance(po.COSINE, a, b, mode)) < 1e-4


def test_mode_c_within_north_star_tolerance_of_mode_r():
    # north-star: f32 distances within 1e-5 relative.  iid N(0,1) 768-D, cosine/euclid/dot.
    rng = np.random.default_rng(7)
    X = rng.standard_normal((64, 768)).astype(np.float32)
    q = rng.standard_normal(768).astype(np.float32)
    for m in (po.EUCLIDEAN,):
        r = po.batch_compute_distance(m, q, X, po.MODE_R)
        c = po.batch_compute_distance(m, q, X, po.MODE_C)
        assert np.max(np.abs(r - c) / np.abs(r)) < 1e-5
    # cosine / dot of random vectors are near zero: the reference's own tests use an absolute
    # bound (1e-5) there; check both abs and rel-to-norm-product
    r = po.batch_compute_distance(po.COSINE, q, X, po.MODE_R)
    c = po.batch_compute_distance(po.COSINE, q, X, po.MODE_C)
    assert np.max(np.abs(r - c)) < 1e-6
    r = po.batch_compute_distance(po.DOT, q, X, po.MODE_R)
    c = po.batch_compute_distance(po.DOT, q, X, po.MODE_C)
    scale = np.linalg.norm(q) * np.linalg.norm(X, axis=1)
    assert np.max(np.abs(r - c) / scale) < 1e-6


def test_mode_c_definition_independent_python():
    # independent restatement of the canonical order in numpy: lane = (i//4) % 64, fmaf chain,
    # butterfly 32..1.  fmaf emulated exactly through float64 (24x24-bit product is exact in f64,
    # and the f64 sum of an exact product and an f32 is correctly rounded to f32 except in rare
    # double-rounding cases, which the integer-valued inputs below cannot trigger).
    rng = np.random.default_rng(3)
    for n in (1, 3, 4, 5, 63, 64, 255, 256, 257, 300, 768, 1000):
        a = rng.integers(-8, 9, n).astype(np.float32)
        b = rng.integers(-8, 9, n).astype(np.float32)
        t = np.zeros(64, dtype=np.float32)
        for i in range(n):
            lane = (i // 4) % 64
            t[lane] = np.float32(np.float64(a[i]) * np.float64(b[i]) + np.float64(t[lane]))
        for s in (32, 16, 8, 4, 2, 1):
            t = (t + t[np.arange(64) ^ s]).astype(np.float32)
        assert po.dot(a, b, po.MODE_C) == float(t[0])


def test_hamming_jaccard_large_and_thresholds():
    # simd_tests.rs:327-360 region: 768-D alternating patterns; threshold is strictly > 0.5
    a = np.where(np.arange(768) % 2 == 0, 1.0, 0.0).astype(np.float32)
    b = np.where(np.arange(768) % 3 == 0, 1.0, 0.0).astype(np.float32)
    exp = int(np.count_nonzero((a > 0.5) != (b > 0.5)))
    assert po.hamming(a, b) == exp
    inter = np.count_nonzero((a > 0.5) & (b > 0.5))
    uni = np.count_nonzero((a > 0.5) | (b > 0.5))
    assert po.jaccard(a, b) == float(np.float32(inter) / np.float32(uni))
    assert po.hamming([0.5], [0.50001]) == 1.0
    assert po.hamming([0.5], [0.5]) == 0.0
    assert po.hamming([float("nan")], [1.0]) == 1.0  # NaN > 0.5 is false
    # packed popcount Hamming (simd_explicit.rs:308-317)
    x = np.array([0xFFFF0000FFFF0000, 0x1], dtype=np.uint64)
    y = np.array([0x0, 0x3], dtype=np.uint64)
    assert po.hamming_binary(x, y) == 32 + 1


def test_scalar_engine_hamming_is_bit_pattern():
    # native/distance.rs:193-200 — CpuDistance Hamming is a different function
    assert po.distance(po.HAMMING, [1.0, 0.7, 0.0], [1.0, 0.9, -0.0], po.MODE_SCALAR) == 2.0
    assert po.distance(po.HAMMING, [1.0, 0.7, 0.0], [1.0, 0.9, -0.0], po.MODE_R) == 0.0


def test_total_cmp_order():
    nan = float("nan")
    neg_nan = -nan
    seq = [neg_nan, -math.inf, -1.0, -0.0, 0.0, 1e-45, 1.0, math.inf, nan]
    for i in range(len(seq) - 1):
        assert po.total_cmp(seq[i], seq[i + 1]) == -1
        assert po.total_cmp(seq[i + 1], seq[i]) == 1
    assert po.total_cmp(nan, nan) == 0


def test_zero_norm_and_engine_mapping():
    z, v = np.zeros(32, np.float32), np.ones(32, np.float32)
    for mode in (po.MODE_R, po.MODE_C, po.MODE_NATIVE):
        assert po.cosine(z, v, mode) == 0.0            # simd_avx512.rs:347-349
        assert po.distance(po.COSINE, z, v, mode) == 1.0
        assert po.distance(po.DOT, v, v, mode) == -32.0  # native/distance.rs:80
        assert po.compute_distance(po.DOT, v, v, mode) == 32.0
        assert po.distance(po.EUCLIDEAN, z, v, mode) == float(np.sqrt(np.float32(32)))
    assert po.distance(po.JACCARD, z, z, po.MODE_R) == 0.0  # 1 - 1.0
    assert math.isnan(po.transform_score(po.COSINE, float("nan")))


def test_dimension_mismatch_panics():
    # simd_tests.rs:220-227 should_panic "Vector dimensions must match"
    with pytest.raises(AssertionError, match="Vector dimensions must match"):
        po.cosine([1, 2, 3], [1, 2])
