"""Pins the oracle's `DistanceEngine` restatements — native/distance.rs: CpuDistance (scalar), SimdDistance (the production engine),
NativeSimdDistance (intrinsics), `distance` :74-85 and `batch_distance` :101-135 — against the reference's own tests in that file
(:221-625), transcribed as data with their lines.  (Three of them were already in tests/golden/reference_kats.json and
tests/test_oracle_kernels.py; this file holds the rest.)  The GPU side of the same surface: vdb_hip_batch_distance,
tests/test_gpu_sweep.py::test_batch_distance_bit_exact.  CPU only."""
import numpy as np
import pytest

from oracle import pyoracle as po

F = np.float32
ENGINES = {"cpu": po.MODE_SCALAR, "simd": po.MODE_R, "native": po.MODE_NATIVE, "gpu-declared": po.MODE_C}

# (engine, metric, a, b, expected engine DISTANCE, tolerance, native/distance.rs lines)
KATS = [
    ("native", po.EUCLIDEAN, [0.0, 0.0, 0.0, 0.0], [3.0, 4.0, 0.0, 0.0], 5.0, 1e-5, "426-434"),
    ("cpu", po.DOT, [1.0, 2.0, 3.0], [4.0, 5.0, 6.0], -32.0, 1e-5, "453-460"),           # distance = -dot
    ("cpu", po.HAMMING, [1.0, 0.0, 1.0, 0.0], [1.0, 1.0, 0.0, 0.0], 2.0, 1e-5, "463-470"),
    ("cpu", po.JACCARD, [1.0, 1.0, 0.0, 0.0], [1.0, 0.0, 1.0, 0.0], 1.0 - 1.0 / 3.0, 1e-5, "473-483"),
    ("simd", po.DOT, [1.0, 2.0, 3.0, 4.0], [1.0, 1.0, 1.0, 1.0], -10.0, 1e-4, "504-511"),
    ("simd", po.EUCLIDEAN, [0.0, 0.0, 0.0, 0.0], [3.0, 4.0, 0.0, 0.0], 5.0, 1e-4, "514-520"),
    ("cpu", po.COSINE, [0.0, 0.0, 0.0], [1.0, 2.0, 3.0], 1.0, 1e-5, "575-584"),           # zero norm: distance 1.0
    ("cpu", po.JACCARD, [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], 1.0, 1e-5, "587-596"),          # zero union: distance 1.0 (the SCALAR engine)
    ("cpu", po.HAMMING, [1.0, 2.0, 3.0], [1.0, 2.0, 3.0], 0.0, 1e-5, "613-617"),
    ("cpu", po.HAMMING, [1.0, 2.0, 3.0], [4.0, 5.0, 6.0], 3.0, 1e-5, "620-625"),          # the scalar engine compares bit patterns
]


@pytest.mark.parametrize("engine,metric,a,b,exp,tol,src", KATS, ids=[f"{k[0]}-{k[1]}@distance.rs:{k[6]}" for k in KATS])
def test_engine_distance_kats(engine, metric, a, b, exp, tol, src):
    assert abs(po.distance(metric, a, b, ENGINES[engine]) - exp) <= tol
    if engine != "cpu" and metric in (po.EUCLIDEAN, po.DOT):
        # the order the GPU declares (mode C) is held to the same literal
        assert abs(po.distance(metric, a, b, po.MODE_C) - exp) <= tol


def test_scalar_and_simd_engines_disagree_where_the_reference_does():
    """The reference has TWO Jaccard / Hamming definitions (CpuDistance: min / max sums and bit patterns, :186-219; SimdDistance: the
    0.5-threshold set forms of simd_explicit.rs) — the empty union is distance 1.0 in one and 0.0 in the other.  The oracle keeps both;
    the GPU implements the production (SIMD) one."""
    z = [0.0, 0.0, 0.0]
    assert po.distance(po.JACCARD, z, z, po.MODE_SCALAR) == 1.0      # :587-596
    assert po.distance(po.JACCARD, z, z, po.MODE_R) == 0.0           # 1 - jaccard_similarity_simd = 1 - 1.0
    assert po.distance(po.HAMMING, [1.0, 2.0, 3.0], [4.0, 5.0, 6.0], po.MODE_R) == 0.0   # all six values are > 0.5: the same set


def test_simd_hamming_and_jaccard_patterns():
    """:266-333, :523-542 — range assertions in the reference; the exact values of the same patterns here"""
    i = np.arange(64)
    a, b = (i % 2 == 0).astype(F), (i % 3 == 0).astype(F)
    d = po.distance(po.HAMMING, a, b, po.MODE_R)
    assert 0.0 <= d <= 64.0
    assert d == float(np.count_nonzero((i % 2 == 0) != (i % 3 == 0))) == 32.0
    dj = po.distance(po.JACCARD, a, b, po.MODE_R)                    # :285-306
    inter, union = np.count_nonzero((i % 2 == 0) & (i % 3 == 0)), np.count_nonzero((i % 2 == 0) | (i % 3 == 0))
    assert 0.0 <= dj <= 1.0 and abs(dj - (1.0 - inter / union)) < 1e-6
    assert po.distance(po.HAMMING, a, a, po.MODE_R) == 0.0           # :309-320
    assert abs(po.distance(po.JACCARD, a, a, po.MODE_R)) < 1e-6      # :323-333
    for eng in ("simd", "native"):
        i32 = np.arange(32)
        assert po.distance(po.HAMMING, (i32 % 2 == 0).astype(F), (i32 % 3 == 0).astype(F), ENGINES[eng]) >= 0.0   # :523-533
        dj = po.distance(po.JACCARD, [1.0, 1.0, 0.0, 0.0], [1.0, 1.0, 1.0, 0.0], ENGINES[eng])                    # :536-542
        assert abs(dj - (1.0 - 2.0 / 3.0)) < 1e-6


def test_batch_distance_with_prefetch_100x768():
    """:337-359 — 100 candidates cos(0.01 (i + 10 j)) against sin(0.01 i): 100 cosine distances, all within [0, 2]"""
    i = np.arange(768, dtype=F)
    q = np.sin(i * F(0.01), dtype=F)
    cands = np.stack([np.cos((i + F(10 * j)) * F(0.01), dtype=F) for j in range(100)])
    for mode in (po.MODE_R, po.MODE_C, po.MODE_NATIVE, po.MODE_SCALAR):
        d = po.batch_distance(po.COSINE, q, cands, mode)
        assert d.shape == (100,) and np.all((d >= 0.0) & (d <= 2.0))


@pytest.mark.parametrize("mode", [po.MODE_R, po.MODE_C, po.MODE_NATIVE, po.MODE_SCALAR])
def test_batch_distance_equals_individual_distances(mode):
    """:362-392 (|batch - individual| < 1e-6; the oracle's batch form is the same function: bit-equal), :599-610 (default impl)"""
    q = np.arange(128, dtype=F)
    cands = np.stack([np.arange(128, dtype=F) + F(j) for j in range(20)])
    batch = po.batch_distance(po.EUCLIDEAN, q, cands, mode)
    single = np.array([po.distance(po.EUCLIDEAN, q, c, mode) for c in cands], dtype=F)
    assert np.array_equal(batch.view(np.uint32), single.view(np.uint32))
    assert np.all(np.abs(batch - np.sqrt(F(128)) * np.arange(20, dtype=F)) < 1e-3)   # rows differ from the query by j in every dimension
    d = po.batch_distance(po.EUCLIDEAN, [0.0, 0.0, 0.0], np.array([[1.0, 0.0, 0.0], [0.0, 2.0, 0.0]], dtype=F), mode)
    assert abs(d[0] - 1.0) < 1e-5 and abs(d[1] - 2.0) < 1e-5


def test_batch_distance_empty():
    """:395-402"""
    assert po.batch_distance(po.COSINE, [1.0, 2.0, 3.0], np.empty((0, 3), dtype=F)).shape == (0,)


def test_native_engine_matches_simd_engine():
    """:409-423 (|simd - native| < 1e-3 on the 768-dimensional sin / cos pair), :437-450 (dot distance negative), :545-572 (batches)"""
    i = np.arange(768, dtype=F)
    a, b = np.sin(i * F(0.01), dtype=F), np.cos(i * F(0.02), dtype=F)
    assert abs(po.distance(po.COSINE, a, b, po.MODE_R) - po.distance(po.COSINE, a, b, po.MODE_NATIVE)) < 1e-3
    j = np.arange(128, dtype=F)
    assert po.distance(po.DOT, j * F(0.1), (F(128) - j) * F(0.1), po.MODE_NATIVE) < 0.0
    cands = np.stack([np.full(16, j + 1, dtype=F) for j in range(5)])
    d = po.batch_distance(po.DOT, np.ones(16, dtype=F), cands, po.MODE_NATIVE)
    assert d.shape == (5,) and np.all(np.abs(d - (-16.0 * np.arange(1, 6))) < 1e-3)
    d = po.batch_distance(po.EUCLIDEAN, np.zeros(8, dtype=F), np.stack([np.ones(8, dtype=F), np.full(8, 2.0, dtype=F)]), po.MODE_NATIVE)
    assert d.shape == (2,) and abs(d[0] - np.sqrt(8.0)) < 1e-5 and abs(d[1] - 2 * np.sqrt(8.0)) < 1e-5


def test_gpu_accelerator_kats_on_the_function_the_gpu_is_compared_with():
    """gpu/gpu_backend_tests.rs:30-178 — GpuAccelerator::batch_cosine_similarity / batch_euclidean_distance / batch_dot_product on flat
    arrays (RAW similarities, within 0.01 in the reference).  The HIP side (`vdb_hip_batch_distance`, kind RAW) is bit-compared with
    `batch_compute_distance` in mode C (tests/test_gpu_sweep.py::test_batch_distance_bit_exact); here that function meets the
    reference's literals, exactly."""
    def run(metric, vectors, query, dim):
        v = np.asarray(vectors, dtype=F).reshape(-1, dim) if len(vectors) else np.empty((0, dim), dtype=F)
        return po.batch_compute_distance(metric, np.asarray(query, dtype=F), v, po.MODE_C)

    assert run(po.COSINE, [], [1.0, 0.0, 0.0], 3).size == 0                                     # :30-35
    assert run(po.COSINE, [1.0, 0.0, 0.0], [1.0, 0.0, 0.0], 3).tolist() == [1.0]                # :39-54
    assert run(po.COSINE, [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], 3).tolist() == [0.0]                # :58-68
    multi = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, -1.0, 0.0, 0.0]                                      # :72-89 same / orthogonal / opposite
    assert run(po.COSINE, multi, [1.0, 0.0, 0.0], 3).tolist() == [1.0, 0.0, -1.0]
    assert run(po.EUCLIDEAN, [], [1.0, 0.0, 0.0], 3).size == 0                                  # :97-102
    assert run(po.EUCLIDEAN, [1.0, 2.0, 3.0], [1.0, 2.0, 3.0], 3).tolist() == [0.0]             # :106-116
    assert run(po.EUCLIDEAN, [3.0, 4.0, 0.0], [0.0, 0.0, 0.0], 3).tolist() == [5.0]             # :120-134
    assert run(po.DOT, [], [1.0, 0.0, 0.0], 3).size == 0                                        # :142-147
    assert run(po.DOT, [0.0, 1.0, 0.0], [1.0, 0.0, 0.0], 3).tolist() == [0.0]                   # :151-161
    assert run(po.DOT, [2.0, 3.0, 4.0], [2.0, 3.0, 4.0], 3).tolist() == [29.0]                  # :165-178
