"""GPU parity test of the bf16 GEMM-distance sweep (BASELINE configs[3]; reference semantics: half_precision.rs:199-255 —
bf16-rounded vectors, f32 accumulation).  The matrix-core instruction adds its 32 exact products in an undocumented
order, so the bar is the tolerance the north star states for f32 (1e-5 relative; here absolute on the cosine, and
relative to |q||v| for the dot product), with the tie-aware id rule of SURVEY.md §8(c): ids and ranks must agree with
the oracle wherever neighbouring oracle scores are further apart than the tolerance."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM = va.DistanceMetric


def check(metric, pm, rows, qs, k, gids, gsc, gcnt):
    eid, esc = po.scan_topk_bf16(pm, rows, qs, k, nthreads=4)
    rr, qq = po.round_bf16(rows).astype(np.float64), po.round_bf16(qs).astype(np.float64)
    full = qq @ rr.T
    scale = np.ones_like(full)
    if metric == DM.Cosine:
        full = full / (np.linalg.norm(qq, axis=1)[:, None] * np.linalg.norm(rr, axis=1)[None, :])
    else:
        scale = np.linalg.norm(qq, axis=1)[:, None] * np.linalg.norm(rr, axis=1)[None, :]
    tol = 1e-5
    for qi in range(qs.shape[0]):
        kk = min(k, rows.shape[0])
        assert gcnt[qi] == kk
        g_i, g_s = gids[qi, :kk].astype(np.int64), gsc[qi, :kk].astype(np.float64)
        # every returned score is the true (f64) score of that row within tolerance, and close to the f32 oracle's
        assert np.all(np.abs(g_s - full[qi, g_i]) <= tol * scale[qi, g_i])
        # best first
        assert np.all(np.diff(g_s) <= 1e-12)
        # nothing better was missed: k-th returned >= k-th true - tol
        kth_true = np.sort(full[qi])[::-1][kk - 1]
        assert g_s[-1] >= kth_true - tol * scale[qi].max()
        # tie-aware rank agreement with the f32 oracle
        e_i, e_s = eid[qi, :kk].astype(np.int64), esc[qi, :kk].astype(np.float64)
        for r in range(kk):
            if g_i[r] != e_i[r]:
                assert abs(e_s[r] - full[qi, g_i[r]]) <= 2 * tol * scale[qi, g_i[r]], (qi, r)


@pytest.mark.parametrize("metric,pm", [(DM.Cosine, po.COSINE), (DM.DotProduct, po.DOT)])
@pytest.mark.parametrize("n,dim", [(6000, 768), (3000, 256), (2000, 100), (500, 40), (40, 8)])
def test_bf16_sweep_matches_half_precision_semantics(metric, pm, n, dim):
    rng = np.random.default_rng(n + dim)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    ix = va.HnswIndex(dim, metric)
    ix.upload(np.arange(n // 2), rows[: n // 2])
    ix.enable_bf16()                       # converts what is there ...
    ix.upload(np.arange(n // 2, n), rows[n // 2:])  # ... and what arrives later
    for nq, k in [(1, 10), (20, 10), (70, 5), (100, 10), (3, 64)]:
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
        gi, gs, gc = ix.search_batch_brute_force_bf16(qs, k)
        check(metric, pm, rows, qs, k, gi, gs, gc)
    ix.close()


def test_bf16_needs_enable_and_metric():
    ix = va.HnswIndex(16, DM.Cosine)
    ix.upload(np.arange(10), np.ones((10, 16), np.float32))
    with pytest.raises(va.VelesHipError):
        ix.search_batch_brute_force_bf16(np.ones((1, 16), np.float32), 3)
    e = va.HnswIndex(16, DM.Euclidean)
    with pytest.raises(va.VelesHipError):
        e.enable_bf16()


def test_bf16_exact_values_are_exact():
    # values representable in bf16 with small integer products: every partial sum is exact, so the scores must be too
    rng = np.random.default_rng(5)
    rows = rng.integers(-4, 5, size=(1000, 64)).astype(np.float32)
    qs = rng.integers(-4, 5, size=(33, 64)).astype(np.float32)
    ix = va.HnswIndex(64, DM.DotProduct)
    ix.upload(np.arange(1000), rows)
    ix.enable_bf16()
    gi, gs, gc = ix.search_batch_brute_force_bf16(qs, 10)
    eid, esc = po.scan_topk_bf16(po.DOT, rows, qs, 10)
    assert np.array_equal(gi, eid) and np.array_equal(gs, esc)


@pytest.mark.parametrize("metric,pm", [(DM.Cosine, po.COSINE), (DM.DotProduct, po.DOT)])
@pytest.mark.parametrize("n,dim", [(10007, 768), (3000, 192), (50, 64), (129, 128)])
def test_bf16_gemm_sweep_large_batches(metric, pm, n, dim):
    # >= 64 queries and dim % 64 == 0: the GEMM-structured kernel over the bf16 rows (sweep_gemm.hip, BF16 variant)
    rng = np.random.default_rng(n * 7 + dim)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    ix = va.HnswIndex(dim, metric)
    ix.upload(np.arange(n), rows)
    ix.enable_bf16()
    for nq, k in [(64, 10), (129, 5), (300, 10)]:
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
        gi, gs, gc = ix.search_batch_brute_force_bf16(qs, k)
        check(metric, pm, rows, qs, k, gi, gs, gc)
    ix.close()


def test_bf16_gemm_exact_values_are_exact():
    rng = np.random.default_rng(6)
    rows = rng.integers(-4, 5, size=(5000, 128)).astype(np.float32)
    qs = rng.integers(-4, 5, size=(200, 128)).astype(np.float32)
    ix = va.HnswIndex(128, DM.DotProduct)
    ix.upload(np.arange(5000), rows)
    ix.enable_bf16()
    gi, gs, gc = ix.search_batch_brute_force_bf16(qs, 10)
    eid, esc = po.scan_topk_bf16(po.DOT, rows, qs, 10, nthreads=4)
    assert np.array_equal(gi, eid) and np.array_equal(gs, esc)
    # soft delete is honoured by the GEMM epilogue too
    assert ix.remove(int(gi[0, 0]))
    gi2, _, _ = ix.search_batch_brute_force_bf16(qs, 10)
    assert int(gi[0, 0]) not in gi2[0].tolist()


@pytest.mark.parametrize("metric,pm", [(DM.Cosine, po.COSINE), (DM.DotProduct, po.DOT)])
@pytest.mark.parametrize("n,dim", [(10007, 768), (700, 64), (300, 192)])
def test_bf16_gemm_big_tile(metric, pm, n, dim):
    # batches that fill 256-query tiles to >= 7/8 and k <= 16: the 256-row x 256-query tile (8 waves, one block per CU);
    # ragged query / row tiles
    rng = np.random.default_rng(n * 3 + dim)
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    ix = va.HnswIndex(dim, metric)
    ix.upload(np.arange(n), rows)
    ix.enable_bf16()
    for nq, k in [(230, 10), (480, 16), (700, 16), (1024, 1)]:
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
        gi, gs, gc = ix.search_batch_brute_force_bf16(qs, k)
        check(metric, pm, rows, qs, k, gi, gs, gc)
    # k > 16 needs 64-entry candidate buffers, which do not fit beside the big tile: the 128 x 128 tile takes over
    qs = rng.standard_normal((500, dim)).astype(np.float32)
    gi, gs, gc = ix.search_batch_brute_force_bf16(qs, 20)
    check(metric, pm, rows, qs, 20, gi, gs, gc)
    ix.close()


def test_bf16_gemm_big_tile_exact_and_deleted_rows():
    rng = np.random.default_rng(16)
    rows = rng.integers(-4, 5, size=(9000, 128)).astype(np.float32)
    qs = rng.integers(-4, 5, size=(520, 128)).astype(np.float32)
    ix = va.HnswIndex(128, DM.DotProduct)
    ix.upload(np.arange(9000), rows)
    ix.enable_bf16()
    gi, gs, gc = ix.search_batch_brute_force_bf16(qs, 10)
    eid, esc = po.scan_topk_bf16(po.DOT, rows, qs, 10, nthreads=4)
    assert np.array_equal(gi, eid) and np.array_equal(gs, esc)   # exact products: ties resolved by row like the oracle
    assert ix.remove(int(gi[7, 0]))
    gi2, _, _ = ix.search_batch_brute_force_bf16(qs, 10)
    assert int(gi[7, 0]) not in gi2[7].tolist()


# ---- BASELINE configs[3]'s own kernel: sweep_topk_gemm_bf16_glds in RESULT mode (>= 65 536 rows, batches that fill
# ---- 256-query tiles, k <= 10) — seed sweep + 1-2 LDS-DMA launches + merges (select_stage.hip brute_bf16_dev) -------------------
def check_sampled(metric, pm, rows, qs, k, gids, gsc, gcnt, sample, tol=1e-5):
    """The rule of check() for a SAMPLE of the batch's queries, with the f64 reference computed in row chunks (the whole
    f64 score matrix of 1 M rows does not fit)."""
    n = rows.shape[0]
    qsel = qs[sample]
    eid, esc = po.scan_topk_bf16(pm, rows, qsel, k, nthreads=po.host_threads())
    qq = po.round_bf16(qsel).astype(np.float64)
    qn = np.linalg.norm(qq, axis=1)
    full = np.empty((len(sample), n), np.float64)
    rnorm = np.empty(n, np.float64)
    for lo in range(0, n, 65536):
        rr = po.round_bf16(rows[lo:lo + 65536]).astype(np.float64)
        full[:, lo:lo + rr.shape[0]] = qq @ rr.T
        rnorm[lo:lo + rr.shape[0]] = np.linalg.norm(rr, axis=1)
    if metric == DM.Cosine:
        full /= qn[:, None] * rnorm[None, :]
        scale = np.ones_like(full)
    else:
        scale = qn[:, None] * rnorm[None, :]
    kk = min(k, n)
    for j, qi in enumerate(sample):
        assert gcnt[qi] == kk
        g_i, g_s = gids[qi, :kk].astype(np.int64), gsc[qi, :kk].astype(np.float64)
        assert len(set(g_i.tolist())) == kk
        assert np.all(np.abs(g_s - full[j, g_i]) <= tol * scale[j, g_i]), (qi,)
        assert np.all(np.diff(g_s) <= 1e-12)
        kth_true = np.partition(full[j], n - kk)[n - kk]
        assert g_s[-1] >= kth_true - tol * scale[j].max(), (qi,)
        e_i, e_s = eid[j, :kk].astype(np.int64), esc[j, :kk].astype(np.float64)
        for r in range(kk):
            if g_i[r] != e_i[r]:
                assert abs(e_s[r] - full[j, g_i[r]]) <= 2 * tol * scale[j, g_i[r]], (qi, r)


def served_by_glds(ix):
    assert ix.last_kernels() & va.KERNEL_GEMM_BF16_GLDS, "sweep_topk_gemm_bf16_glds did not serve the call (mask %#x)" % ix.last_kernels()


@pytest.mark.parametrize("metric,pm,n,dim,cases", [
    (DM.Cosine, po.COSINE, 70_000, 128, [(224, 10), (600, 1), (1024, 10)]),
    (DM.DotProduct, po.DOT, 70_000, 768, [(230, 10), (1024, 1)]),
    (DM.DotProduct, po.DOT, 300_001, 128, [(600, 10), (1024, 1)]),
    (DM.Cosine, po.COSINE, 300_001, 768, [(600, 10), (224, 1)]),
    (DM.Cosine, po.COSINE, 1_000_000, 768, [(1024, 10), (1024, 1)]),       # two LDS-DMA launches, re-seeded in between
    (DM.DotProduct, po.DOT, 1_000_000, 128, [(1024, 10), (480, 3)]),
])
def test_bf16_glds_result_mode(metric, pm, n, dim, cases):
    rng = np.random.default_rng(n * 13 + dim)
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    ix = va.HnswIndex(dim, metric, va.HnswParams(16, 100, n))
    ix.upload(np.arange(n), rows)
    ix.enable_bf16()
    for nq, k in cases:
        qs = rng.standard_normal((nq, dim), dtype=np.float32)
        gi, gs, gc = ix.search_batch_brute_force_bf16(qs, k)
        served_by_glds(ix)
        sample = np.unique(np.concatenate([[0, nq - 1, 255 % nq, 256 % nq], rng.integers(0, nq, 20)]))
        check_sampled(metric, pm, rows, qs, k, gi, gs, gc, sample)
    ix.close()


@pytest.mark.parametrize("metric,pm", [(DM.DotProduct, po.DOT), (DM.Cosine, po.COSINE)])
@pytest.mark.parametrize("n", [70_077, 600_077])   # one launch / two launches; ragged last row tile
def test_bf16_glds_exact_products_bit_equal(metric, pm, n):
    # small integers: every product and every partial sum is exact in f32 whatever the order, so ids, ranks (exact ties
    # broken by row) and score BITS must equal the oracle's — with duplicates of good rows planted in other row tiles,
    # row groups and launches (ties straddling tiles), zero rows, rows whose norm overflows / is below f32::EPSILON
    # (half_precision.rs:247: the score is 0.0), and soft-deleted rows
    dim, nq, k = 128, 300, 10
    rng = np.random.default_rng(n)
    rows = rng.integers(-4, 5, size=(n, dim)).astype(np.float32)
    qs = rng.integers(-4, 5, size=(nq, dim)).astype(np.float32)
    best = np.argsort(-(qs[:8] @ rows.T), axis=1)[:, :2].ravel()          # good rows for the first queries ...
    spots = np.array([255, 256, 257, 16383, 16384, 16385, 65535, 65536, 70_000, n - 2, n - 1, n // 2, n // 2 + 255, 300, 4000, 9999])
    rows[spots] = rows[best]                                            # ... duplicated across tile / launch boundaries
    rows[[5, 20_000, n - 3]] = 0.0
    rows[[7, 30_001]] = 3e19        # |v|^2 overflows: cosine = dot / inf = 0; the dot product itself stays finite and exact
    rows[[9, 40_003]] = 1e-9        # norm 1.1e-8 < f32::EPSILON: cosine 0.0 by the reference's rule (the plain quotient: +-1)
    qs[3] = 0.0                     # a zero query: every score 0, ties by row
    ix = va.HnswIndex(dim, metric, va.HnswParams(16, 100, n))
    ix.upload(np.arange(n), rows)
    ix.enable_bf16()
    gi, gs, gc = ix.search_batch_brute_force_bf16(qs, k)
    served_by_glds(ix)
    eid, esc = po.scan_topk_bf16(pm, rows, qs, k, nthreads=po.host_threads())
    assert np.array_equal(gi, eid)
    assert np.array_equal(gs.view(np.uint32), esc.view(np.uint32))
    assert np.all(gc == k)
    # soft deletes: the best row of some queries and one of the planted duplicates
    dead = sorted({int(gi[0, 0]), int(gi[17, 0]), int(spots[1]), int(gi[299, 9])})
    for d in dead:
        assert ix.remove(d)
    gi2, gs2, _ = ix.search_batch_brute_force_bf16(qs, k)
    served_by_glds(ix)
    keep = np.ones(n, bool)
    keep[dead] = False
    live = np.flatnonzero(keep)
    eid2, esc2 = po.scan_topk_bf16(pm, rows[keep], qs, k, nthreads=po.host_threads())
    assert np.array_equal(gi2, live[eid2.astype(np.int64)].astype(np.uint64))
    assert np.array_equal(gs2.view(np.uint32), esc2.view(np.uint32))
    ix.close()


def test_bf16_small_batches_follow_the_epsilon_rule_too():
    # the streaming kernel (<= 96 queries) and the register-staged GEMM kernels (< 65 536 rows): same half_precision.rs rule
    dim, n = 128, 5000
    rng = np.random.default_rng(77)
    rows = rng.integers(-4, 5, size=(n, dim)).astype(np.float32)
    rows[[9, 4003]] = 1e-9
    rows[[11]] = 0.0
    ix = va.HnswIndex(dim, DM.Cosine)
    ix.upload(np.arange(n), rows)
    ix.enable_bf16()
    for nq in (3, 70, 300):
        qs = -np.abs(rng.integers(-4, 5, size=(nq, dim))).astype(np.float32)   # mostly negative scores: the 0.0 rows rank high
        qs[:, ::2] *= -1
        gi, gs, _ = ix.search_batch_brute_force_bf16(qs, 10)
        eid, esc = po.scan_topk_bf16(po.COSINE, rows, qs, 10, nthreads=4)
        assert np.array_equal(gi, eid) and np.array_equal(gs.view(np.uint32), esc.view(np.uint32)), nq
    ix.close()
