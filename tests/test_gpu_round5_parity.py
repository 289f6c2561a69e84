"""The three parity questions round 4 wrote down and did not answer (Weak 1), as fixed `-m gpu` tests:

  (a) SUBNORMAL PRODUCTS ON THE MATRIX PIPE — the reference's `test_very_small_values` input (simd_avx512_tests.rs:335: components
      1e-20, products 1e-40, subnormal in f32) and mixed-scale rows through the exact f32 sweep at 1, 4, 68 and 96 queries
      (streaming matrix-core kernel, GEMM-structured kernel, selection stage): ids and score BITS equal the oracle in the mode the
      index reports.  Measured answer (profiles/r05a_subnormal_products_matrix_pipe.log): v_mfma_f32_16x16x4_f32 and the bf16
      selection + exact re-scoring do NOT flush subnormal products differently from the oracle's fmaf chain — 24 / 24 cases bit-equal;
  (b) EUCLIDEAN AND SQ8 AT THE configs[4] SHARD SIZE — 6 250 000 x 768 rows (4.8e9 elements, beyond 32-bit element offsets) through
      the Euclidean selection (augmented bf16 image + l2_rerank_verify) and the SQ8 storage mode (4.8 GB of codes, the dequantised
      bf16 image, the one-lane-per-row exact sweep for small calls) against the oracle's chunked scan
      (profiles/r05a_shard_size_6p25m_euclidean_sq8.log: 8 / 8);
  (c) HAMMING / JACCARD AT 1 M x 768 x 1 024 QUERIES THROUGH THE FOUR-BIT MATRIX PATH — the shape bench.py times
      (simd_explicit.rs:234-287, 372-443 arithmetic; multi-launch schedule, seeded thresholds): 40 sampled queries, ids + integer
      score bits == oracle, ties at the cut included.
"""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM = va.DistanceMetric
D, K = 768, 10


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ------------------------------------------------------------------------------------------------------------------ (a)
def _subnormal_cases():
    rng = np.random.default_rng(5)
    n = 70_000                              # >= 65 536 rows: large batches reach the selection stage
    tiny_rows = (rng.standard_normal((n, D)) * 1e-20).astype(np.float32)
    tiny_q = (rng.standard_normal((96, D)) * 1e-20).astype(np.float32)
    mixed = rng.standard_normal((n, D)).astype(np.float32)
    mixed[::7] *= np.float32(1e-25)         # every 7th row: products with a unit-scale query are subnormal-ish
    mixed[::11] *= np.float32(1e-38)
    unit_q = rng.standard_normal((96, D)).astype(np.float32)
    unit_rows = rng.standard_normal((n, D)).astype(np.float32)
    return {"all-tiny": (tiny_rows, tiny_q), "mixed-scale rows": (mixed, unit_q), "tiny queries, unit rows": (unit_rows, tiny_q)}


@pytest.fixture(scope="module")
def subnormal_cases():
    if va.device_count() == 0:
        pytest.skip("no HIP device visible")
    return _subnormal_cases()


@pytest.mark.parametrize("metric", [DM.DotProduct, DM.Cosine])
@pytest.mark.parametrize("case", ["all-tiny", "mixed-scale rows", "tiny queries, unit rows"])
def test_subnormal_products_on_the_matrix_pipe_bit_equal_oracle(gpu_required, subnormal_cases, metric, case):
    """simd_avx512_tests.rs:335 test_very_small_values (components 1e-20) scaled up to whole sweeps: the matrix-core kernels
    (v_mfma_f32_16x16x4_f32 exact sweeps; bf16 selection + exact re-scoring + proof) against the oracle's scan, ids AND score bits.
    If the matrix pipe flushed subnormal products or accumulators, the all-tiny scores (|q.v| ~ 1e-38) would come out 0 or differ
    in their last bits, and the selection proof's error bound (relative to |q||v|) would be unsound there."""
    rows, queries = subnormal_cases[case]
    ix = va.HnswIndex(D, metric)
    ix.upload(np.arange(rows.shape[0], dtype=np.uint64), rows)
    nt = po.host_threads()
    try:
        for nq in (1, 4, 68, 96):
            q = queries[:nq]
            ids, sc, cnt = ix.search_batch_brute_force(q, K)
            mode = po.MODE_M if ix.sweep_arith_mode(K) == "M" else po.MODE_C
            eid, esc = po.scan_topk(int(metric), rows, q, K, mode, nthreads=nt)
            assert np.all(cnt == K)
            assert np.array_equal(ids, eid), (case, metric, nq, ix.last_select_level())
            assert np.array_equal(bits(sc), bits(esc)), (case, metric, nq, ix.last_select_level(), sc[0][:3], esc[0][:3])
            if case == "all-tiny" and metric == DM.DotProduct:
                # the scores themselves are subnormal or nearly so, and not all flushed to zero
                assert np.all(np.abs(esc) < 1e-35) and np.count_nonzero(sc) > sc.size // 2
    finally:
        ix.close()


# ------------------------------------------------------------------------------------------------------------------ (b)
def test_configs4_shard_size_6p25m_euclidean_and_sq8_vs_oracle(gpu_required):
    """One shard of BASELINE configs[4] (6 250 000 x 768) through the two paths its f32 cosine sibling
    (test_gpu_headline_sizes.py::test_configs4_shard_size_6p25m_f32_vs_oracle) does not reach: Euclidean batches (augmented bf16 image,
    l2_rerank_verify; distance.rs:76-103 ordering — ascending) and the SQ8 storage mode (quantization.rs:410-554 asymmetric distances).
    Rows are generated chunk-wise on the device, every chunk is scanned once by the oracle on the host for sampled queries, per-chunk
    lists are merged in the canonical order.  Bar: ids AND score bits."""
    torch = pytest.importorskip("torch")
    SR, BQ, chunk = 6_250_000, 1024, 1_000_000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    stream = torch.cuda.current_stream().cuda_stream
    gq = torch.Generator(device=dev)
    gq.manual_seed(48)
    qs = torch.randn((BQ, D), generator=gq, device=dev).cpu().numpy()
    rng = np.random.default_rng(9)
    s_l2 = np.unique(np.concatenate([[0, 255, 256, 1023], rng.integers(0, BQ, 8)]))
    s_sq = np.unique(np.concatenate([[0, 1023], rng.integers(0, BQ, 4)]))
    ixe = va.HnswIndex(D, DM.Euclidean, va.HnswParams(16, 100, SR))
    ixs = va.HnswIndex(D, DM.Cosine, va.HnswParams(16, 100, SR))
    ixs.set_storage_mode(va.StorageMode.SQ8)
    nt = po.host_threads()
    acc = {"l2": [np.empty((len(s_l2), 0), np.int64), np.empty((len(s_l2), 0), np.float32)],
           "sq": [np.empty((len(s_sq), 0), np.int64), np.empty((len(s_sq), 0), np.float32)]}

    def fold(key, ei, es, base, ascending):
        bi = np.concatenate([acc[key][0], ei.astype(np.int64) + base], axis=1)
        bs = np.concatenate([acc[key][1], es], axis=1)
        s64 = bs.astype(np.float64)
        order = np.lexsort((bi, s64 if ascending else -s64), axis=1)[:, :K]
        acc[key] = [np.take_along_axis(bi, order, axis=1), np.take_along_axis(bs, order, axis=1)]

    try:
        for base in range(0, SR, chunk):
            n_c = min(chunk, SR - base)
            c = torch.randn((n_c, D), generator=g, device=dev)
            torch.cuda.synchronize()
            ixe.upload_dev(base, c.data_ptr(), n_c, stream)
            ixs.upload_dev(base, c.data_ptr(), n_c, stream)
            torch.cuda.synchronize()
            host = c.cpu().numpy()
            del c
            ei, es = po.scan_topk(po.EUCLIDEAN, host, qs[s_l2], K, po.MODE_C, nthreads=nt)
            fold("l2", ei, es, base, True)
            ei, es = po.scan_topk_sq8(po.COSINE, host, qs[s_sq], K, nthreads=nt)
            fold("sq", ei, es, base, False)
            del host
        # Euclidean, 1 024 queries: the selection stage over the augmented image
        gi, gs, gc = ixe.search_batch_brute_force(qs, K)
        assert ixe.last_select_level() == 4, ixe.last_split_stats()   # (4 = the WIDE selection, the default at every k since round 6)
        ixe.set_option(va.OPT_SELECTOR_LEVEL, 2)                      # ... and pinned at level 2 (block-local lists): the same bits
        gi2, gs2, _ = ixe.search_batch_brute_force(qs, K)
        assert ixe.last_select_level() == 2 and np.array_equal(gi2, gi) and np.array_equal(bits(gs2), bits(gs))
        ixe.set_option(va.OPT_SELECTOR_LEVEL, -1)
        assert np.all(gc == K) and np.array_equal(gi[s_l2].astype(np.int64), acc["l2"][0])
        assert np.array_equal(bits(gs[s_l2]), bits(acc["l2"][1]))
        i4, s4, _ = ixe.search_batch_brute_force(qs[s_l2[:4]], K)      # the small-batch exact kernel over the same 4.8e9 elements
        assert np.array_equal(i4.astype(np.int64), acc["l2"][0][:4]) and np.array_equal(bits(s4), bits(acc["l2"][1][:4]))
        ixe.close()
        ixe = None
        torch.cuda.empty_cache()
        # SQ8 storage mode, 1 024 queries: selection over the dequantised image + the reference's chain; then the exact code sweep
        gi, gs, gc = ixs.search_batch_sq8(qs, K)
        assert ixs.last_select_level() == 4, ixs.last_split_stats()    # (selector level 3, the default: the WIDE selection at every k)
        assert np.all(gc == K) and np.array_equal(gi[s_sq].astype(np.int64), acc["sq"][0])
        assert np.array_equal(bits(gs[s_sq]), bits(acc["sq"][1]))
        va.set_split_selector(2)                                       # pinned: the block-local lists of the SQ8 mode, same bits
        gi2, gs2, _ = ixs.search_batch_sq8(qs, K)
        assert ixs.last_select_level() == 3, ixs.last_split_stats()
        va.set_split_selector(3)
        assert np.array_equal(gi2, gi) and np.array_equal(bits(gs2), bits(gs))
        i4, s4, _ = ixs.search_batch_sq8(qs[s_sq[:4]], K)
        assert np.array_equal(i4.astype(np.int64), acc["sq"][0][:4]) and np.array_equal(bits(s4), bits(acc["sq"][1][:4]))
    finally:
        if ixe is not None:
            ixe.close()
        ixs.close()
        torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------------------------ (c)
@pytest.mark.parametrize("metric", [DM.Hamming, DM.Jaccard])
def test_bit_metrics_1m_x_768_x_1024_queries_vs_oracle(gpu_required, metric):
    """SURVEY 8(d)'s Hamming workload at full size through the path bench.py times: 1 000 000 x 768 N(0,1) rows thresholded at 0.5
    (simd_explicit.rs:234-287 hamming_distance, :372-443 jaccard_similarity over f32 inputs), 1 024 queries in ONE call, k = 10 — the
    four-bit GEMM on v_mfma_scale_f32_16x16x128_f8f6f4 with its multi-launch schedule and seeded thresholds.  Integer distances are
    ~Binomial(768, 0.427) (sigma 13.7): hundreds of rows tie at the cut, so rank 10 is decided by the canonical (distance, row)
    order in almost every query.  40 sampled queries: ids + score bits == oracle; every query: equal to the vector-ALU kernels."""
    N, BQ = 1_000_000, 1024
    rng = np.random.default_rng(1234 + int(metric))
    rows = np.empty((N, D), dtype=np.float32)
    for lo in range(0, N, 100_000):
        rows[lo:lo + 100_000] = (rng.standard_normal((100_000, D), dtype=np.float32) > 0.5)
    qs = (rng.standard_normal((BQ, D), dtype=np.float32) > 0.5).astype(np.float32)
    qs[5] = rows[123_456]                    # an exact hit: distance 0 / similarity 1
    qs[6] = 0.0                              # the empty query (Jaccard: empty unions score 1.0 only against empty rows)
    ix = va.HnswIndex(D, metric, va.HnswParams(16, 100, N))
    try:
        assert ix.upload(np.arange(N, dtype=np.uint64), rows) == N
        gi, gs, gc = ix.search_batch_brute_force(qs, K)
        assert ix.last_kernels() & va.KERNEL_BITS_GEMM, "the four-bit matrix path did not serve the batch"
        assert np.all(gc == K)
        sel = np.unique(np.concatenate([[0, 5, 6, 255, 256, 1023], rng.choice(BQ, 36, replace=False)]))
        ei, es = po.scan_topk(int(metric), rows, qs[sel], K, po.MODE_C, nthreads=po.host_threads())
        assert np.array_equal(gi[sel].astype(np.int64), ei.astype(np.int64)), "ids / ranks differ from the oracle (ties at the cut?)"
        assert np.array_equal(bits(gs[sel]), bits(es))
        assert int(gi[5, 0]) == 123_456
        # ties at the cut really occur at this size: most sampled queries have a rank-10 score shared with rank 9 or rank 11-to-be
        if metric == DM.Hamming:
            tied = sum(1 for j in range(len(sel)) if es[j, K - 1] == es[j, K - 2])
            assert tied >= len(sel) // 4, tied
        # every query against the vector-ALU kernels (AND + popcount over the packed rows): same ids, same bits
        ix.set_option(va.OPT_SWEEP_ENGINE, 0)
        vi, vs, vc = ix.search_batch_brute_force(qs, K)
        assert not (ix.last_kernels() & va.KERNEL_BITS_GEMM)
        ix.set_option(va.OPT_SWEEP_ENGINE, -1)
        assert np.array_equal(gi, vi) and np.array_equal(bits(gs), bits(vs)) and np.array_equal(gc, vc)
    finally:
        ix.close()


# ------------------------------------------------------------------ one or two packed-bit queries in ONE launch (sweep_bits_fused)
@pytest.mark.parametrize("metric", [DM.Hamming, DM.Jaccard])
@pytest.mark.parametrize("n,dim", [(1_200_000, 128), (300_000, 768), (70_001, 384), (4_100, 1536), (130, 256), (64, 128)])
def test_one_launch_packed_bit_query_vs_oracle_and_three_launch_path(gpu_required, metric, n, dim):
    """A call of one or two Hamming / Jaccard queries is ONE kernel (sweep.hip sweep_bits_fused: the blocks pack the query, keep their
    keys in registers, extract the k best, and the block that draws the last ticket merges).  Ids, order (ties by row: dim 128 ties a
    lot) and score bits must equal the oracle's (simd_explicit.rs:234-287, 372-443) and the three-launch path's (engine 0); 1.2 M rows
    make a wave carry its list over a second batch of chunks; soft-deleted rows, external ids, k = 1 / 10 / 16, a ragged last chunk,
    fewer rows than k, an empty query (Jaccard 1.0 against empty rows) and back-to-back calls (the ticket counter returns to zero)."""
    rng = np.random.default_rng(n + dim + int(metric))
    rows = (rng.random((n, dim)) > 0.6915).astype(np.float32)
    rows[min(7, n - 1)] = 0.0
    ids = np.arange(n, dtype=np.uint64) * 3 + 11
    ix = va.HnswIndex(dim, metric)
    assert ix.upload(ids, rows) == n
    Q = (rng.random((6, dim)) > 0.6915).astype(np.float32)
    Q[1] = 0.0
    live = None
    for phase in range(2):
        for k in (10, 1, 16):
            for q0, nq in ((0, 1), (1, 1), (2, 2), (4, 1), (5, 1)):
                qs = Q[q0:q0 + nq]
                ix.set_option(va.OPT_SWEEP_ENGINE, 1)
                gid, gsc, gcnt = ix.search_batch_brute_force(qs, k)
                assert ix.last_kernels() & va.KERNEL_BITS
                ix.set_option(va.OPT_SWEEP_ENGINE, 0)
                oid, osc, ocnt = ix.search_batch_brute_force(qs, k)
                ix.set_option(va.OPT_SWEEP_ENGINE, -1)
                sel = np.arange(n) if live is None else np.nonzero(live)[0]
                kk = min(k, len(sel))
                r, s = po.scan_topk(int(metric), rows[sel], qs, max(kk, 1), po.MODE_C)
                for qi in range(nq):
                    assert gcnt[qi] == kk == ocnt[qi], (metric, n, dim, k, q0, qi)
                    assert np.array_equal(gid[qi, :kk], ids[sel[r[qi, :kk].astype(np.int64)]]), (metric, n, dim, k, q0, qi)
                    assert np.array_equal(bits(gsc[qi, :kk]), bits(s[qi, :kk]))
                    assert np.array_equal(gid[qi, :kk], oid[qi, :kk]) and np.array_equal(bits(gsc[qi, :kk]), bits(osc[qi, :kk]))
        if phase == 0:   # soft deletes: a seventh of the rows, among them the best answers of query 0
            dead = set(rng.choice(n, max(1, n // 7), replace=False).tolist())
            first = ix.search_batch_brute_force(Q[:1], min(5, n))[0][0]
            dead.update(int((i - 11) // 3) for i in first[:3])
            for d in dead:
                assert ix.remove(int(ids[d]))
            live = np.ones(n, bool)
            live[list(dead)] = False
    ix.close()
