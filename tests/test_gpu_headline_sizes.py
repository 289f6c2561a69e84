"""GPU parity tests AT THE HEADLINE SIZES (BASELINE.json configs[1] / configs[2]: 1 000 000 x 768 f32 N(0,1), cosine,
k = 10) — the launches `bench.py` quotes its numbers on, compared with the oracle on the same inputs:

  * the GEMM-structured matrix-core sweep with several query tiles in ONE search_batch_brute_force call
    (HnswIndex::search_brute_force over a batch, search.rs:176-219): ids + score bits == oracle mode M;
  * the batched GPU graph construction at 1 M nodes followed by the traversal kernel (NativeHnsw::search,
    native/graph.rs:251-270,438-520) on >= 1 024 queries, ef = 128: ids + score bits + the kernel's distance-evaluation /
    expansion counters == oracle mode C over the very same graph (handed over in the reference's file format);
    the fraction of queries whose id lists differ from the reference's own summation order (mode R) is measured and bounded.

The corpus is 3 GB of host memory (module-scoped); the whole module runs in about two minutes on the MI355X box.
"""
import os

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM = va.DistanceMetric
SQ = va.SearchQuality

N, D, K = 1_000_000, 768, 10


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def corpus():
    if va.device_count() == 0:
        pytest.skip("no HIP device visible")
    rng = np.random.default_rng(42)
    rows = np.empty((N, D), dtype=np.float32)
    for lo in range(0, N, 100_000):  # bounded temporaries
        rows[lo:lo + 100_000] = rng.standard_normal((100_000, D), dtype=np.float32)
    qs = np.random.default_rng(43).standard_normal((1024, D), dtype=np.float32)
    return rows, qs


@pytest.fixture(scope="module")
def index(corpus):
    rows, _ = corpus
    ix = va.HnswIndex(D, DM.Cosine, va.HnswParams(32, 400, N))
    assert ix.upload(np.arange(N, dtype=np.uint64), rows) == N
    yield ix
    ix.close()


def test_headline_1m_gemm_vs_oracle(corpus, index):
    """configs[1] at full size: 320 queries in one call = the GEMM kernel with 3 query tiles (107 queries each, NQF = 4)
    over 1 M rows (>= 15 row tiles per block); then the bench's 1 024-query launch shape (8 query tiles), of which 96
    spread over every tile are compared."""
    rows, qs = corpus
    ncores = po.host_threads()
    assert index.sweep_arith_mode(K) == "M"
    nq = 320
    ids, sc, cnt = index.search_batch_brute_force(qs[:nq], K)
    # WHICH kernel answered: the selection stage (bf16 matrix cores + exact re-scoring + proof; level 4 = the WIDE selection, the default
    # at every k since round 6) — a silent fall-back to the exact f32 kernel would produce the same bits and pass everything below
    assert index.last_select_level() == 4 and index.last_kernels() & va.KERNEL_SELECT_BF16, "the selection stage did not serve the 320-query call"
    eid, esc = po.scan_topk(po.COSINE, rows, qs[:nq], K, po.MODE_M, nthreads=ncores)
    assert np.all(cnt == K)
    assert np.array_equal(ids, eid), "ids / ranks differ from the oracle (mode M) at 1M x 320 queries"
    assert np.array_equal(bits(sc), bits(esc)), "score bits differ from the oracle (mode M)"
    # the bench's launch: 1 024 queries in one call; queries 320.. are new, every query tile is sampled
    ids2, sc2, cnt2 = index.search_batch_brute_force(qs, K)
    assert index.last_select_level() == 4 and index.last_kernels() & va.KERNEL_SELECT_BF16, "the selection stage did not serve the 1 024-query call"
    # pinned at level 2 (block-local candidate lists, the per-query proof): the same bits from the other kernel instance
    index.set_option(va.OPT_SELECTOR_LEVEL, 2)
    ids_l2, sc_l2, _ = index.search_batch_brute_force(qs, K)
    assert index.last_select_level() == 2
    index.set_option(va.OPT_SELECTOR_LEVEL, -1)
    assert np.array_equal(ids_l2, ids2) and np.array_equal(bits(sc_l2), bits(sc2)), "levels 2 and 4 disagree at 1M x 1 024 queries"
    assert index.last_split_stats()[1] <= 2, "more unproven queries than the benchmark data ever produced (one in ~8 000): the bound has drifted"
    assert np.array_equal(ids2[:nq], ids) and np.array_equal(bits(sc2[:nq]), bits(sc)), "results depend on the batch size"
    sample = np.arange(nq + 3, 1024, 7)[:96]
    eid2, esc2 = po.scan_topk(po.COSINE, rows, qs[sample], K, po.MODE_M, nthreads=ncores)
    assert np.array_equal(ids2[sample], eid2) and np.array_equal(bits(sc2[sample]), bits(esc2))
    # north-star tolerance against the reference's own summation order (mode R): 1e-5 relative, tie-aware ids
    rid, rsc = po.scan_topk(po.COSINE, rows, qs[:64], K, po.MODE_R, nthreads=ncores)
    assert np.max(np.abs(sc[:64] - rsc) / np.abs(rsc)) < 1e-5
    for i in range(64):
        if not np.array_equal(ids[i], rid[i]):  # a swap is only legitimate inside a reference tie group
            diff = np.nonzero(ids[i] != rid[i])[0]
            gaps = np.abs(rsc[i][diff] - sc[i][diff]) / np.abs(rsc[i][diff])
            assert np.all(gaps < 1e-5), f"query {i}: ids differ from mode R outside a tie group"


@pytest.mark.parametrize("k", [50, 100])
def test_headline_1m_k50_k100_vs_oracle(corpus, index, k):
    """configs[1]'s corpus at the k the reference also benches (benches/hnsw_benchmark.rs:152-159: 10 / 50 / 100): 1 024 queries in one
    call through the WIDE selection (csrc/sweep_wide.hip), 40 of them — spread over every query tile — compared with the oracle."""
    rows, qs = corpus
    ids, sc, cnt = index.search_batch_brute_force(qs, k)
    assert index.last_select_level() == 4 and index.last_kernels() & va.KERNEL_SELECT_BF16, "the WIDE selection did not serve the call"
    assert index.last_split_stats()[1] <= 4, "unproven queries on the benchmark data: the bound or the list capacity has drifted"
    assert np.all(cnt == k)
    sample = np.arange(5, 1024, 26)[:40]
    eid, esc = po.scan_topk(po.COSINE, rows, qs[sample], k, po.MODE_M, nthreads=po.host_threads())
    assert np.array_equal(ids[sample], eid), f"ids / ranks differ from the oracle (mode M) at 1M x 1024 queries, k = {k}"
    assert np.array_equal(bits(sc[sample]), bits(esc)), f"score bits differ from the oracle (mode M) at k = {k}"
    # the first 10 of a k = 50 answer are the k = 10 answer (another kernel instance, another pool: the same exact scores)
    ids10, sc10, _ = index.search_batch_brute_force(qs[:256], K)
    assert np.array_equal(ids10, ids[:256, :K]) and np.array_equal(bits(sc10), bits(sc[:256, :K]))


def test_hnsw_1m_vs_oracle(corpus, index, tmp_path, record_property):
    """configs[2] at full size: batched GPU build of the 1 M-node graph (M 32, ef_construction 400), 1 024 queries at
    ef = 128 through the traversal kernel, compared with the oracle searching the SAME graph."""
    rows, qs = corpus
    ncores = po.host_threads()
    index.build_graph(0)
    nl, ml, ep = index.graph_info()
    assert index.node_count() == N and 0 <= ep < N and ml < nl
    nq, ef = 1024, 128
    res = index.search_batch_parallel(qs[:nq], K, SQ.Custom(ef))
    nd_gpu, ne_gpu = index.last_search_stats()
    index.save(str(tmp_path), "native_hnsw")
    og = po.NativeHnsw.file_load(str(tmp_path), "native_hnsw", po.COSINE, po.MODE_C)
    oi, od, oc, nd, ne = og.search_batch(qs[:nq], K, ef, po.TIE_CANONICAL, nthreads=ncores)
    assert (nd_gpu, ne_gpu) == (nd, ne), "distance-evaluation / expansion counters differ from the oracle at 1M"
    gid = np.array([[r[0] for r in q] for q in res], dtype=np.uint64)
    gsc = np.array([[r[1] for r in q] for q in res], dtype=np.float32)
    assert np.all(oc == K) and gid.shape == (nq, K)
    assert np.array_equal(gid, oi), "traversal ids / ranks differ from the oracle (mode C) at 1M"
    one_minus = np.float32(1.0) - od  # transform_score for Cosine: clamp(1 - d, 0, 1) (backend_adapter.rs:160-168)
    osim = np.minimum(np.maximum(one_minus, np.float32(0.0)), np.float32(1.0)).astype(np.float32)
    assert np.array_equal(bits(gsc), bits(osim)), "traversal score bits differ from the oracle (mode C) at 1M"
    # calls of at most one query per CU over a corpus beyond the Infinity Cache take the latency-mode kernel (speculative row
    # fetch beside the visited test, 1 024-thread blocks): the same ids, score bits and counters, query by query
    for lo, cnt in ((0, 1), (1, 5), (6, 16), (22, 200)):
        small = index.search_batch_parallel(qs[lo:lo + cnt], K, SQ.Custom(ef))
        nd_s, ne_s = index.last_search_stats()
        sid = np.array([[r[0] for r in q] for q in small], dtype=np.uint64)
        ssc = np.array([[r[1] for r in q] for q in small], dtype=np.float32)
        assert np.array_equal(sid, gid[lo:lo + cnt]) and np.array_equal(bits(ssc), bits(gsc[lo:lo + cnt])), (lo, cnt)
        _, _, _, nd_o, ne_o = og.search_batch(qs[lo:lo + cnt], K, ef, po.TIE_CANONICAL, nthreads=ncores)
        assert (nd_s, ne_s) == (nd_o, ne_o), "latency-mode counters differ from the oracle"
    del og
    # the reference's own arithmetic (mode R, reference heap / tie order) over the same graph: how often does a sub-ulp
    # difference in a distance change the id list?  
    ogr = po.NativeHnsw.file_load(str(tmp_path), "native_hnsw", po.COSINE, po.MODE_R)
    ri, rd, rc, _, _ = ogr.search_batch(qs[:nq], K, ef, po.TIE_REFERENCE, nthreads=ncores)
    del ogr
    seq_diff = int(np.sum(np.any(gid != ri, axis=1)))
    set_diff = int(sum(set(gid[i].tolist()) != set(ri[i].tolist()) for i in range(nq)))
    record_property("queries", nq)
    record_property("id_lists_differing_from_mode_R", seq_diff)
    record_property("id_sets_differing_from_mode_R", set_diff)
    print(f"\n[1M HNSW] queries whose id list differs from mode R: {seq_diff}/{nq} (as sets: {set_diff}/{nq})")
    rsim = np.minimum(np.maximum(np.float32(1.0) - rd, np.float32(0.0)), np.float32(1.0))
    same = np.all(gid == ri, axis=1)
    assert np.all(np.abs(gsc[same] - rsim[same]) <= 1e-5 * np.maximum(np.abs(rsim[same]), 1e-3))
    # measured: 0 of 1 024 (profiles/r02a_pytest_headline_sizes.log); pinned there with a small allowance — a drift of the GPU's
    # arithmetic away from the reference's would show up here first
    assert seq_diff <= 2, f"{seq_diff} of {nq} queries change their id list under the reference's summation order (measured: 0)"


def test_configs3_full_size_10m_bf16_vs_oracle(gpu_required):
    """BASELINE configs[3] at FULL size: 10 000 000 x 768 bf16, 1 024 queries per batch, k = 10 — the launch bench.py quotes
    `bf16_gemm` on (seed sweep + two launches of sweep_topk_gemm_bf16_glds + merges, select_stage.hip brute_bf16_dev).  The corpus is
    generated chunk-wise on the device exactly as bench.py does and uploaded chunk by chunk; every chunk is also copied to the
    host once, where the oracle (half_precision.rs:199-255 semantics) scans it for 32 sampled queries and the per-chunk lists are
    merged — the same scan an index over all 10 M rows would get.  Compared by the rule of tests/test_gpu_bf16.py: same ids
    wherever neighbouring oracle scores are further apart than the tolerance, scores within 2e-5 of the oracle's (two f32
    summation orders), the k-th score not below the oracle's."""
    torch = pytest.importorskip("torch")
    BR, BQ, chunk, nsample = 10_000_000, 1024, 1_000_000, 32
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(45)
    stream = torch.cuda.current_stream().cuda_stream
    ix = va.HnswIndex(D, DM.Cosine, va.HnswParams(16, 100, BR))
    ix.enable_bf16()
    gq = torch.Generator(device=dev)
    gq.manual_seed(46)
    qs = torch.randn((BQ, D), generator=gq, device=dev).cpu().numpy()
    sample = np.unique(np.concatenate([[0, 255, 256, 1023], np.random.default_rng(7).integers(0, BQ, nsample)]))
    best_i = np.empty((len(sample), 0), np.int64)
    best_s = np.empty((len(sample), 0), np.float32)
    nt = po.host_threads()
    for base in range(0, BR, chunk):
        c = torch.randn((chunk, D), generator=g, device=dev)
        torch.cuda.synchronize()
        ix.upload_dev(base, c.data_ptr(), chunk, stream)
        torch.cuda.synchronize()
        host = c.cpu().numpy()
        del c
        ei, es = po.scan_topk_bf16(po.COSINE, host, qs[sample], K, nthreads=nt)
        del host
        best_i = np.concatenate([best_i, ei.astype(np.int64) + base], axis=1)
        best_s = np.concatenate([best_s, es], axis=1)
        order = np.lexsort((best_i, -best_s.astype(np.float64)), axis=1)[:, :K]   # score descending, row ascending
        best_i = np.take_along_axis(best_i, order, axis=1)
        best_s = np.take_along_axis(best_s, order, axis=1)
    gi, gs, gc = ix.search_batch_brute_force_bf16(qs, K)
    assert ix.last_kernels() & va.KERNEL_GEMM_BF16_GLDS, "sweep_topk_gemm_bf16_glds did not serve the 10 M batch"
    assert np.all(gc == K)
    tol = 2e-5
    differing = 0
    for j, qi in enumerate(sample):
        g_i, g_s = gi[qi].astype(np.int64), gs[qi].astype(np.float64)
        e_i, e_s = best_i[j], best_s[j].astype(np.float64)
        assert np.all(np.diff(g_s) <= 1e-12) and len(set(g_i.tolist())) == K
        assert g_s[-1] >= e_s[-1] - tol, (qi,)
        for r in range(K):
            hit = np.nonzero(e_i == g_i[r])[0]
            if len(hit):
                assert abs(g_s[r] - e_s[hit[0]]) <= tol, (qi, r)      # the same row: the same score within two summation orders
            else:
                assert g_s[r] >= e_s[-1] - tol, (qi, r)                 # another row: only from inside a near-tie at the cut
            if g_i[r] != e_i[r]:
                differing += 1
                assert abs(e_s[r] - g_s[r]) <= tol, (qi, r)             # a different row at this rank: a near-tie of scores
    assert differing <= 2 * len(sample) // 10 + 2, differing               # (measured: 0)
    ix.close()


def test_configs4_shard_size_6p25m_f32_vs_oracle(gpu_required):
    """BASELINE configs[4] is 50 M x 768 f32 range-sharded over 8 GPUs: every GPU answers the exact sweep over ITS 6 250 000 rows
    (19.2 GB of f32 rows — 4.8e9 elements, beyond 32-bit element offsets — plus their 9.6 GB bf16 selection image), then the merge of
    tests/test_gpu_sharded.py / test_sharded_cpu.py.  This is one shard at full size on one GPU: 1 024 cosine queries, k = 10, through
    the default path (selection on the bf16 matrix cores, exact re-scoring, proof; HnswIndex::search_brute_force,
    index/hnsw/index/search.rs:176-219).  The rows are generated chunk-wise on the device as bench.py's sharded leg does; every chunk
    is copied to the host once, where the oracle scans it for 36 sampled queries, and the per-chunk lists are merged in the canonical
    order (score descending, row ascending) — the scan an index over all rows would get.  Bar: ids AND score bits equal."""
    torch = pytest.importorskip("torch")
    SR, BQ, chunk, nsample = 6_250_000, 1024, 1_000_000, 32
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    stream = torch.cuda.current_stream().cuda_stream
    ix = va.HnswIndex(D, DM.Cosine, va.HnswParams(16, 100, SR))
    gq = torch.Generator(device=dev)
    gq.manual_seed(47)
    qs = torch.randn((BQ, D), generator=gq, device=dev).cpu().numpy()
    sample = np.unique(np.concatenate([[0, 255, 256, 1023], np.random.default_rng(8).integers(0, BQ, nsample)]))
    mode = po.MODE_M if ix.sweep_arith_mode(K) == "M" else po.MODE_C
    best_i = np.empty((len(sample), 0), np.int64)
    best_s = np.empty((len(sample), 0), np.float32)
    nt = po.host_threads()
    for base in range(0, SR, chunk):
        n_c = min(chunk, SR - base)
        c = torch.randn((n_c, D), generator=g, device=dev)
        torch.cuda.synchronize()
        ix.upload_dev(base, c.data_ptr(), n_c, stream)
        torch.cuda.synchronize()
        host = c.cpu().numpy()
        del c
        ei, es = po.scan_topk(po.COSINE, host, qs[sample], K, mode, nthreads=nt)
        del host
        best_i = np.concatenate([best_i, ei.astype(np.int64) + base], axis=1)
        best_s = np.concatenate([best_s, es], axis=1)
        order = np.lexsort((best_i, -best_s.astype(np.float64)), axis=1)[:, :K]   # score descending, row ascending
        best_i = np.take_along_axis(best_i, order, axis=1)
        best_s = np.take_along_axis(best_s, order, axis=1)
    assert ix.len() == SR
    gi, gs, gc = ix.search_batch_brute_force(qs, K)
    assert ix.last_select_level() == 4, "the selection stage did not serve the 6.25 M-row batch"
    assert np.all(gc == K)
    assert np.array_equal(gi[sample].astype(np.int64), best_i), "ids / ranks differ from the oracle's scan of the same rows"
    assert np.array_equal(bits(gs[sample]), bits(best_s)), "score bits differ from the oracle's"
    # the other exact kernels over the same 4.8e9 elements, same bits: the GEMM-structured exact f32 kernel (selection switched off,
    # 72 queries), the small-batch streaming kernel (4 queries) and a single query
    ix.set_option(va.OPT_SELECTOR_LEVEL, 0)
    q72 = np.concatenate([qs[sample], qs[sample]])
    i72, s72, _ = ix.search_batch_brute_force(q72, K)
    ix.set_option(va.OPT_SELECTOR_LEVEL, -1)
    assert ix.last_select_level() == 0
    both = np.concatenate([best_i, best_i]), np.concatenate([best_s, best_s])
    assert np.array_equal(i72.astype(np.int64), both[0]) and np.array_equal(bits(s72), bits(both[1]))
    i4, s4, _ = ix.search_batch_brute_force(qs[sample[:4]], K)
    assert np.array_equal(i4.astype(np.int64), best_i[:4]) and np.array_equal(bits(s4), bits(best_s[:4]))
    i1, s1, _ = ix.search_batch_brute_force(qs[sample[5]], K)
    assert np.array_equal(i1.astype(np.int64)[0], best_i[5]) and np.array_equal(bits(s1)[0], bits(best_s)[5])
    # every query, not only the sampled ones: ordered, distinct, inside the shard
    assert np.all(np.diff(gs.astype(np.float64), axis=1) <= 0) and np.all(gi < SR)
    assert all(len(set(r.tolist())) == K for r in gi)
    # the last rows of the shard (beyond 2^32 elements) are reachable: a query equal to one of them finds it first
    tail_rows = [SR - 1, SR - 257, 5_600_000]
    g2 = torch.Generator(device=dev)
    g2.manual_seed(4242)
    probe = np.empty((len(tail_rows), D), np.float32)
    for base in range(0, SR, chunk):  # regenerate the stream: the same chunks, only the wanted rows kept
        n_c = min(chunk, SR - base)
        c = torch.randn((n_c, D), generator=g2, device=dev)
        for j, r in enumerate(tail_rows):
            if base <= r < base + n_c:
                probe[j] = c[r - base].cpu().numpy()
        del c
    pi, ps, _ = ix.search_batch_brute_force(np.repeat(probe, 6, axis=0), K)   # 18 queries: the selection stage again (>= 16)
    assert [int(pi[6 * j, 0]) for j in range(len(tail_rows))] == tail_rows
    assert np.all(np.abs(ps[::6, 0] - 1.0) < 1e-6)
    ix.close()
