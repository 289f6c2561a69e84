"""GPU parity tests of the selection stage for k beyond 10 (csrc/sweep_wide.hip + the WIDE instance of the selection kernel):
HnswIndex::search_brute_force over a batch (index/hnsw/index/search.rs:176-219) with k = 11 ... 128 — the reference benches
k = 10 / 50 / 100 (benches/hnsw_benchmark.rs:152-159) and its Perfect / rerank calls take any k (search.rs:118-160).

Bar, as for every exact path: ids, ranks and score BITS of the exact kernels (oracle mode M), whatever the batch size and whichever
kernel answered; `last_select_level() == 4` says the WIDE selection did.  Adversarial data (duplicates = exact ties by the
thousand, NaN / inf / zero rows, soft deletes, clusters tighter than the error bound) must come out exact too — through the proof
by construction or through the gathered exact pass of the queries the selection gave up."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM = va.DistanceMetric
PO = {DM.Cosine: po.COSINE, DM.DotProduct: po.DOT, DM.Euclidean: po.EUCLIDEAN}
LEVEL_WIDE = 4


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def check(ix, metric, rows, qs, k, level=LEVEL_WIDE, alive=None):
    ids, sc, cnt = ix.search_batch_brute_force(qs, k)
    if level is not None:
        assert ix.last_select_level() == level, f"select level {ix.last_select_level()} served k = {k}, expected {level}"
    mode = po.MODE_M if ix.sweep_arith_mode(k) == "M" else po.MODE_C
    if alive is None:
        eid, esc = po.scan_topk(PO[metric], rows, qs, k, mode, nthreads=po.host_threads())
    else:
        keep = np.nonzero(alive)[0]
        eid, esc = po.scan_topk(PO[metric], rows[keep], qs, k, mode, nthreads=po.host_threads())
        eid = keep[eid.astype(np.int64)].astype(np.uint64)
    kk = eid.shape[1]
    assert np.all(cnt == kk)
    assert np.array_equal(ids[:, :kk], eid), f"ids / ranks differ from the oracle at k = {k}"
    assert np.array_equal(bits(sc[:, :kk]), bits(esc)), f"score bits differ from the oracle at k = {k}"
    return ids, sc


@pytest.fixture(scope="module")
def corpus(gpu_required):
    rng = np.random.default_rng(2026)
    rows = rng.standard_normal((70_000, 768), dtype=np.float32)
    qs = rng.standard_normal((300, 768), dtype=np.float32)
    return rows, qs


@pytest.mark.parametrize("metric", [DM.Cosine, DM.DotProduct])
def test_wide_k_vs_oracle(corpus, metric):
    rows, qs = corpus
    ix = va.HnswIndex(768, metric)
    ix.upload(np.arange(len(rows), dtype=np.uint64), rows)
    for k, nq in ((11, 96), (50, 300), (100, 64), (128, 17)):
        check(ix, metric, rows, qs[:nq], k)
    # the same queries in another batch: the same bits (a query's answer does not depend on its neighbours in the batch)
    a, sa = ix.search_batch_brute_force(qs[:40], 50)[:2]
    b, sb = ix.search_batch_brute_force(qs[:300], 50)[:2]
    assert np.array_equal(a, b[:40]) and np.array_equal(bits(sa), bits(sb[:40]))
    # k <= 10 takes the WIDE selection too (selector level 3, the default); pinned at level 2 it keeps the block-local lists; k beyond
    # the WIDE limit: the exact kernels — the same oracle every way
    check(ix, metric, rows, qs[:96], 10, level=LEVEL_WIDE)
    check(ix, metric, rows, qs[:96], 1, level=LEVEL_WIDE)
    ix.set_option(va.OPT_SELECTOR_LEVEL, 2)
    check(ix, metric, rows, qs[:96], 10, level=2)
    check(ix, metric, rows, qs[:96], 50, level=LEVEL_WIDE)
    ix.set_option(va.OPT_SELECTOR_LEVEL, -1)
    check(ix, metric, rows, qs[:32], 129, level=0)
    # the selector switched off: the exact kernels answer k = 50 with the same bits
    ix.set_option(va.OPT_SELECTOR_LEVEL, 0)
    c, sc_ = ix.search_batch_brute_force(qs[:96], 50)[:2]
    assert ix.last_select_level() == 0
    ix.set_option(va.OPT_SELECTOR_LEVEL, -1)
    d, sd = ix.search_batch_brute_force(qs[:96], 50)[:2]
    assert np.array_equal(c, d) and np.array_equal(bits(sc_), bits(sd))
    ix.close()


def test_wide_k_full_query_tiles_and_other_dims(gpu_required):
    rng = np.random.default_rng(7)
    for dim, n, nq, k in ((128, 90_000, 1024, 50), (256, 66_000, 513, 100), (1024, 66_000, 80, 64)):
        rows = rng.standard_normal((n, dim), dtype=np.float32)
        qs = rng.standard_normal((nq, dim), dtype=np.float32)
        ix = va.HnswIndex(dim, DM.Cosine)
        ix.upload(np.arange(n, dtype=np.uint64), rows)
        check(ix, DM.Cosine, rows, qs, k)
        ix.close()


def test_wide_k_adversarial_rows(gpu_required):
    """duplicates (exact ties across the cut), a tight cluster (thousands of rows inside the error bound: the list overflows and the
    gathered exact pass answers), zero / NaN / inf rows, soft-deleted rows — the answer is the oracle's in every case."""
    rng = np.random.default_rng(99)
    n, dim, k = 80_000, 768, 50
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    rows[1000:1400] = rows[1000]                       # 400 copies of one row: exact ties, ids ascending inside the group
    centre = rng.standard_normal(dim).astype(np.float32)
    rows[20_000:26_000] = centre + 1e-3 * rng.standard_normal((6000, dim)).astype(np.float32)   # a cluster tighter than delta
    rows[5] = 0.0
    rows[6, 3] = np.inf
    rows[7, 9] = np.nan
    qs = rng.standard_normal((64, dim), dtype=np.float32)
    qs[0] = rows[1000]          # its best 400 rows tie exactly
    qs[1] = centre              # thousands of rows within the bound of each other
    qs[2] = rows[7]             # a NaN query
    ix = va.HnswIndex(dim, DM.Cosine)
    ix.upload(np.arange(n, dtype=np.uint64), rows)
    ids, sc, cnt = ix.search_batch_brute_force(qs, k)
    assert ix.last_select_level() == LEVEL_WIDE
    mode = po.MODE_M if ix.sweep_arith_mode(k) == "M" else po.MODE_C
    eid, esc = po.scan_topk(po.COSINE, rows, qs, k, mode, nthreads=po.host_threads())
    assert np.array_equal(ids, eid) and np.array_equal(bits(sc), bits(esc))
    nq_l, unproven = ix.last_split_stats()
    assert nq_l == 64 and 1 <= unproven <= 8, f"{unproven} unproven queries (expected: the cluster's and the non-finite ones)"
    # soft deletes: the best rows of query 3 disappear from the answer
    gone = eid[3, :20].astype(np.int64)
    for r in gone:
        ix.remove(int(r))
    alive = np.ones(n, dtype=bool)
    alive[gone] = False
    check(ix, DM.Cosine, rows, qs[3:40], k, alive=alive)
    ix.close()


def test_wide_k_parks_itself_when_the_data_defeats_it(gpu_required):
    """a corpus that is one tight cluster: every list overflows; after such a batch the handle answers the next batches with k > 10
    from the exact kernels (64 of them), like levels 1 / 2 do — results identical throughout."""
    rng = np.random.default_rng(5)
    n, dim, k = 66_000, 256, 20
    centre = rng.standard_normal(dim).astype(np.float32)
    rows = centre + 1e-4 * rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((32, dim), dtype=np.float32)
    ix = va.HnswIndex(dim, DM.Cosine)
    ix.upload(np.arange(n, dtype=np.uint64), rows)
    mode = po.MODE_M if ix.sweep_arith_mode(k) == "M" else po.MODE_C
    eid, esc = po.scan_topk(po.COSINE, rows, qs, k, mode, nthreads=po.host_threads())
    levels = []
    for _ in range(3):
        ids, sc, _ = ix.search_batch_brute_force(qs, k)
        levels.append(ix.last_select_level())
        assert np.array_equal(ids, eid) and np.array_equal(bits(sc), bits(esc))
    assert levels[0] == LEVEL_WIDE and levels[-1] == 0, levels
    ix.close()


def test_wide_k_euclidean_vs_oracle(corpus):
    """Euclidean batches at k > 10 through the same WIDE selection over the augmented DotProduct form s = q.v - |v|^2 / 2
    (sweep_split.hip): candidates re-scored with the canonical (q - v)^2 chain (oracle mode C — the bits a single query gets), the
    proof checked per query in the squared-distance domain; near-duplicates of a query (distances far below the form's error bound)
    leave the proof to the gathered exact pass."""
    rows, qs = corpus
    rows = rows.copy()
    rows[500:540] = qs[5] + 1e-3 * np.random.default_rng(3).standard_normal((40, 768)).astype(np.float32)   # 40 near-copies of query 5
    ix = va.HnswIndex(768, DM.Euclidean)
    ix.upload(np.arange(len(rows), dtype=np.uint64), rows)
    for k, nq in ((11, 96), (50, 300), (100, 64), (128, 17)):
        ids, sc, cnt = ix.search_batch_brute_force(qs[:nq], k)
        assert ix.last_select_level() == LEVEL_WIDE, (k, ix.last_select_level())
        eid, esc = po.scan_topk(po.EUCLIDEAN, rows, qs[:nq], k, po.MODE_C, nthreads=po.host_threads())
        assert np.all(cnt == k)
        assert np.array_equal(ids, eid), f"Euclidean ids / ranks differ from the oracle (mode C) at k = {k}"
        assert np.array_equal(bits(sc), bits(esc)), f"Euclidean score bits differ from the oracle at k = {k}"
    nq_l, unproven = ix.last_split_stats()
    # one query alone (the canonical vector-ALU kernel) gives the same bits as the batch
    one, s1, _ = ix.search_batch_brute_force(qs[3:4], 50)
    many, sm, _ = ix.search_batch_brute_force(qs[:64], 50)
    assert np.array_equal(one[0], many[3]) and np.array_equal(bits(s1[0]), bits(sm[3]))
    ix.close()
    # other dims, a ragged corpus, a full query tile
    rng = np.random.default_rng(17)
    for dim, n, nq, k in ((128, 70_077, 256, 20), (256, 66_000, 300, 64)):
        r2 = rng.standard_normal((n, dim), dtype=np.float32)
        q2 = rng.standard_normal((nq, dim), dtype=np.float32)
        ix = va.HnswIndex(dim, DM.Euclidean)
        ix.upload(np.arange(n, dtype=np.uint64), r2)
        ids, sc, cnt = ix.search_batch_brute_force(q2, k)
        assert ix.last_select_level() == LEVEL_WIDE
        eid, esc = po.scan_topk(po.EUCLIDEAN, r2, q2, k, po.MODE_C, nthreads=po.host_threads())
        assert np.array_equal(ids, eid) and np.array_equal(bits(sc), bits(esc)), (dim, n, nq, k)
        ix.close()


@pytest.mark.parametrize("metric", [DM.Cosine, DM.DotProduct])
def test_wide_k_sq8_storage_mode_vs_oracle(corpus, metric):
    """StorageMode::SQ8 (core/quantization.rs:17-29): batches at k > 10 select over the dequantised bf16 image and re-score their
    candidates with the reference's asymmetric chain over the codes (dot_product_quantized_simd / cosine_similarity_quantized_simd,
    quantization.rs:410-554) — bit-identical to the oracle's restatement of that scalar code, as at k <= 10."""
    rows, qs = corpus
    ix = va.HnswIndex(768, metric)
    ix.upload(np.arange(len(rows), dtype=np.uint64), rows)
    ix.set_storage_mode(va.StorageMode.SQ8)
    for k, nq in ((11, 40), (50, 300), (100, 64)):
        ids, sc, cnt = ix.search_batch_sq8(qs[:nq], k)
        assert ix.last_select_level() == LEVEL_WIDE, (k, ix.last_select_level())
        eid, esc = po.scan_topk_sq8(PO[metric], rows, qs[:nq], k, nthreads=po.host_threads())
        assert np.all(cnt == k)
        assert np.array_equal(ids, eid.astype(np.uint64)), f"SQ8 ids / ranks differ from the oracle at k = {k}"
        assert np.array_equal(bits(sc), bits(esc)), f"SQ8 score bits differ from the oracle at k = {k}"
    # k = 10 takes the WIDE selection too at selector level 3 (the default); pinned to level 2 it runs the block-local lists over
    # the same image (level 3 of the SQ8 mode) — the same bits either way; a handful of queries: the exact SQ8 sweep
    w10 = ix.search_batch_sq8(qs[:64], 10)
    assert ix.last_select_level() == (LEVEL_WIDE if os.environ.get("VELESDB_WIDE_SMALL_K") != "0" else 3)
    va.set_split_selector(2)
    l10 = ix.search_batch_sq8(qs[:64], 10)
    assert ix.last_select_level() == 3
    va.set_split_selector(3)
    assert np.array_equal(w10[0], l10[0]) and np.array_equal(bits(w10[1]), bits(l10[1]))
    e10i, e10s = po.scan_topk_sq8(PO[metric], rows, qs[:64], 10, nthreads=po.host_threads())
    assert np.array_equal(w10[0], e10i.astype(np.uint64)) and np.array_equal(bits(w10[1]), bits(e10s))
    a, sa, _ = ix.search_batch_sq8(qs[:3], 50)
    b, sb, _ = ix.search_batch_sq8(qs[:300], 50)
    assert np.array_equal(a, b[:3]) and np.array_equal(bits(sa), bits(sb[:3]))
    ix.close()
