"""world_size-2 (and 4 / 8) gloo tests (CPU) of the one-process-per-GPU composition: what the launcher side owns
(velesdb_amd/sharded.py: the wire record of the all-gather, the query split of replica mode) plus the merge rule the HIP
kernel implements (restated in the oracle, vo_merge_shard_records).  Range-sharded exact search = per-shard top-k + ONE
all-gather of packed 12-byte records + merge must equal the exact top-k over the whole corpus, ties included.  The
per-shard top-k (the GPU sweep in production) is supplied here by the oracle and the collective is gloo's, so the test
exercises exactly the N>1 logic that needs no GPU; tests/test_gpu_sharded.py runs the real thing (RCCL + merge kernel)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyoracle as po
from velesdb_amd.sharded import RECORD_DTYPE, pack_records, query_slice


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, metric, hib, n, dim, nq, k, seed, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(seed)
        if metric == po.HAMMING:
            rows = (rng.random((n, dim)) > 0.6915).astype(np.float32)   # integer distances: many exact ties
            qs = (rng.random((nq, dim)) > 0.6915).astype(np.float32)
        else:
            rows = rng.standard_normal((n, dim)).astype(np.float32)
            qs = rng.standard_normal((nq, dim)).astype(np.float32)
        # uneven shards: rank 0 gets 1/3, the last shard may hold fewer than k rows
        cuts = [0, n // 3, n] if world == 2 else np.linspace(0, n, world + 1).astype(int).tolist()
        lo, hi = cuts[rank], cuts[rank + 1]
        kk = min(k, hi - lo)
        lid, lsc = po.scan_topk(metric, rows[lo:hi], qs, kk, po.MODE_C)
        ids = np.zeros((nq, k), dtype=np.uint64)
        sc = np.zeros((nq, k), dtype=np.float32)
        ids[:, :kk] = lid + np.uint64(lo)   # external id = global row
        sc[:, :kk] = lsc
        rec = pack_records(ids, sc, np.full(nq, kk, dtype=np.uint32))
        assert rec.dtype == RECORD_DTYPE and rec.nbytes == nq * k * 12
        # ONE all-gather of nq * k * 12 bytes per rank
        mine = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy())
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        allrec = np.stack([g.numpy().view(np.uint32).reshape(nq, k, 3) for g in gathered])
        gi, gs, gc = po.merge_shard_records(allrec, k, hib)
        eid, esc = po.scan_topk(metric, rows, qs, min(k, n), po.MODE_C)
        ok = bool(np.array_equal(gi[:, :eid.shape[1]], eid)
                  and np.array_equal(gs[:, :esc.shape[1]].view(np.uint32), esc.view(np.uint32))
                  and int(gc.min()) == min(k, n) and int(gc.max()) == min(k, n))
        # replica mode: the slices of all ranks tile [0, nq)
        sl = torch.tensor(list(query_slice(nq, rank, world)), dtype=torch.int64)
        allsl = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allsl, sl)
        tiles = allsl[0][0].item() == 0 and allsl[-1][1].item() == nq and all(
            allsl[i][1].item() == allsl[i + 1][0].item() for i in range(world - 1))
        out[rank] = ok and tiles
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric,hib,n,dim,k", [
    (po.COSINE, True, 3000, 64, 10),
    (po.EUCLIDEAN, False, 2000, 48, 10),
    (po.HAMMING, False, 4000, 64, 10),   # ties across shards must come out in global-row order
    (po.DOT, True, 25, 16, 10),          # the first shard has 8 rows < k
])
def test_range_sharded_topk_world2(metric, hib, n, dim, k):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, metric, hib, n, dim, 17, k, 1234, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


@pytest.mark.parametrize("world,metric,hib,n,dim,k", [
    (4, po.COSINE, True, 3000, 64, 10),
    (8, po.HAMMING, False, 4000, 64, 10),   # BASELINE configs[4]'s shard count; integer distances tie across all eight shards
    (8, po.DOT, True, 25, 16, 10),          # every shard holds 3-4 rows < k: the merged list is assembled from eight short ones
])
def test_range_sharded_topk_world4_and_8(world, metric, hib, n, dim, k):
    """The same composition at the rank counts the scaling run uses (north_star: 1 / 2 / 4 / 8 GPUs)."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, metric, hib, n, dim, 17, k, 4321, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


def test_merge_rule_ties_and_short_lists():
    # equal scores: shard order, then position; empty slots skipped; fewer than k records in total
    def rec(entries, k):
        ids = np.zeros((1, k), dtype=np.uint64)
        sc = np.zeros((1, k), dtype=np.float32)
        for i, (a, b) in enumerate(entries):
            ids[0, i], sc[0, i] = a, b
        return pack_records(ids, sc, np.array([len(entries)], dtype=np.uint32)).view(np.uint32).reshape(1, k, 3)
    k = 4
    allrec = np.stack([rec([(10, 1.0), (11, 2.0)], k), rec([(20, 1.0), (21, 1.0), (22, 3.0)], k), rec([], k)])
    ids, sc, cnt = po.merge_shard_records(allrec, k, False)
    assert ids[0].tolist() == [10, 20, 21, 11] and cnt[0] == 4
    ids, sc, cnt = po.merge_shard_records(allrec, k, True)
    assert ids[0].tolist() == [22, 11, 10, 20]
    ids, sc, cnt = po.merge_shard_records(allrec[[0, 2]], k, False)
    assert cnt[0] == 2 and ids[0, :2].tolist() == [10, 11] and ids[0, 2] == np.uint64(0xFFFFFFFFFFFFFFFF)


def test_query_slice_properties():
    for nq in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            parts = [query_slice(nq, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == nq
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_merge_is_the_stable_sort_of_the_concatenation_property():
    """The size-independent property of the sharded path (hypothesis): for any number of shards, any k, any per-shard lists that are
    themselves in the metric's order — short lists, empty shards, heavy ties, infinities — the merged top-k is the first k of a
    STABLE sort of the shard-order concatenation (equal scores: shard order, then position = global row order), and its count is
    min(k, total).  This is DistanceMetric::sort_results (distance.rs:95-103) applied to what one big index would have sorted."""
    from hypothesis import given, settings, strategies as st

    scores = st.sampled_from([0.0, -0.0, 1.0, 1.0, 2.0, 2.5, -1.0, float("inf"), float("-inf"), 3.0])

    @settings(max_examples=300, deadline=None, derandomize=True, database=None)   # the same 300 layouts on every run
    @given(st.integers(1, 8), st.integers(1, 10), st.booleans(), st.data())
    def prop(world, k, hib, data):
        metric = po.COSINE if hib else po.EUCLIDEAN
        lists, next_id = [], 0
        for s in range(world):
            n = data.draw(st.integers(0, k))
            sc = [data.draw(scores) for _ in range(n)]
            pairs = po.sort_results(metric, [(next_id + i, x) for i, x in enumerate(sc)])   # a shard's own top-k is in the metric's order
            # ids inside a shard follow the shard's row order among equals (stable), shards own disjoint ascending id ranges
            lists.append(pairs)
            next_id += 1000
        allrec = []
        for pairs in lists:
            ids = np.zeros((1, k), dtype=np.uint64)
            sc = np.zeros((1, k), dtype=np.float32)
            for i, (a, b) in enumerate(pairs):
                ids[0, i], sc[0, i] = a, b
            allrec.append(pack_records(ids, sc, np.array([len(pairs)], dtype=np.uint32)).view(np.uint32).reshape(1, k, 3))
        gi, gs, gc = po.merge_shard_records(np.stack(allrec), k, hib)
        concat = [p for pairs in lists for p in pairs]
        exp = po.sort_results(metric, concat)[:k]
        assert int(gc[0]) == len(exp) == min(k, len(concat))
        assert gi[0, :len(exp)].tolist() == [a for a, _ in exp]
        assert gs[0, :len(exp)].view(np.uint32).tolist() == np.array([b for _, b in exp], dtype=np.float32).view(np.uint32).tolist()

    prop()
