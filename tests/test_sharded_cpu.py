"""world_size-2 (and 4 / 8) gloo tests (CPU) of the one-process-per-GPU composition: range-sharded exact search = per-shard top-k +
ONE all-gather of packed 12-byte records + merge must equal the exact top-k over the whole corpus, ties included.

WHAT IS UNDER TEST.  The record layout, its sentinels, the selection key and the rank rule of the merge are the PRODUCT's own text:
`velesdb_amd/csrc/vdb_shard_wire.hpp` — the inline functions `shard_group.hip`'s kernels (`pack_shard_records`, `merge_shards_topk`)
are written over — compiled for the host by `tests/shard_wire_model.cpp` (g++; `wire_pack_all` = the pack kernel's loop,
`wire_merge_all` = the merge kernel's body, query after query).  Every case runs that model AND the oracle's independent restatement
(`vo_merge_shard_records`: a stable sort of the concatenation) and holds the two to each other and to the exact answer;
`velesdb_amd/sharded.py:pack_records` (the launcher-side restatement of the record) is held to the model's bytes, so a drift of the wire
format between the Python side and the kernels is caught here, without a GPU.  What this file does NOT run: the kernels themselves, the
RCCL transport, the per-shard sweep (supplied by the oracle here; the collective is gloo's) — `tests/test_gpu_sharded.py` runs those on
the GPU (co-located shards, the loop-back stand-in of tests/stub_rccl, world-1 RCCL), and no run on more than one GPU exists."""
import ctypes as C
import os
import socket
import subprocess
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pyoracle as po
from velesdb_amd.sharded import RECORD_DTYPE, pack_records, query_slice

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_MODEL_SO = os.path.join(tempfile.gettempdir(), "vdb_shard_wire_model_%d" % os.getuid(), "libshard_wire_model.so")


def wire_model():
    """The product's wire format + merge rule on the host (tests/shard_wire_model.cpp over csrc/vdb_shard_wire.hpp), built on first use."""
    src = os.path.join(ROOT, "tests", "shard_wire_model.cpp")
    hdr = os.path.join(ROOT, "velesdb_amd", "csrc", "vdb_shard_wire.hpp")
    if not os.path.exists(_MODEL_SO) or os.path.getmtime(_MODEL_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(_MODEL_SO), exist_ok=True)
        tmp = _MODEL_SO + ".%d" % os.getpid()
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-I",
                               os.path.join(ROOT, "velesdb_amd", "csrc"), "-o", tmp, src])
        os.replace(tmp, _MODEL_SO)
    L = C.CDLL(_MODEL_SO)
    vp = C.c_void_p
    L.wire_pack_all.restype, L.wire_pack_all.argtypes = None, [vp, vp, vp, vp, C.c_uint32, C.c_uint32]
    L.wire_merge_all.restype, L.wire_merge_all.argtypes = None, [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, vp, vp, vp]
    L.wire_record_bytes.restype = C.c_uint32
    return L


def product_pack(ids, scores, counts):
    """pack_shard_records' loop (the product's text) -> [nq][k][3] u32"""
    ids = np.ascontiguousarray(ids, dtype=np.uint64)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    nq, k = ids.shape
    rec = np.empty((nq, k, 3), dtype=np.uint32)
    wire_model().wire_pack_all(ids.ctypes.data, scores.ctypes.data, counts.ctypes.data, rec.ctypes.data, nq, k)
    return rec


def product_merge(allrec, k, hib):
    """merge_shards_topk's body (the product's text) over [S][nq][k][3] records"""
    allrec = np.ascontiguousarray(allrec, dtype=np.uint32)
    S, nq = allrec.shape[0], allrec.shape[1]
    oi = np.empty((nq, k), dtype=np.uint64)
    osc = np.empty((nq, k), dtype=np.float32)
    on = np.empty(nq, dtype=np.uint32)
    wire_model().wire_merge_all(allrec.ctypes.data, S, nq, k, 1 if hib else 0, oi.ctypes.data, osc.ctypes.data, on.ctypes.data)
    return oi, osc, on


def same_merge(a, b):
    """two merge results agree: counts, and ids + score bits in front of the count (behind it both hold the filler)"""
    (ai, asc, an), (bi, bsc, bn) = a, b
    if not np.array_equal(np.asarray(an, dtype=np.uint32), np.asarray(bn, dtype=np.uint32)):
        return False
    return all(np.array_equal(ai[q, :n], bi[q, :n]) and np.array_equal(asc[q, :n].view(np.uint32), bsc[q, :n].view(np.uint32))
               for q, n in enumerate(np.asarray(an, dtype=np.int64)))


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
        # the launcher-side restatement == the product's packing (vdb_shard_wire.hpp through the host model), byte for byte
        assert rec.tobytes() == product_pack(ids, sc, np.full(nq, kk, dtype=np.uint32)).tobytes()
        # ONE all-gather of nq * k * 12 bytes per rank
        mine = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy())
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        allrec = np.stack([g.numpy().view(np.uint32).reshape(nq, k, 3) for g in gathered])
        gi, gs, gc = product_merge(allrec, k, hib)                      # the product's merge rule ...
        assert same_merge((gi, gs, gc), po.merge_shard_records(allrec, k, hib))  # ... and the oracle's independent statement of it
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
    # (a shard's list arrives best first in the metric's direction — the product's rank rule counts on it, the oracle's sort does not)
    allrec_hib = np.stack([rec([(11, 2.0), (10, 1.0)], k), rec([(22, 3.0), (20, 1.0), (21, 1.0)], k), rec([], k)])
    for merge in (po.merge_shard_records, product_merge):   # the oracle's statement and the product's text
        ids, sc, cnt = merge(allrec, k, False)
        assert ids[0].tolist() == [10, 20, 21, 11] and cnt[0] == 4
        ids, sc, cnt = merge(allrec_hib, k, True)
        assert ids[0].tolist() == [22, 11, 10, 20]
        ids, sc, cnt = merge(allrec[[0, 2]], k, False)
        assert cnt[0] == 2 and ids[0, :2].tolist() == [10, 11] and ids[0, 2] == np.uint64(0xFFFFFFFFFFFFFFFF)


def test_wire_format_python_side_equals_the_products_text():
    """velesdb_amd/sharded.py:pack_records against pack_shard_records' loop (vdb_shard_wire.hpp, host model): every byte, for full lists,
    short lists, empty lists, ids beyond 32 bits, NaN / inf / signed-zero scores, and the overflow marker of a device-resident call
    (count 0xFFFFFFFF: first record's score bits 0xFFFFFFFE, everything else the empty sentinel) — which also survives the merge."""
    assert wire_model().wire_record_bytes() == RECORD_DTYPE.itemsize == 12
    rng = np.random.default_rng(5)
    for nq, k in ((1, 1), (3, 10), (17, 10), (64, 100), (5, 7)):
        ids = rng.integers(0, 2**63, size=(nq, k), dtype=np.uint64) * np.uint64(2) + np.uint64(1)
        sc = rng.standard_normal((nq, k)).astype(np.float32)
        sc.reshape(-1)[::5] = np.float32("nan")
        sc.reshape(-1)[1::7] = np.float32("-inf")
        sc.reshape(-1)[2::11] = np.float32(-0.0)
        counts = rng.integers(0, k + 1, size=nq).astype(np.uint32)
        counts[0] = k
        if nq > 2:
            counts[1] = 0
            counts[2] = 0xFFFFFFFF   # overflow marker
        a = pack_records(ids, sc, counts).view(np.uint32).reshape(nq, k, 3)
        b = product_pack(ids, sc, counts)
        assert a.tobytes() == b.tobytes(), (nq, k)
        if nq > 2:
            assert b[2, 0].tolist() == [0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFE] and (k == 1 or b[2, 1].tolist() == [0xFFFFFFFF] * 3)
            _, _, cnt = product_merge(np.stack([b, b]), k, True)
            assert cnt[2] == 0xFFFFFFFF and cnt[1] == 0 and cnt[0] == k


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
        gi, gs, gc = product_merge(np.stack(allrec), k, hib)   # the product's rule (vdb_shard_wire.hpp) ...
        assert same_merge((gi, gs, gc), po.merge_shard_records(np.stack(allrec), k, hib))   # ... == the oracle's stable sort
        concat = [p for pairs in lists for p in pairs]
        exp = po.sort_results(metric, concat)[:k]
        assert int(gc[0]) == len(exp) == min(k, len(concat))
        assert gi[0, :len(exp)].tolist() == [a for a, _ in exp]
        assert gs[0, :len(exp)].view(np.uint32).tolist() == np.array([b for _, b in exp], dtype=np.float32).view(np.uint32).tolist()

    prop()
