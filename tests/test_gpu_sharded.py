"""GPU tests of the multi-device handle and the one-process-per-GPU shard group (BASELINE configs[4]),
all through the C ABI:

  * VDB_SHARD_RANGE handle: per-shard sweep -> pack -> exchange -> merge_shards_topk == the oracle on the whole corpus
    AND == the single-device index bit for bit (ids, ranks, score bits, counts), for every metric, with exact ties that
    straddle shards, shards shorter than k, removals, duplicate ids, single inserts.  On a one-GPU box the shards are
    co-located (`devices=[0, 0, 0]`: device-to-device exchange); with several GPUs the same test runs over distinct
    devices = RCCL (ncclCommInitAll / ncclAllGather).
  * VDB_SHARD_REPLICA handle: identical graphs on every replica, query batch split, results == single-device index.
  * per-shard graphs + merge on a range handle == merging the results of independent single-device indexes.
  * test_sharded_rccl: `world = torch.cuda.device_count()` processes (1 on the driver's box, 8 on a full node), nccl
    process group, every rank a different row range of ONE corpus through search_batch_dev, merged ids / scores on every
    rank == the oracle on the whole corpus — the configs[4] path end to end (vdb_hip_index_join_group,
    ncclAllGather of 12-byte records, merge kernel).
"""
import os
import socket

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM = va.DistanceMetric
SQ = va.SearchQuality
PO_METRIC = {DM.Cosine: po.COSINE, DM.Euclidean: po.EUCLIDEAN, DM.DotProduct: po.DOT, DM.Hamming: po.HAMMING,
             DM.Jaccard: po.JACCARD}


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def shard_devices(n):
    """n shards over the visible GPUs: distinct devices when there are enough (RCCL), else co-located on device 0"""
    nd = va.device_count()
    return list(range(n)) if nd >= n else [0] * n


def rand_rows(rng, n, d, metric):
    if metric in (DM.Hamming, DM.Jaccard):
        return (rng.random((n, d)) > 0.6915).astype(np.float32)
    return rng.standard_normal((n, d)).astype(np.float32)


def oracle_mode(ix, k):
    return po.MODE_M if ix.sweep_arith_mode(k) == "M" else po.MODE_C


@pytest.mark.parametrize("metric", [DM.Cosine, DM.Euclidean, DM.DotProduct, DM.Hamming, DM.Jaccard])
@pytest.mark.parametrize("n,dim,shards,nq", [(5000, 96, 3, 70), (900, 768, 2, 5), (23, 16, 4, 3)])
def test_range_handle_equals_oracle_and_single_device(gpu_required, metric, n, dim, shards, nq):
    rng = np.random.default_rng(n + dim + int(metric))
    rows = rand_rows(rng, n, dim, metric)
    qs = rand_rows(rng, nq, dim, metric)
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(7) + np.uint64(1 << 40))  # external ids far from the row numbers
    k = 10
    # max_elements fixes the rows per shard: make the last shard short (n = 23, 4 shards: 6 + 6 + 6 + 5 rows < k)
    sh = va.HnswIndex(dim, metric, va.HnswParams(8, 50, n), devices=shard_devices(shards), shard_mode=va.SHARD_RANGE)
    one = va.HnswIndex(dim, metric, va.HnswParams(8, 50, n))
    info = sh.shard_info()
    assert info["n_shards"] == shards and info["shard_mode"] == va.SHARD_RANGE
    assert info["transport"] == ("rccl" if va.device_count() >= shards else "d2d-copy")
    half = n // 2
    assert sh.upload(ids[:half], rows[:half]) == half and one.upload(ids[:half], rows[:half]) == half
    # second batch crosses shard boundaries and repeats three ids (skipped once, by the handle)
    ids2 = np.concatenate([ids[half:], ids[:3]])
    rows2 = np.concatenate([rows[half:], rows[:3]])
    assert sh.upload(ids2, rows2) == n - half and one.upload(ids2, rows2) == n - half
    assert sh.len() == n == one.len() and sh.node_count() == n
    got = sh.search_batch_brute_force(qs, k)
    ref = one.search_batch_brute_force(qs, k)
    eid, esc = po.scan_topk(PO_METRIC[metric], rows, qs, min(k, n), oracle_mode(one, k))
    for g in (got, ref):
        assert np.array_equal(g[0][:, :eid.shape[1]], ids[eid.astype(np.int64)]), "ids / ranks differ from the oracle"
        assert np.array_equal(bits(g[1][:, :esc.shape[1]]), bits(esc))
        assert np.all(g[2] == min(k, n))
    # soft delete on the handle: gone from the results, like the single-device index (search.rs:86-91)
    victims = [int(got[0][0, 0]), int(got[0][0, min(k, n) - 1])]
    victims.append(int(next(i for i in ids[::-1] if int(i) not in victims)))
    for v in victims:
        assert sh.remove(v) and one.remove(v)
    assert not sh.remove(victims[0])
    a, b = sh.search_batch_brute_force(qs, k), one.search_batch_brute_force(qs, k)
    assert np.array_equal(a[2], b[2])
    for q in range(nq):
        c = int(a[2][q])
        assert np.array_equal(a[0][q, :c], b[0][q, :c]) and np.array_equal(bits(a[1][q, :c]), bits(b[1][q, :c]))
        assert not set(victims) & set(a[0][q, :c].tolist())
    assert sh.len() == n - 3 and sh.tombstone_count() == 3
    sh.close()
    one.close()


def test_range_handle_hamming_ties_straddle_shards(gpu_required):
    # 4 distinct rows repeated: every distance is shared by hundreds of rows in EVERY shard; the merged order must be the
    # global insertion order (sort_results is a stable sort, core/distance.rs:95-103)
    rng = np.random.default_rng(5)
    base = (rng.random((4, 64)) > 0.5).astype(np.float32)
    rows = base[rng.integers(0, 4, 3000)]
    qs = (rng.random((9, 64)) > 0.5).astype(np.float32)
    sh = va.HnswIndex(64, DM.Hamming, va.HnswParams(8, 50, 3000), devices=shard_devices(3), shard_mode=va.SHARD_RANGE)
    sh.upload(np.arange(3000), rows)
    ids, sc, cnt = sh.search_batch_brute_force(qs, 25)
    eid, esc = po.scan_topk(po.HAMMING, rows, qs, 25, po.MODE_C)
    assert np.array_equal(ids, eid) and np.array_equal(bits(sc), bits(esc))
    sh.close()


def test_range_handle_other_exact_modes_equal_single_device(gpu_required):
    # bf16 GEMM distance, SQ8 and sign-bit scans go through the same pack / exchange / merge
    rng = np.random.default_rng(9)
    n, dim, k = 6000, 128, 10
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((80, dim)).astype(np.float32)
    sh = va.HnswIndex(dim, DM.Cosine, va.HnswParams(8, 50, n), devices=shard_devices(2), shard_mode=va.SHARD_RANGE)
    one = va.HnswIndex(dim, DM.Cosine, va.HnswParams(8, 50, n))
    for ix in (sh, one):
        ix.upload(np.arange(n), rows)
    for setup, fn in ((lambda ix: ix.enable_bf16(), "search_batch_brute_force_bf16"),
                      (lambda ix: ix.set_storage_mode(va.StorageMode.SQ8), "search_batch_sq8"),
                      (lambda ix: ix.set_storage_mode(va.StorageMode.Binary), "search_batch_binary")):
        setup(sh)
        setup(one)
        a, b = getattr(sh, fn)(qs, k), getattr(one, fn)(qs, k)
        assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2]), fn
    sh.close()
    one.close()


def test_range_handle_inserts_and_per_shard_graphs(gpu_required):
    # single inserts route to the shard of the next global row; HNSW modes search one graph per shard and merge: equal to
    # merging the results of independent single-device indexes built over the same row ranges (deterministic builds)
    rng = np.random.default_rng(12)
    n, dim, k, ef = 1200, 64, 10, 64
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((20, dim)).astype(np.float32)
    sh = va.HnswIndex(dim, DM.Euclidean, va.HnswParams(8, 60, n), devices=shard_devices(2), shard_mode=va.SHARD_RANGE)
    parts = [va.HnswIndex(dim, DM.Euclidean, va.HnswParams(8, 60, n // 2)) for _ in range(2)]
    for i in range(n):
        sh.insert(i, rows[i])
        parts[i // (n // 2)].insert(i, rows[i])
    sh.insert(5, rows[5])  # duplicate id: ignored (trait_impl.rs:23-25)
    assert sh.len() == n
    got = sh.search_batch_parallel(qs, k, SQ.Custom(ef))
    for qi in range(len(qs)):
        cand = [r for p in parts for r in p.search_batch_parallel(qs[qi:qi + 1], k, SQ.Custom(ef))[0]]
        cand.sort(key=lambda r: r[1])  # stable: shard order on equal distances
        assert got[qi] == cand[:k]
    # exact search over the same handle still equals the oracle
    ids, sc, cnt = sh.search_batch_brute_force(qs, k)
    eid, esc = po.scan_topk(po.EUCLIDEAN, rows, qs, k, po.MODE_C)
    assert np.array_equal(ids, eid) and np.array_equal(bits(sc), bits(esc))
    sh.close()
    for p in parts:
        p.close()


def test_replica_handle_splits_queries(gpu_required):
    rng = np.random.default_rng(21)
    n, dim, k = 1500, 96, 10
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((37, dim)).astype(np.float32)
    rep = va.HnswIndex(dim, DM.Cosine, va.HnswParams(8, 60, n), devices=shard_devices(2), shard_mode=va.SHARD_REPLICA)
    one = va.HnswIndex(dim, DM.Cosine, va.HnswParams(8, 60, n))
    items = [(i, rows[i]) for i in range(n)]
    assert rep.insert_batch_parallel(items, 256) == n and one.insert_batch_parallel(items, 256) == n
    assert rep.shard_info()["transport"] == "none"
    assert rep.graph_info() == one.graph_info()
    for node in (0, 17, n - 1):
        assert rep.neighbors(0, node) == one.neighbors(0, node)
    assert rep.search_batch_parallel(qs, k, SQ.Custom(64)) == one.search_batch_parallel(qs, k, SQ.Custom(64))
    a, b = rep.search_batch_brute_force(qs, k), one.search_batch_brute_force(qs, k)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    assert rep.search(qs[0], k) == one.search(qs[0], k)
    rep.close()
    one.close()


def test_range_handle_device_pointer_entry(gpu_required):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(33)
    n, dim, k, nq = 4000, 128, 10, 66
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    qs = rng.standard_normal((nq, dim)).astype(np.float32)
    sh = va.HnswIndex(dim, DM.DotProduct, va.HnswParams(8, 50, n), devices=shard_devices(2), shard_mode=va.SHARD_RANGE)
    sh.upload(np.arange(n), rows)
    dq = torch.from_numpy(qs).cuda()
    d_ids = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    d_sc = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    d_n = torch.empty((nq,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    sh.search_batch_dev(dq.data_ptr(), nq, k, 0, va.MODE_BRUTE, d_ids.data_ptr(), d_sc.data_ptr(), d_n.data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    eid, esc = po.scan_topk(po.DOT, rows, qs, k, oracle_mode(sh, k))
    assert np.array_equal(d_ids.cpu().numpy().astype(np.uint64), eid) and np.array_equal(bits(d_sc.cpu().numpy()), bits(esc))
    assert np.all(d_n.cpu().numpy() == k)
    sh.close()


# ---------------------------------------------------------------------------------------------------------------
# one process per GPU over RCCL
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _corpus_chunk(seed, c, rows, dim, hamming):
    rng = np.random.default_rng(seed * 1000 + c)
    if hamming:
        return (rng.random((rows, dim)) > 0.6915).astype(np.float32)
    return rng.standard_normal((rows, dim), dtype=np.float32)


def _rccl_worker(rank, world, port, cfg, out):
    import torch
    import torch.distributed as dist
    import velesdb_amd as va_
    from oracle import pyoracle as po_
    from velesdb_amd.sharded import join_process_group
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        ok = True
        for (metric, n, dim, chunk_rows, nq, k, seed) in cfg:
            hamming = metric == int(va_.DistanceMetric.Hamming)
            nchunks = n // chunk_rows
            per = nchunks // world  # chunks per rank (the last rank takes the remainder)
            c_lo = rank * per
            c_hi = nchunks if rank == world - 1 else c_lo + per
            ix = va_.HnswIndex(dim, va_.DistanceMetric(metric), va_.HnswParams(8, 50, (c_hi - c_lo) * chunk_rows), device=rank)
            for c in range(c_lo, c_hi):
                ix.upload(np.arange(c * chunk_rows, (c + 1) * chunk_rows, dtype=np.uint64), _corpus_chunk(seed, c, chunk_rows, dim, hamming))
            join_process_group(ix, rank, world, dev)
            info = ix.shard_info()
            ok &= info["world"] == world and info["rank"] == rank and info["transport"] == "rccl"
            qs = _corpus_chunk(seed, 10_000, nq, dim, hamming)
            dq = torch.from_numpy(qs).to(dev)
            d_ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
            d_sc = torch.empty((nq, k), dtype=torch.float32, device=dev)
            d_n = torch.empty((nq,), dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(2):  # twice: the gather buffer is reused
                ix.search_batch_dev(dq.data_ptr(), nq, k, 0, va_.MODE_BRUTE, d_ids.data_ptr(), d_sc.data_ptr(), d_n.data_ptr(), st)
            torch.cuda.synchronize()
            hid, hsc, hn = ix.search_batch_brute_force(qs, k)  # host entry point: same collective, same result
            gi, gs = d_ids.cpu().numpy().astype(np.uint64), d_sc.cpu().numpy()
            ok &= bool(np.array_equal(gi, hid) and np.array_equal(gs.view(np.uint32), hsc.view(np.uint32)))
            if rank == 0:  # the checker: the oracle over the WHOLE corpus
                rows = np.concatenate([_corpus_chunk(seed, c, chunk_rows, dim, hamming) for c in range(nchunks)])
                pm = {0: po_.COSINE, 1: po_.EUCLIDEAN, 2: po_.DOT, 3: po_.HAMMING, 4: po_.JACCARD}[metric]
                mode = po_.MODE_M if ix.sweep_arith_mode(k) == "M" else po_.MODE_C
                eid, esc = po_.scan_topk(pm, rows, qs, k, mode, nthreads=po_.host_threads())
                ok &= bool(np.array_equal(gi, eid) and np.array_equal(gs.view(np.uint32), esc.view(np.uint32)))
                ok &= bool(np.all(d_n.cpu().numpy() == k))
            # every rank holds the same merged result
            t = d_ids.clone()
            dist.broadcast(t, src=0)
            ok &= bool(torch.equal(t, d_ids))
            ix.close()
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_sharded_rccl(gpu_required):
    """BASELINE configs[4] end to end at a size the oracle checks in seconds: ONE 1 M x 768 cosine corpus cut into
    `world` row ranges (world = visible GPUs), 128 queries (the GEMM kernel) — plus a Hamming corpus whose ties straddle
    the ranks."""
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    world = torch.cuda.device_count()
    assert world >= 1
    cfg = [(int(DM.Cosine), 1_000_000, 768, 125_000, 128, 10, 7),
           (int(DM.Hamming), 64_000, 128, 8_000, 40, 20, 8)]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rccl_worker, args=(world, _free_port(), cfg, out), nprocs=world, join=True)
    assert all(out.get(r) for r in range(world)), dict(out)


_LOOPBACK_SCRIPT = r'''
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.environ["VDB_TEST_ROOT"])
from velesdb_amd import _ffi as _f
_f.use_library(_f.PROBE_LIB_PATH)   # the test hooks below (VELESDB_RCCL_LIB, VELESDB_SHARD_FORCE_COLLECTIVE) exist in the probe build only
import velesdb_amd as va
from oracle import pyoracle as po
DM = va.DistanceMetric
bits = lambda a: np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
stub = ctypes.CDLL(os.environ["VELESDB_RCCL_LIB"])   # the same object the product dlopen()s: shared counters
for metric, pm, n, dim, shards, nq, k in [(DM.Cosine, po.COSINE, 5000, 96, 3, 70, 10), (DM.Hamming, po.HAMMING, 3000, 64, 4, 9, 25),
                                          (DM.Euclidean, po.EUCLIDEAN, 23, 16, 4, 3, 10)]:
    rng = np.random.default_rng(n)
    if metric == DM.Hamming:
        rows = ((rng.random((4, dim)) > 0.5).astype(np.float32))[rng.integers(0, 4, n)]   # ties in every shard
        qs = (rng.random((nq, dim)) > 0.5).astype(np.float32)
    else:
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        qs = rng.standard_normal((nq, dim)).astype(np.float32)
    sh = va.HnswIndex(dim, metric, va.HnswParams(8, 50, n), devices=[0] * shards, shard_mode=va.SHARD_RANGE)
    one = va.HnswIndex(dim, metric, va.HnswParams(8, 50, n))
    assert sh.shard_info()["transport"] == "rccl", sh.shard_info()      # forced: the collective branch
    sh.upload(np.arange(n), rows)
    one.upload(np.arange(n), rows)
    before = stub.velesdb_stub_allgathers()
    for rep in range(2):                                                # twice: gather buffers and communicators are reused
        a, b = sh.search_batch_brute_force(qs, k), one.search_batch_brute_force(qs, k)
        assert np.array_equal(a[2], b[2])
        kk = min(k, n)
        assert np.array_equal(a[0][:, :kk], b[0][:, :kk]) and np.array_equal(bits(a[1][:, :kk]), bits(b[1][:, :kk]))
    mode = po.MODE_M if one.sweep_arith_mode(k) == "M" else po.MODE_C
    eid, esc = po.scan_topk(pm, rows, qs, min(k, n), mode)
    assert np.array_equal(a[0][:, :eid.shape[1]], eid) and np.array_equal(bits(a[1][:, :esc.shape[1]]), bits(esc))
    assert stub.velesdb_stub_allgathers() - before == 2 * shards        # ONE grouped all-gather per shard and search
    sh.close(); one.close()
assert stub.velesdb_stub_init_all_calls() == 3                          # ncclCommInitAll once per handle
# the one-process-per-GPU path over the same transport (world = 1)
ix = va.HnswIndex(32, DM.Cosine)
rows = np.random.default_rng(1).standard_normal((500, 32)).astype(np.float32)
ix.upload(np.arange(500), rows)
ref = ix.search_batch_brute_force(rows[:7], 5)
ix.join_group(va.comm_unique_id(), 0, 1)
got = ix.search_batch_brute_force(rows[:7], 5)
assert np.array_equal(got[0], ref[0]) and np.array_equal(bits(got[1]), bits(ref[1]))
print("LOOPBACK-OK")
'''


def test_collective_branch_over_a_loopback_transport(gpu_required):
    """The in-process multi-device branch (ncclCommInitAll + one grouped ncclAllGather per shard + merge,
    shard_group.hip ensure_group_comms / group_exchange_merge) needs shards on DISTINCT devices — which a one-GPU box
    never has.  Here it runs anyway: VELESDB_SHARD_FORCE_COLLECTIVE=1 sends co-located shards down that branch and
    VELESDB_RCCL_LIB binds tests/stub_rccl (a loop-back all-gather with RCCL's signatures) instead of librccl."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = os.path.join(root, "tests", "stub_rccl", "libstub_rccl.so")
    assert os.path.exists(stub), "tests/stub_rccl/libstub_rccl.so is missing: __graft_entry__.build() compiles it"
    env = dict(os.environ, VELESDB_RCCL_LIB=stub, VELESDB_SHARD_FORCE_COLLECTIVE="1", VDB_TEST_ROOT=root)
    r = subprocess.run([sys.executable, "-c", _LOOPBACK_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "LOOPBACK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_distinct_devices_without_rccl_fail_loudly(gpu_required):
    """>= 2 GPUs but no loadable RCCL: a range-sharded search must fail with VDB_ERR_UNSUPPORTED and name the library — not
    crash, not silently fall back to copies.  (On a one-GPU box the same is forced through the collective hook.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["VDB_TEST_ROOT"])
from velesdb_amd import _ffi as _f
_f.use_library(_f.PROBE_LIB_PATH)   # the test hooks below (VELESDB_RCCL_LIB, VELESDB_SHARD_FORCE_COLLECTIVE) exist in the probe build only
import velesdb_amd as va
sh = va.HnswIndex(16, va.DistanceMetric.Cosine, va.HnswParams(8, 50, 100), devices=[0, 0], shard_mode=va.SHARD_RANGE)
sh.upload(np.arange(100), np.random.default_rng(0).standard_normal((100, 16)).astype(np.float32))
try:
    sh.search_batch_brute_force(np.ones((2, 16), np.float32), 3)
    print("NO-ERROR")
except va.VelesHipError as e:
    print("CODE", e.code, str(e))
try:
    va.comm_unique_id()
    print("NO-ERROR")
except va.VelesHipError as e:
    print("CODE", e.code, str(e))
'''
    env = dict(os.environ, VELESDB_RCCL_LIB="/nonexistent/librccl-missing.so", VELESDB_SHARD_FORCE_COLLECTIVE="1", VDB_TEST_ROOT=root)
    r = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith(("CODE", "NO-ERROR"))]
    assert len(lines) == 2 and all(l.startswith("CODE -7") and "librccl-missing" in l for l in lines), r.stdout
