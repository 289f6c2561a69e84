"""The combining front of the host-pointer search entry points (csrc/search_front.hip): the reference's calling pattern — many
threads, ONE query per `VectorIndex::search` call under a read lock (index/hnsw/index/search.rs:80; the stress tests
index/hnsw/native/tests.rs:264-416; velesdb-server/src/handlers/search.rs:34-73) — must return, from any number of threads and in
whatever batch a call lands, the bits the same query gets alone and in one big batch:

  * 32 native threads (tools/callers_bench.cpp over the C ABI; Python threads would measure the interpreter lock) x one query per
    call on a graph and on the exact sweep: every result compared bit for bit with a batched reference that is itself checked
    against the oracle; the front's counters show that calls really shared launches;
  * calls of different shapes (k, ef, mode) in flight together land in different batches and keep their own answers;
  * combining switched off, one batch in flight, a fixed cap of 2 queries per batch: same bits;
  * an error inside a combined launch reaches every caller of that launch with its message;
  * searches keep running — and stay well-formed — while another thread inserts, and see the final state afterwards.
"""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
DM = va.DistanceMetric
SQ = va.SearchQuality
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def callers_lib():
    path = os.path.join(ROOT, "tools", "libcallers_bench.so")
    if not os.path.exists(path):
        pytest.fail(f"{path} is missing: __graft_entry__.build() compiles it")
    cb = C.CDLL(path)
    cb.callers_run.restype = C.c_int
    cb.callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int, C.c_double,
                               C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return cb


def run_callers(cb, ix, q, k, ef, mode, threads, ref, seconds=0.4, min_calls=30, per_call=1):
    out = np.zeros(8, dtype=np.float64)
    rid, rsc, rn = [np.ascontiguousarray(x) for x in ref]
    rc = cb.callers_run(ix._h, q.ctypes.data, q.shape[0], q.shape[1], k, ef, mode, threads, seconds, min_calls, per_call,
                        rid.ctypes.data, rsc.ctypes.data, rn.ctypes.data, out.ctypes.data)
    assert rc == 0
    return {"qps": out[0], "p50_us": out[1], "p99_us": out[2], "calls": int(out[4]), "mismatch": int(out[5]), "failed": int(out[6])}


@pytest.fixture(scope="module")
def graph_index(gpu_required):
    rng = np.random.default_rng(21)
    n, dim = 120_000, 768     # 368 MB of rows: beyond the Infinity Cache => the latency-mode walk for small calls, as at 1 M
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    qs = rng.standard_normal((512, dim), dtype=np.float32)
    ix = va.HnswIndex(dim, DM.Cosine, va.HnswParams(16, 100, n))
    ix.upload(np.arange(n), rows)
    ix.build_graph()
    yield ix, rows, qs
    ix.close()


def test_32_native_threads_one_query_per_call_bit_equal(graph_index, callers_lib):
    ix, rows, qs = graph_index
    k = 10
    # references: ONE batched call each (512 queries: launches alone, never combined), the exact one checked against the oracle
    ref_h = ix._search_raw(qs, k, 128, va.MODE_HNSW)
    ref_b = ix._search_raw(qs, k, 0, va.MODE_BRUTE)
    eid, esc = po.scan_topk(po.COSINE, rows, qs[:64], k, po.MODE_M if ix.sweep_arith_mode(k) == "M" else po.MODE_C, nthreads=po.host_threads())
    assert np.array_equal(ref_b[0][:64], eid) and np.array_equal(bits(ref_b[1][:64]), bits(esc))
    # the same walk one query at a time from one thread (a lone caller: its own leader every time)
    one = [ix._search_raw(qs[i:i + 1], k, 128, va.MODE_HNSW) for i in range(16)]
    for i, (a, b, c) in enumerate(one):
        assert c[0] == ref_h[2][i] and np.array_equal(a[0], ref_h[0][i]) and np.array_equal(bits(b[0]), bits(ref_h[1][i]))
    s0 = ix.combine_stats()
    assert s0[1] == s0[0], "a lone caller's calls never share a launch"
    for mode, ef, ref in ((va.MODE_HNSW, 128, ref_h), (va.MODE_BRUTE, 0, ref_b)):
        before = ix.combine_stats()
        r = run_callers(callers_lib, ix, qs, k, ef, mode, 32, ref)
        after = ix.combine_stats()
        launches, calls = after[0] - before[0], after[1] - before[1]
        print(f"\n[callers] mode {mode}: 32 threads {r['qps']:.0f} q/s, p50 {r['p50_us']:.0f} us, p99 {r['p99_us']:.0f} us; "
              f"{calls} calls in {launches} launches ({calls / max(launches, 1):.1f} per launch, largest batch {after[3]})")
        assert r["failed"] == 0 and r["mismatch"] == 0, r
        assert calls == r["calls"] and launches * 4 < calls, (launches, calls)   # callers really shared launches
    # several queries per call (search_batch_parallel-sized calls from many threads) combine too
    r = run_callers(callers_lib, ix, qs, k, 128, va.MODE_HNSW, 8, ref_h, per_call=8)
    assert r["failed"] == 0 and r["mismatch"] == 0, r


def test_front_settings_do_not_change_bits(graph_index, callers_lib):
    ix, rows, qs = graph_index
    k = 5
    ref = ix._search_raw(qs, k, 64, va.MODE_HNSW)
    try:
        for opt, val in ((va.OPT_COMBINE_MAX_BATCH, 0), (va.OPT_COMBINE_MAX_BATCH, 2), (va.OPT_COMBINE_INFLIGHT, 1), (va.OPT_COMBINE_INFLIGHT, 4),
                         (va.OPT_COMBINE_WINDOW_US, 0), (va.OPT_COMBINE_WINDOW_US, 300)):
            ix.set_option(opt, val)
            assert ix.get_option(opt) == val
            before = ix.combine_stats()
            r = run_callers(callers_lib, ix, qs, k, 64, va.MODE_HNSW, 12, ref, seconds=0.2, min_calls=10)
            after = ix.combine_stats()
            assert r["failed"] == 0 and r["mismatch"] == 0, (opt, val, r)
            if opt == va.OPT_COMBINE_MAX_BATCH and val == 0:
                assert after[:3] == before[:3], "combining off: the front must not see the calls"
            if opt == va.OPT_COMBINE_MAX_BATCH and val == 2:
                assert after[3] <= max(before[3], 2) and (after[2] - before[2]) <= 2 * (after[0] - before[0])
            ix.set_option(opt, -1)
    finally:
        for opt in (va.OPT_COMBINE_MAX_BATCH, va.OPT_COMBINE_INFLIGHT, va.OPT_COMBINE_WINDOW_US):
            ix.set_option(opt, -1)
    assert ix.get_option(va.OPT_COMBINE_MAX_BATCH) == 256 and ix.get_option(va.OPT_COMBINE_INFLIGHT) == 0
    with pytest.raises(va.VelesHipError):
        ix.set_option(va.OPT_COMBINE_INFLIGHT, 9)
    with pytest.raises(va.VelesHipError):
        ix.set_option(va.OPT_COMBINE_MAX_BATCH, 5000)


def test_calls_of_different_shapes_keep_their_own_answers(graph_index):
    ix, rows, qs = graph_index
    shapes = [(10, 128, va.MODE_HNSW), (3, 64, va.MODE_HNSW), (10, 0, va.MODE_BRUTE), (1, 0, va.MODE_BRUTE), (7, 200, va.MODE_AUTO)]
    refs = [ix._search_raw(qs[:96], k, ef, mode) for k, ef, mode in shapes]
    errors = []

    def worker(t):
        try:
            for it in range(24):
                s = (t + it) % len(shapes)
                k, ef, mode = shapes[s]
                i = (t * 24 + it) % 96
                ids, sc, cnt = ix._search_raw(qs[i:i + 1], k, ef, mode)
                rid, rsc, rn = refs[s]
                assert cnt[0] == rn[i] and np.array_equal(ids[0, :cnt[0]], rid[i, :cnt[0]]) and \
                    np.array_equal(bits(sc[0, :cnt[0]]), bits(rsc[i, :cnt[0]])), (s, i)
            # rerank calls carry their own shape too
            r1 = ix.search_with_rerank(qs[t], 5, 20)
            assert len(r1) == 5
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    want = ix.search_with_rerank(qs[3], 5, 20)
    assert want == ix.search_with_rerank(qs[3], 5, 20)


def test_an_error_in_a_combined_launch_reaches_every_caller(gpu_required):
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((300, 32)).astype(np.float32)
    ix = va.HnswIndex(32, DM.Euclidean, va.HnswParams(8, 40, 400))
    ix.upload(np.arange(300), rows)     # rows without a graph: HNSW modes answer VDB_ERR_STATE
    ix.set_option(va.OPT_COMBINE_WINDOW_US, 2000)   # (let the eight calls meet)
    codes, msgs, ok = [], [], []
    start = threading.Barrier(8)

    def worker(t):
        start.wait()
        try:
            ix._search_raw(rows[t:t + 1], 3, 32, va.MODE_HNSW)
            codes.append(0)
        except va.VelesHipError as e:
            codes.append(e.code)
            msgs.append(str(e))
        ids, sc, cnt = ix._search_raw(rows[t:t + 1], 3, 0, va.MODE_BRUTE)   # the handle keeps serving
        ok.append(int(ids[0, 0]) == t and cnt[0] == 3)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert codes == [-8] * 8, codes
    assert all("graph" in m for m in msgs) and len(msgs) == 8, msgs[:2]
    assert all(ok) and len(ok) == 8
    ix.close()


def test_callers_while_another_thread_inserts(gpu_required):
    """Searches (one query per call, 12 threads) while one thread inserts one row at a time — the writer is not starved by the
    readers that keep arriving, every answer is well-formed, rows become visible, and the final state equals the oracle."""
    rng = np.random.default_rng(13)
    n0, n_add, dim, k = 4000, 300, 64, 10
    rows = rng.standard_normal((n0 + n_add, dim)).astype(np.float32)
    qs = rng.standard_normal((64, dim)).astype(np.float32)
    ix = va.HnswIndex(dim, DM.Euclidean, va.HnswParams(8, 60, n0 + n_add))
    ix.insert_batch_parallel([(i, rows[i]) for i in range(n0)], 512)
    errors, done, seen_new = [], threading.Event(), [0]

    def searcher(t):
        try:
            it = 0
            while not done.is_set() or it < 5:
                i = (t * 5 + it) % 64
                if it % 3 == 2:
                    ids, sc, cnt = ix._search_raw(qs[i:i + 1], k, 0, va.MODE_BRUTE)
                else:
                    ids, sc, cnt = ix._search_raw(qs[i:i + 1], k, 64, va.MODE_HNSW)
                assert cnt[0] == k and np.all(np.diff(sc[0]) >= 0) and ids.max() < n0 + n_add, (ids, sc, cnt)
                if ids.max() >= n0:
                    seen_new[0] += 1
                it += 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def inserter():
        try:
            for i in range(n0, n0 + n_add):
                ix.insert(i, rows[i])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
        finally:
            done.set()

    th = [threading.Thread(target=searcher, args=(t,)) for t in range(12)] + [threading.Thread(target=inserter)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]
    assert ix.len() == n0 + n_add
    ids, sc, _ = ix.search_batch_brute_force(qs[:8], k)
    eid, esc = po.scan_topk(po.EUCLIDEAN, rows, qs[:8], k, po.MODE_C)
    assert np.array_equal(ids, eid) and np.array_equal(bits(sc), bits(esc))
    # ... and through the front, one query per call
    for i in range(4):
        a, b, c = ix._search_raw(qs[i:i + 1], k, 0, va.MODE_BRUTE)
        assert np.array_equal(a[0], eid[i]) and np.array_equal(bits(b[0]), bits(esc[i]))
    ix.close()


def test_first_use_image_build_on_a_foreign_stream_races_a_host_search(gpu_required):
    """a first-use image build enqueued on a CALLER's stream by the device-resident entry point must be
    complete before another search context may take the image over.  Thread A's first selection batch arrives on its own torch
    stream while thread B searches through the host entry point: both must equal the oracle."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(17)
    n, dim, k = 70_000, 256, 10
    rows = rng.standard_normal((n, dim), dtype=np.float32)
    qs = rng.standard_normal((128, dim), dtype=np.float32)
    eid, esc = po.scan_topk(po.COSINE, rows, qs, k, po.MODE_M, nthreads=po.host_threads())
    for attempt in range(3):
        ix = va.HnswIndex(dim, DM.Cosine, va.HnswParams(8, 40, n))
        ix.upload(np.arange(n), rows)
        dev = torch.device("cuda", 0)
        dq = torch.from_numpy(qs).to(dev)
        d_i = torch.empty((128, k), dtype=torch.int64, device=dev)
        d_s = torch.empty((128, k), dtype=torch.float32, device=dev)
        d_n = torch.empty((128,), dtype=torch.int32, device=dev)
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        res, errors = {}, []
        go = threading.Barrier(2)

        def dev_thread():
            try:
                go.wait()
                ix.search_batch_dev(dq.data_ptr(), 128, k, 0, va.MODE_BRUTE, d_i.data_ptr(), d_s.data_ptr(), d_n.data_ptr(), side.cuda_stream)
                side.synchronize()
                res["dev"] = (d_i.cpu().numpy().astype(np.uint64), d_s.cpu().numpy())
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        def host_thread():
            try:
                go.wait()
                res["host"] = ix._search_raw(qs[:100], k, 0, va.MODE_BRUTE)   # 100 queries: launches alone, selection stage
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        th = [threading.Thread(target=dev_thread), threading.Thread(target=host_thread)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
        assert np.array_equal(res["dev"][0], eid) and np.array_equal(bits(res["dev"][1]), bits(esc)), attempt
        assert np.array_equal(res["host"][0], eid[:100]) and np.array_equal(bits(res["host"][1]), bits(esc[:100])), attempt
        ix.close()
