#!/usr/bin/env python3
"""GPU fuzz of the exact-sweep kernels against the oracle (not part of the test-suite: random shapes and adversarial
data for a fixed wall-clock budget).  f32: ids and score bits must equal the oracle's (mode M / C as the library
reports); bf16: the tolerance check of tests/test_gpu_bf16.py.

    python tools/fuzz_sweep.py --seconds 240 --seed 1
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
from oracle import pyoracle as po  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--seconds", type=float, default=240)
p.add_argument("--seed", type=int, default=1)
p.add_argument("--big", action="store_true", help="few cases over 100K-400K rows (many row tiles / groups per wave and block)")
p.add_argument("--select", action="store_true", help="batches that take the selection stage (>= 224 queries, >= 65 536 rows, "
               "k <= 10; levels 1 and 2 at random) incl. data built to defeat the proofs (gathered / GEMM fallbacks, level parking)")
p.add_argument("--wide", action="store_true", help="round 6: --select's shapes and data at 10 < k <= 128 (the WIDE selection, sweep_wide.hip; Cosine / DotProduct / Euclidean)")
p.add_argument("--bf16-big", action="store_true", help="bf16 result batches that the 256 x 256 LDS-DMA kernel serves (>= 65 536 rows, "
               ">= 224 queries, k <= 10 — BASELINE configs[3]'s kernel), asserted through last_kernels()")
p.add_argument("--engine", type=int, default=1, help="0 = vector-ALU kernels for cosine / dot too")
p.add_argument("--only-it", type=int, default=0, help="replay: generate every case, run only this one (verbose)")
p.add_argument("--euclid", action="store_true", help="Euclidean only (matrix-core batches + exact re-scoring)")
p.add_argument("--bits", action="store_true", help="Hamming / Jaccard (packed-bit kernels) instead of cosine / dot")
p.add_argument("--bits-big", action="store_true", help="Hamming / Jaccard batches that take the four-bit GEMM path (>= 32 queries, >= 65 536 rows)")
a = p.parse_args()
rng = np.random.default_rng(a.seed)
NT = po.host_threads()
va.set_sweep_engine(a.engine)
DM = va.DistanceMetric


def make_rows(kind, n, dim, q0):
    if kind == "normal":
        return rng.standard_normal((n, dim)).astype(np.float32)
    if kind == "dups":       # few distinct rows: massive exact ties, candidate buffers overflow with equal keys
        base = rng.standard_normal((max(1, n // 50), dim)).astype(np.float32)
        return base[rng.integers(0, base.shape[0], n)]
    if kind == "ascending":  # every later row is better than all before it: every row tile floods the epilogue
        t = np.linspace(0.0, 1.0, n, dtype=np.float32)[:, None]
        noise = rng.standard_normal((n, dim)).astype(np.float32)
        return (t * 4.0) * q0[None, :] + noise * 0.05 + q0[None, :] * 0.01
    if kind == "small_ints":
        return rng.integers(-3, 4, size=(n, dim)).astype(np.float32)
    if kind == "zeros_mixed":
        r = rng.standard_normal((n, dim)).astype(np.float32)
        r[rng.random(n) < 0.2] = 0.0
        return r
    raise ValueError(kind)


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


from test_gpu_bf16 import check as bf16_check, check_sampled as bf16_check_sampled  # noqa: E402

t_end = time.time() + a.seconds
it = 0
stats = {"f32": 0, "bf16": 0}
while time.time() < t_end:
    it += 1
    bf16 = rng.random() < 0.4
    metric = [DM.Cosine, DM.DotProduct][int(rng.integers(0, 2))]
    if a.bits:
        bf16 = False
        metric = [DM.Hamming, DM.Jaccard][int(rng.integers(0, 2))]
    if a.euclid:
        bf16 = False
        metric = DM.Euclidean
    dim = int(rng.choice([64, 128, 192, 256, 768])) if bf16 else int(rng.choice([8, 17, 64, 100, 128, 256, 300, 768]))
    n = int(rng.choice([1, 7, 100, 129, 1000, 4097, 12000, 30000]))
    nq = int(rng.choice([1, 5, 63, 64, 100, 128, 129, 230, 256, 300, 480, 512, 700, 1024, 1100]))
    k = int(rng.choice([1, 3, 10, 16, 17, 32, 48, 64]))
    kind = str(rng.choice(["normal", "dups", "ascending", "small_ints", "zeros_mixed"]))
    if a.big:
        n = int(rng.choice([100_000, 250_000, 400_000]))
        dim = int(rng.choice([64, 256, 512, 768])) if not a.bits else int(rng.choice([64, 256, 768, 1000]))
        nq = int(rng.choice([1, 8, 24, 40, 64, 130, 300]))
        k = int(rng.choice([1, 10, 32]))
        kind = str(rng.choice(["normal", "normal", "dups", "small_ints"]))
    if a.select:
        bf16 = False
        n = int(rng.choice([66_000, 150_000, 300_000]))
        dim = int(rng.choice([64, 128, 256, 768]))
        nq = int(rng.choice([80, 100, 150, 230, 256, 300, 480, 620, 1000]))
        k = int(rng.choice([1, 3, 10]))
        kind = str(rng.choice(["normal", "normal", "dups", "small_ints", "ascending", "zeros_mixed", "clusters"]))
        va.set_split_selector(int(rng.choice([1, 2, 3])))
        if rng.random() < 0.3:
            metric = DM.Euclidean
    if a.wide:  # the WIDE selection: no block-local lists, candidate lists that can overflow (clusters, duplicates), gathered fallback
        bf16 = False
        metric = [DM.Cosine, DM.DotProduct][int(rng.integers(0, 2))]
        n = int(rng.choice([66_000, 70_077, 150_000, 300_000]))
        dim = int(rng.choice([128, 192, 256, 768]))
        nq = int(rng.choice([16, 17, 64, 100, 230, 256, 300, 513, 620, 1000, 1024]))
        k = int(rng.choice([11, 12, 20, 33, 50, 64, 65, 100, 127, 128]))
        kind = str(rng.choice(["normal", "normal", "dups", "small_ints", "ascending", "zeros_mixed", "clusters"]))
        if rng.random() < 0.3:
            metric = DM.Euclidean  # the augmented DotProduct form, canonical re-scoring, the proof checked per query
    if a.bits_big:
        a.bits = True
        bf16 = False
        metric = [DM.Hamming, DM.Jaccard][int(rng.integers(0, 2))]
        n = int(rng.choice([65_536, 66_000, 70_077, 150_000, 300_001]))
        dim = int(rng.choice([33, 64, 100, 256, 768, 1000]))
        nq = int(rng.choice([32, 33, 64, 100, 224, 256, 300, 480, 600, 700, 1024, 1100]))
        k = int(rng.choice([1, 3, 10]))
        kind = str(rng.choice(["normal", "normal", "dups", "zeros_mixed"]))
    if a.bf16_big:
        bf16 = True
        metric = [DM.Cosine, DM.DotProduct][int(rng.integers(0, 2))]
        n = int(rng.choice([66_000, 70_077, 150_000, 300_001, 600_000]))
        dim = int(rng.choice([128, 192, 256, 768]))
        nq = int(rng.choice([224, 256, 300, 480, 600, 700, 1024]))
        k = int(rng.choice([1, 3, 10]))
        kind = str(rng.choice(["normal", "normal", "dups", "small_ints", "zeros_mixed", "ascending"]))
    q0 = rng.standard_normal(dim).astype(np.float32)
    rows = make_rows("normal" if kind == "clusters" else kind, n, dim, q0)
    Q = rng.standard_normal((nq, dim)).astype(np.float32)
    if kind == "clusters":  # noisy copies of some queries, spreads from "inside every bound" to "inside level 2's only"
        spread_hi = float(rng.choice([1e-6, 1e-3, 0.15]))
        for j in rng.choice(nq, min(nq, 60), replace=False):
            m = int(rng.choice([12, 60, 200]))
            where = rng.choice(n, m, replace=False)
            sp = np.linspace(spread_hi / 3, spread_hi, m, dtype=np.float32)[:, None]
            rows[where] = Q[j] + sp * rng.standard_normal((m, dim)).astype(np.float32)
    if a.bits:  # 0/1 data with a random density (sparse rows: empty unions; dense rows: everything ties)
        dens = float(rng.choice([0.02, 0.3085, 0.5, 0.97]))
        rows = (rng.random((n, dim)) < dens).astype(np.float32)
        Q = (rng.random((nq, dim)) < dens).astype(np.float32)
        if kind == "dups":
            rows = rows[rng.integers(0, max(1, n // 50), n)]
        if kind == "zeros_mixed":
            rows[rng.random(n) < 0.2] = 0.0
            Q[rng.random(nq) < 0.2] = 0.0
    if kind == "ascending" and not a.bits:
        Q = (Q * 0.05 + q0[None, :]).astype(np.float32)
    if kind == "small_ints" and not a.bits:
        Q = rng.integers(-3, 4, size=(nq, dim)).astype(np.float32)
    if a.only_it and it != a.only_it:
        if it > a.only_it:
            break
        continue
    ix = va.HnswIndex(dim, metric)
    ix.upload(np.arange(n), rows)
    tag = f"it={it} bf16={bf16} {metric.name} n={n} dim={dim} nq={nq} k={k} {kind}"
    kk = min(k, n)
    if a.only_it:
        gi, gs, gc = ix.search_batch_brute_force(Q, k)
        mode = po.MODE_M if ix.sweep_arith_mode(k) == "M" else po.MODE_C
        eid, esc = po.scan_topk(int(metric), rows, Q, kk, mode, nthreads=NT)
        bad = np.nonzero(np.any(gi[:, :kk] != eid, axis=1) | np.any(bits(gs[:, :kk]) != bits(esc), axis=1))[0]
        print(tag, "mode", mode, "bad queries", bad.tolist()[:20], "of", nq)
        for qi in bad[:3]:
            d = np.nonzero(gi[qi, :kk] != eid[qi])[0]
            print(" q", qi, "first diff rank", d[:5], "gpu", gi[qi, d[:5]], gs[qi, d[:5]], "oracle", eid[qi, d[:5]], esc[qi, d[:5]])
            one = ix.search_batch_brute_force(Q[qi:qi + 1], k)
            print("   alone equal to oracle:", np.array_equal(one[0][0, :kk], eid[qi]), " count", gc[qi], one[2][0])
        sys.exit(0)
    if bf16 and a.bf16_big:
        ix.enable_bf16()
        gi, gs, gc = ix.search_batch_brute_force_bf16(Q, k)
        assert ix.last_kernels() & va.KERNEL_GEMM_BF16_GLDS, tag + " (not served by sweep_topk_gemm_bf16_glds)"
        pm = po.COSINE if metric == DM.Cosine else po.DOT
        sample = np.unique(np.concatenate([[0, nq - 1, 255 % nq, 256 % nq], rng.integers(0, nq, 12)]))
        if kind == "small_ints":  # exact products: ids, ranks (ties by row) and score bits
            eid, esc = po.scan_topk_bf16(pm, rows, Q[sample], kk, nthreads=NT)
            assert np.array_equal(gi[sample][:, :kk], eid) and np.array_equal(bits(gs[sample][:, :kk]), bits(esc)), tag
        elif kind in ("dups", "zeros_mixed", "ascending"):  # tie groups / near-ties everywhere: values only
            rr64 = po.round_bf16(rows).astype(np.float64)
            for qi in sample:
                q64 = po.round_bf16(Q[qi]).astype(np.float64)
                full = rr64 @ q64
                scale = np.linalg.norm(rr64, axis=1) * np.linalg.norm(q64)
                if metric == DM.Cosine:
                    with np.errstate(invalid="ignore", divide="ignore"):
                        full = np.where(scale >= 1.1920929e-7 ** 1, full / np.where(scale > 0, scale, 1.0), 0.0)
                    tol = 1e-5 * np.ones_like(full)
                else:
                    tol = 1e-5 * np.maximum(scale, 1e-30)
                g_i, g_s = gi[qi, :kk].astype(np.int64), gs[qi, :kk].astype(np.float64)
                assert np.all(np.abs(g_s - full[g_i]) <= tol[g_i] + 1e-12), tag
                kth = np.partition(full, n - kk)[n - kk]
                assert g_s[-1] >= kth - tol.max() - 1e-12, tag
                assert len(set(g_i.tolist())) == kk, tag
            del rr64
        else:
            bf16_check_sampled(metric, pm, rows, Q, k, gi, gs, gc, sample)
        stats["bf16"] += 1
    elif bf16:
        ix.enable_bf16()
        gi, gs, gc = ix.search_batch_brute_force_bf16(Q, k)
        pm = po.COSINE if metric == DM.Cosine else po.DOT
        if kind in ("small_ints",) and metric == DM.DotProduct:
            eid, esc = po.scan_topk_bf16(pm, rows, Q, kk, nthreads=NT)
            assert np.array_equal(gi[:, :kk], eid) and np.array_equal(gs[:, :kk], esc), tag
        else:
            # the tolerance check needs separated scores to compare ids; duplicates make every rank a tie group: values only
            if kind == "dups" or kind == "zeros_mixed":
                rr, qq = po.round_bf16(rows).astype(np.float64), po.round_bf16(Q).astype(np.float64)
                full = qq @ rr.T
                scale = np.linalg.norm(qq, axis=1)[:, None] * np.linalg.norm(rr, axis=1)[None, :]
                if metric == DM.Cosine:
                    with np.errstate(invalid="ignore", divide="ignore"):
                        full = np.where(scale > 0, full / scale, 0.0)
                    scale = np.ones_like(full)
                for qi in range(nq):
                    g_i, g_s = gi[qi, :kk].astype(np.int64), gs[qi, :kk].astype(np.float64)
                    assert np.all(np.abs(g_s - full[qi, g_i]) <= 1e-5 * np.maximum(scale[qi, g_i], 1e-30) + 1e-12), tag
                    kth = np.sort(full[qi])[::-1][kk - 1]
                    assert g_s[-1] >= kth - 1e-5 * max(scale[qi].max(), 1e-30) - 1e-12, tag
                    assert len(set(g_i.tolist())) == kk, tag
            else:
                bf16_check(metric, pm, rows, Q, k, gi, gs, gc)
        stats["bf16"] += 1
    elif a.bits_big:  # the four-bit GEMM path: every query against the vector-ALU kernels, a sample against the oracle
        gi, gs, gc = ix.search_batch_brute_force(Q, k)
        assert ix.last_kernels() & va.KERNEL_BITS_GEMM, tag + " (not served by the four-bit GEMM path)"
        ix.set_option(va.OPT_SWEEP_ENGINE, 0)
        vi, vs, vc = ix.search_batch_brute_force(Q, k)
        ix.set_option(va.OPT_SWEEP_ENGINE, -1)
        assert np.array_equal(gc, vc) and np.array_equal(gi, vi) and np.array_equal(bits(gs), bits(vs)), tag + " (GEMM vs vector-ALU kernels)"
        sample = np.unique(np.concatenate([[0, nq - 1, 255 % nq, 256 % nq], rng.integers(0, nq, 12)]))
        eid, esc = po.scan_topk(int(metric), rows, Q[sample], kk, po.MODE_C, nthreads=NT)
        assert np.all(gc == kk), tag
        assert np.array_equal(gi[sample][:, :kk], eid) and np.array_equal(bits(gs[sample][:, :kk]), bits(esc)), tag
        stats["bits_gemm"] = stats.get("bits_gemm", 0) + 1
    else:
        gi, gs, gc = ix.search_batch_brute_force(Q, k)
        mode = po.MODE_M if ix.sweep_arith_mode(k) == "M" else po.MODE_C
        eid, esc = po.scan_topk(int(metric), rows, Q, kk, mode, nthreads=NT)
        assert np.all(gc == kk), tag
        assert np.array_equal(gi[:, :kk], eid), tag
        assert np.array_equal(bits(gs[:, :kk]), bits(esc)), tag
        if a.select or a.wide:  # again (a parked handle answers at the other level) and with a tail chunk that is no selection batch
            lv, st2 = ix.last_select_level(), ix.last_split_stats()
            stats.setdefault(f"level{lv}", 0)
            stats[f"level{lv}"] += 1
            stats["unproven"] = stats.get("unproven", 0) + st2[1]
            gi2, gs2, _ = ix.search_batch_brute_force(Q, k)
            assert np.array_equal(gi2, gi) and np.array_equal(bits(gs2), bits(gs)), tag + " (second call)"
        stats["f32"] += 1
    ix.close()
    if it % 20 == 0:
        print(f"[fuzz] {it} cases ok ({stats})", flush=True)
print(f"[fuzz] done: {it} cases, all equal to the oracle ({stats})")
