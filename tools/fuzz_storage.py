#!/usr/bin/env python3
"""GPU fuzz of the storage-mode scans (SQ8 asymmetric distances, sign-bit Hamming) and the int8 dual-precision traversal
against the oracle (not part of the test-suite): random shapes, constant / zero / duplicate rows, deletions.

    python tools/fuzz_storage.py --seconds 240 --seed 1
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
from oracle import pyoracle as po  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--seconds", type=float, default=240)
p.add_argument("--seed", type=int, default=1)
p.add_argument("--select", action="store_true", help="SQ8 batches that take the selection stage (>= 80 queries, >= 65 536 rows, dim % 64 == 0; k <= 10: block-local lists, k <= 128: WIDE)")
a = p.parse_args()
rng = np.random.default_rng(a.seed)
DM, SM = va.DistanceMetric, va.StorageMode


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)


def make(kind, n, dim):
    r = rng.standard_normal((n, dim)).astype(np.float32)
    if kind == "dups":
        r = r[rng.integers(0, max(1, n // 30), n)]
    elif kind == "const":
        m = rng.random(n) < 0.3
        r[m] = rng.standard_normal((int(m.sum()), 1)).astype(np.float32)      # constant rows: range < EPSILON branch
    elif kind == "zeros":
        r[rng.random(n) < 0.3] = 0.0
    elif kind == "ints":
        r = rng.integers(-3, 4, size=(n, dim)).astype(np.float32)
    elif kind == "wide":
        r *= rng.choice([1e-6, 1.0, 1e4], size=(n, 1)).astype(np.float32)
    return r


t_end = time.time() + a.seconds
it = 0
stats = {"sq8": 0, "binary": 0}
while time.time() < t_end:
    it += 1
    mode = "sq8" if rng.random() < 0.6 else "binary"
    metric = [DM.Cosine, DM.Euclidean, DM.DotProduct][int(rng.integers(0, 3))]
    n = int(rng.choice([1, 9, 100, 257, 1000, 5000, 20000]))
    dim = int(rng.choice([1, 3, 17, 64, 100, 128, 333, 768]))
    nq = int(rng.choice([1, 2, 3, 4, 5, 9, 33, 70, 200]))
    k = int(rng.choice([1, 5, 10, 31, 64]))
    kind = str(rng.choice(["normal", "dups", "const", "zeros", "ints", "wide"]))
    if a.select:
        mode = "sq8"
        metric = [DM.Cosine, DM.DotProduct, DM.Euclidean][int(rng.integers(0, 3))]
        n = int(rng.choice([66_000, 120_000]))
        dim = int(rng.choice([128, 256, 768]))
        nq = int(rng.choice([80, 150, 256, 480]))
        k = int(rng.choice([1, 5, 10, 11, 31, 64, 100, 128]))   # (k > 10: the WIDE selection over the SQ8 image, Cosine / DotProduct)
    rows = make(kind, n, dim)
    Q = make(kind if kind != "dups" else "normal", nq, dim)
    ids = rng.permutation(n).astype(np.uint64) * 3 + 1
    tag = f"it={it} {mode} {metric.name} n={n} dim={dim} nq={nq} k={k} {kind}"
    ix = va.HnswIndex(dim, metric)
    half = n // 2
    ix.upload(ids[:half], rows[:half])
    ix.set_storage_mode(SM.SQ8 if mode == "sq8" else SM.Binary)
    ix.upload(ids[half:], rows[half:])
    live = np.ones(n, bool)
    if n > 20 and rng.random() < 0.3:
        dead = rng.choice(n, n // 5, replace=False)
        for d in dead:
            ix.remove(int(ids[d]))
        live[dead] = False
    sel = np.nonzero(live)[0]
    kk = min(k, len(sel))
    if mode == "sq8":
        pm = {DM.Cosine: po.COSINE, DM.Euclidean: po.EUCLIDEAN, DM.DotProduct: po.DOT}[metric]
        gi, gs, gc = ix.search_batch_sq8(Q, k)
        ei, es = po.scan_topk_sq8(pm, rows[sel], Q, max(kk, 1), nthreads=po.host_threads())
        if a.select:
            stats["selected"] = stats.get("selected", 0) + (1 if ix.last_select_level() in (3, 4) else 0)
            stats["wide"] = stats.get("wide", 0) + (1 if ix.last_select_level() == 4 else 0)
            stats["unproven"] = stats.get("unproven", 0) + ix.last_split_stats()[1]
    else:
        gi, gs, gc = ix.search_batch_binary(Q, k)
        ei, es = po.scan_topk_binary(rows[sel], Q, max(kk, 1))
    assert np.all(gc == kk), tag
    if kk:
        assert np.array_equal(gi[:, :kk], ids[sel[ei[:, :kk].astype(np.int64)]]), tag
        assert np.array_equal(bits(gs[:, :kk]), bits(es[:, :kk])), tag
    stats[mode] += 1
    ix.close()
    if it % 20 == 0:
        print(f"[fuzz-storage] {it} cases ok ({stats})", flush=True)
print(f"[fuzz-storage] done: {it} cases, all equal to the oracle ({stats})")
