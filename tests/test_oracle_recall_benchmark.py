"""The reference's recall benchmark (crates/velesdb-core/benches/recall_benchmark.rs) as a CPU test of the oracle: its LCG data
generator (:22-34), `HnswParams::max_recall(128)` (M 32, ef_construction 500), cosine, sequential inserts, queries from seeds n .. ,
recall@k = |truth ∩ result| / |truth| (metrics.rs:46-57) against the exact ranking, for every SearchQuality.  The reference publishes
no threshold in the bench itself; its parameter tables promise >= 95 % from Balanced up (params.rs:59-71) — that is what is asserted,
plus what the definitions force (Perfect is the exact scan: 1.0).  This is the workload behind the recall numbers of bench.py's graph
legs; the GPU walks are bit-compared with this oracle on the same graphs (tests/test_gpu_hnsw.py).  CPU only, ~15 s."""
import numpy as np
import pytest

from oracle import pyoracle as po
from velesdb_amd.metrics import recall_at_k
from velesdb_amd.params import HnswParams

QUALITIES = [("fast", po.Q_FAST), ("balanced", po.Q_BALANCED), ("accurate", po.Q_ACCURATE), ("perfect", po.Q_PERFECT)]


def generate_vectors(dim, seeds):
    """recall_benchmark.rs:22-34 for a vector of seeds: state = state * 1103515245 + 12345 (wrapping u64),
    value = ((state >> 16) & 0x7FFF) as f32 / 32768.0 * 2.0 - 1.0"""
    state = np.asarray(seeds, dtype=np.uint64).copy()
    out = np.empty((state.size, dim), dtype=np.float32)
    with np.errstate(over="ignore"):
        for j in range(dim):
            state = state * np.uint64(1103515245) + np.uint64(12345)
            out[:, j] = ((state >> np.uint64(16)) & np.uint64(0x7FFF)).astype(np.float32) / np.float32(32768.0) * np.float32(2.0) - np.float32(1.0)
    return out


def test_generator_matches_plain_integer_arithmetic():
    M64 = (1 << 64) - 1
    for seed in (0, 1, 12345, 10_000 + 99, (1 << 63) + 5):
        s, exp = seed, []
        for _ in range(16):
            s = (s * 1103515245 + 12345) & M64
            exp.append(np.float32(np.float32((s >> 16) & 0x7FFF) / np.float32(32768.0)) * np.float32(2.0) - np.float32(1.0))
        assert generate_vectors(16, [seed])[0].tolist() == [float(x) for x in exp]
    v = generate_vectors(128, np.arange(50))
    assert np.all((v >= -1.0) & (v < 1.0)) and abs(float(v.mean())) < 0.05      # "Range [-1, 1]"


def build(n, dim):
    data = generate_vectors(dim, np.arange(n))
    p = HnswParams.max_recall(dim)
    assert (p.max_connections, p.ef_construction) == (32, 500)                    # :92-94
    ix = po.HnswIndex(dim, po.COSINE, po.MODE_R, p.max_connections, p.ef_construction)
    for i, v in enumerate(data):
        ix.insert(i, v)
    return data, ix


def recalls(data, ix, queries, k):
    # brute_force_knn (:37-69): cosine similarity of every row, descending
    dn = data / np.linalg.norm(data.astype(np.float64), axis=1, keepdims=True)
    out = {}
    for name, q in QUALITIES:
        tot = 0.0
        for qv in queries:
            sims = dn @ (qv.astype(np.float64) / np.linalg.norm(qv.astype(np.float64)))
            truth = np.argsort(-sims, kind="stable")[:k].tolist()
            ids, sc = ix.search_with_quality(qv, k, q)
            assert len(ids) == k and np.all(np.diff(sc) <= 0)
            tot += recall_at_k(truth, ids.tolist())                       # metrics.rs:46-57, as the bench does (:137)
        out[name] = tot / len(queries)
    return out


@pytest.mark.parametrize("k", [10, 100])
def test_recall_1000x128(k):
    """bench_hnsw_recall, n = 1000 (:84-118): ten queries from seeds 1000 .. 1009"""
    data, ix = build(1000, 128)
    r = recalls(data, ix, generate_vectors(128, np.arange(1000, 1010)), k)
    assert r == {"fast": 1.0, "balanced": 1.0, "accurate": 1.0, "perfect": 1.0}, r


def test_recall_10000x128():
    """print_recall_stats' index (:171-200): 10 000 x 128, queries from seeds 10 000 .. (twenty of its hundred)"""
    data, ix = build(10_000, 128)
    r = recalls(data, ix, generate_vectors(128, np.arange(10_000, 10_020)), 10)
    assert r["perfect"] == 1.0 and r["accurate"] >= 0.99 and r["balanced"] >= 0.95 and r["fast"] >= 0.85, r
    assert r["fast"] <= r["balanced"] <= r["accurate"] <= r["perfect"]
