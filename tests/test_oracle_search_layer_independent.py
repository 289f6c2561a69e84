"""An independent statement of the hot function itself — `NativeHnsw::search_layer` (native/graph.rs:438-520: the layer-0 candidate
expansion BASELINE's north_star names) and the greedy descent `search_layer_single` (:407-436) — written in plain Python from the
reference's text over Python's own heaps, run on graphs the oracle built, and compared with the oracle's `search_layer` /
`search_layer_single` / `search`: same nodes, same order, same distance bits, and the same number of distance evaluations and
expansions (the counters the GPU kernels report and the roofline's algorithmic bytes are computed from).  Random float data: no two
distances tie, so the heaps' tie artefacts (restated separately, tests/test_oracle_graph.py) cannot enter.  CPU only."""
import heapq

import numpy as np
import pytest

from oracle import pyoracle as po

F = np.float32
F32_MAX = float(np.finfo(F).max)


def search_layer_py(g, metric, mode, q, entry_points, ef, layer):
    """graph.rs:438-520, statement for statement (prefetching left out: it moves no value)"""
    dist = lambda node: po.distance(metric, q, g.vector(node), mode)
    visited, candidates, results = set(), [], []          # candidates: min-heap by (dist, node); results: max-heap by (dist, node)
    n_dist = n_expand = 0
    for ep in entry_points:                               # :464-469
        d = dist(ep)
        n_dist += 1
        heapq.heappush(candidates, (d, ep))
        heapq.heappush(results, (-d, -ep))
        visited.add(ep)
    while candidates:                                     # :471
        c_dist, c_node = heapq.heappop(candidates)
        furthest = -results[0][0] if results else F32_MAX
        if c_dist > furthest and len(results) >= ef:      # :474-476
            break
        n_expand += 1
        for nb in g.neighbors(layer, c_node):             # :478, :491
            if nb not in visited:                         # :500 visited.insert
                visited.add(nb)
                d = dist(nb)
                n_dist += 1
                furthest = -results[0][0] if results else F32_MAX
                if d < furthest or len(results) < ef:     # :504
                    heapq.heappush(candidates, (d, nb))
                    heapq.heappush(results, (-d, -nb))
                    if len(results) > ef:                 # :508-510
                        heapq.heappop(results)
    out = sorted(((-nd, -nn) for nd, nn in results))      # :515-518 ascending distance
    return [n for _, n in out], [d for d, _ in out], n_dist, n_expand


def search_layer_single_py(g, metric, mode, q, entry, layer, count=None):
    """graph.rs:407-436: move to the closest neighbour until no neighbour is closer (one evaluation for the entry point of the call,
    one per neighbour looked at)"""
    best, best_d = entry, po.distance(metric, q, g.vector(entry), mode)
    n = 1
    while True:
        improved = False
        for nb in g.neighbors(layer, best):
            d = po.distance(metric, q, g.vector(nb), mode)
            n += 1
            if d < best_d:
                best, best_d, improved = nb, d, True
        if not improved:
            if count is not None:
                count[0] += n
            return best


def build(n, dim, metric, M, efc, seed, mode=po.MODE_R):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, dim)).astype(F)
    g = po.NativeHnsw(dim, metric, M, efc, mode)
    for v in X:
        g.insert(v)
    return g, X, rng


@pytest.mark.parametrize("metric", [po.EUCLIDEAN, po.COSINE, po.DOT])
@pytest.mark.parametrize("ef", [1, 8, 50, 200])
def test_search_layer_matches_the_plain_statement(metric, ef):
    g, X, rng = build(600, 24, metric, 8, 60, seed=11 + metric)
    for _ in range(6):
        q = rng.standard_normal(24).astype(F)
        eps = [int(g.entry_point)] if ef != 8 else [int(g.entry_point), 5, 17]          # several entry points too (insert's upper layers)
        nodes, dists, n_dist, n_expand = search_layer_py(g, metric, po.MODE_R, q, eps, ef, 0)
        ids, ds = g.search_layer(q, eps, ef, 0)
        assert ids.tolist() == nodes
        assert ds.view(np.uint32).tolist() == np.array(dists, dtype=F).view(np.uint32).tolist()
        assert len(nodes) == min(max(ef, len(eps)), 600) or len(nodes) <= ef + len(eps)
        assert len(set(np.array(dists, dtype=F).tolist())) == len(dists)            # the premise: no ties in this data


def test_full_search_is_descent_plus_layer0_expansion_and_counts_what_it_touches():
    """NativeHnsw::search (graph.rs:240-286): greedy descent through the upper layers, search_layer at layer 0 with ef, first k —
    and the oracle's counters equal what this statement evaluates and expands"""
    g, X, rng = build(1500, 32, po.EUCLIDEAN, 8, 80, seed=3)
    assert g.num_layers >= 2
    for _ in range(8):
        q = rng.standard_normal(32).astype(F)
        cur = int(g.entry_point)
        descent = [0]
        for layer in range(g.max_layer, 0, -1):
            nxt = search_layer_single_py(g, po.EUCLIDEAN, po.MODE_R, q, cur, layer, descent)
            assert nxt == g.search_layer_single(q, cur, layer)
            cur = nxt
        nodes, dists, n_dist, n_expand = search_layer_py(g, po.EUCLIDEAN, po.MODE_R, q, [cur], 64, 0)
        ids, ds = g.search(q, 10, 64)
        assert ids.tolist() == nodes[:10]
        assert ds.view(np.uint32).tolist() == np.array(dists[:10], dtype=F).view(np.uint32).tolist()
        # the oracle's counters (what the GPU kernels must report, what `alg_bytes_per_launch` is computed from): every distance
        # evaluation of the descent and of the layer-0 expansion, every candidate expanded at layer 0
        assert po.NativeHnsw.last_stats() == (descent[0] + n_dist, n_expand)
        # brute-force sanity: the graph's answer holds the true nearest neighbour
        assert int(np.argmin(((X - q) ** 2).sum(axis=1))) in nodes[:10]


def select_neighbors_py(g, metric, mode, candidates, max_neighbors, alpha=1.0):
    """graph.rs:522-578, statement for statement: all candidates when they fit; otherwise a candidate is kept while
    alpha * d(q, c) <= d(c, s) holds for EVERY already selected s (the first candidate always), then the quota is filled with the
    closest candidates not yet selected, in candidate order"""
    if not candidates:
        return []
    if len(candidates) <= max_neighbors:
        return [c for c, _ in candidates]
    selected = []
    for cid, cdist in candidates:
        if len(selected) >= max_neighbors:
            break
        cv = g.vector(cid)
        diverse = all(F(alpha) * F(cdist) <= F(po.distance(metric, cv, g.vector(s), mode)) for s in selected)
        if diverse or not selected:
            selected.append(cid)
    if len(selected) < max_neighbors:
        for cid, _ in candidates:
            if len(selected) >= max_neighbors:
                break
            if cid not in selected:
                selected.append(cid)
    return selected


@pytest.mark.parametrize("metric", [po.EUCLIDEAN, po.COSINE])
@pytest.mark.parametrize("alpha", [1.0, 1.2])
def test_select_neighbors_matches_the_plain_statement(metric, alpha):
    """the candidate lists are what insert passes: search_layer's output at ef_construction, ascending (graph.rs:196-204)"""
    g, X, rng = build(500, 16, metric, 8, 60, seed=21 + metric)
    g.set_alpha(alpha)
    for _ in range(10):
        q = rng.standard_normal(16).astype(F)
        ids, ds = g.search_layer(q, [int(g.entry_point)], 60, 0)
        cand = list(zip(ids.tolist(), ds.tolist()))
        for max_nb in (4, 8, 16, 59, 60, 100):
            assert g.select_neighbors(cand, max_nb) == select_neighbors_py(g, metric, po.MODE_R, cand, max_nb, alpha), (max_nb, alpha)
    assert g.select_neighbors([], 8) == []
