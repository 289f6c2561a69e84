#!/usr/bin/env python3
"""Writes tests/golden/reference_kats.json: the known-answer vectors the REFERENCE's own
tests hold for the hot path (inputs + expected outputs + tolerance + the reference
file:line that states them).  Pure data: no reference source text is copied.
Run:  python tests/golden/make_reference_kats.py
"""
import json
import os

C = "crates/velesdb-core/src/"
kats = []


def kat(fn, args, expect, tol, src, **kw):
    kats.append(dict(fn=fn, args=args, expect=expect, tol=tol, src=src, **kw))


# simd_tests.rs — cosine / euclid / dot / norm / sql2
kat("cosine", [[1, 2, 3, 4], [1, 2, 3, 4]], 1.0, 1e-5, C + "simd_tests.rs:24-32")
kat("cosine", [[1, 0, 0, 0], [0, 1, 0, 0]], 0.0, 1e-5, C + "simd_tests.rs:34-43")
kat("cosine", [[1, 2, 3, 4], [-1, -2, -3, -4]], -1.0, 1e-5, C + "simd_tests.rs:45-54")
kat("cosine", [[1, 2, 3], [0, 0, 0]], 0.0, 1e-5, C + "simd_tests.rs:56-62")
kat("euclidean", [[1, 2, 3, 4], [1, 2, 3, 4]], 0.0, 1e-5, C + "simd_tests.rs:64-72")
kat("euclidean", [[0, 0, 0], [3, 4, 0]], 5.0, 1e-5, C + "simd_tests.rs:74-83")
kat("dot", [[1, 2, 3, 4], [5, 6, 7, 8]], 70.0, 1e-5, C + "simd_tests.rs:105-112")
kat("dot", [[1, 2, 3, 4, 5], [5, 4, 3, 2, 1]], 35.0, 1e-5, C + "simd_tests.rs:186-194")
kat("dot", [[3.0], [4.0]], 12.0, 1e-5, C + "simd_tests.rs:206-212")
kat("euclidean", [[3.0], [4.0]], 1.0, 1e-5, C + "simd_tests.rs:206-213")
kat("cosine", [[1, 0], [0, 1]], 0.0, 1e-5, C + "simd_tests.rs:215-218")
kat("norm", [[0, 0, 0]], 0.0, 1e-5, C + "simd_tests.rs:231-235")
kat("norm", [[1, 0, 0]], 1.0, 1e-5, C + "simd_tests.rs:237-241")
kat("norm", [[3, 4]], 5.0, 1e-5, C + "simd_tests.rs:243-247")
kat("sql2", [[1, 2, 3], [1, 2, 3]], 0.0, 1e-5, C + "simd_tests.rs:251-255")
kat("sql2", [[0, 0], [3, 4]], 25.0, 1e-5, C + "simd_tests.rs:257-262")
# Hamming over f32 thresholded at 0.5 (exact)
kat("hamming", [[1, 0, 1, 0], [1, 0, 1, 0]], 0.0, 0, C + "simd_tests.rs:266-270")
kat("hamming", [[1, 0, 1, 0], [0, 1, 0, 1]], 4.0, 0, C + "simd_tests.rs:272-277")
kat("hamming", [[1, 1, 0, 0], [1, 0, 0, 1]], 2.0, 0, C + "simd_tests.rs:279-284")
kat("hamming", [[1, 0, 1, 0, 1], [0, 0, 1, 1, 1]], 2.0, 0, C + "simd_tests.rs:286-291")
kat("hamming", [[1.0] * 32, [1.0] * 32], 0.0, 0, C + "simd_dispatch.rs:484-489")
kat("hamming", [[1.0] * 32, [0.0] * 32], 32.0, 0, C + "simd_dispatch.rs:491-497")
kat("hamming", [[1.0] * 32, [0.0] * 16 + [1.0] * 16], 16.0, 0, C + "simd_dispatch.rs:499-509")
kat("hamming", [[1, 0, 1, 0], [0, 1, 1, 0]], 2.0, 0, C + "simd_dispatch.rs:600-610")
# Jaccard
kat("jaccard", [[1, 0, 1, 0], [1, 0, 1, 0]], 1.0, 1e-5, C + "simd_tests.rs:295-299")
kat("jaccard", [[1, 0, 0, 0], [0, 1, 0, 0]], 0.0, 1e-5, C + "simd_tests.rs:301-306")
kat("jaccard", [[1, 1, 0, 0], [1, 0, 1, 0]], 1.0 / 3.0, 1e-5, C + "simd_tests.rs:308-314")
kat("jaccard", [[0, 0, 0, 0], [0, 0, 0, 0]], 1.0, 1e-5, C + "simd_tests.rs:316-321")
# simd_dispatch.rs / simd_avx512_tests.rs
kat("euclidean", [[0, 0], [3, 4]], 5.0, 1e-5, C + "simd_dispatch.rs:436-442")
kat("euclidean", [[1.0] * 64, [1.0] * 64], 0.0, 1e-6, C + "simd_dispatch.rs:444-449")
kat("cosine", [[1.0] * 32, [1.0] * 32], 1.0, 1e-5, C + "simd_dispatch.rs:451-456")
kat("cosine", [[1.0] + [0.0] * 31, [0.0, 1.0] + [0.0] * 30], 0.0, 1e-5, C + "simd_dispatch.rs:458-467")
kat("cosine", [[1.0] * 16, [-1.0] * 16], -1.0, 1e-5, C + "simd_dispatch.rs:469-475")
kat("dot", [[1.0] * 16, [2.0] * 16], 32.0, 1e-5, C + "simd_avx512_tests.rs:44-53")
kat("sql2", [[0.0] * 16, [3.0, 4.0] + [0.0] * 14], 25.0, 1e-5, C + "simd_avx512_tests.rs:95-107")
# DistanceEngine-level (native/distance.rs tests): distances
H = C + "index/hnsw/native/"
kat("engine_distance", ["euclidean", [0, 0, 0], [3, 4, 0]], 5.0, 1e-5, H + "distance.rs:236-243", engine="scalar")
kat("engine_distance", ["cosine", [1, 2, 3], [1, 2, 3]], 0.0, 1e-5, H + "distance.rs:225-234", engine="scalar")
kat("engine_distance",
    ["jaccard", [1.0] * 32 + [0.0] * 32, [1.0] * 48 + [0.0] * 16], 1.0 - 32.0 / 48.0, 1e-4,
    H + "distance.rs:284-306", engine="simd")
# transform_score (backend_adapter_tests.rs:90-115)
kat("transform_score", ["euclidean", 0.5], 0.5, 1.2e-7, H + "backend_adapter_tests.rs:90-96")
kat("transform_score", ["cosine", 0.3], 0.7, 1.2e-7, H + "backend_adapter_tests.rs:98-106")
kat("transform_score", ["cosine", 1.5], 0.0, 1.2e-7, H + "backend_adapter_tests.rs:98-106")
kat("transform_score", ["dot", 0.5], -0.5, 1.2e-7, H + "backend_adapter_tests.rs:108-115")
# SearchQuality::ef_search (hnsw/params.rs:309-319)
P = C + "index/hnsw/params.rs:309-319"
for q, k, e in [("fast", 10, 64), ("fast", 50, 100), ("balanced", 10, 128), ("balanced", 100, 400),
                ("accurate", 10, 512), ("accurate", 50, 800), ("perfect", 10, 4096), ("perfect", 50, 5000)]:
    kat("ef_search", [q, k], e, 0, P)
kat("ef_search", ["custom:30", 50], 50, 0, P)
kat("ef_search", ["custom:300", 50], 300, 0, P)

# closed-form datasets the reference's tests use (generators only; data is regenerated)
datasets = {
    "generate_test_vector": {"formula": "v[i]=sin(seed+0.1*i) (f32 ops)", "src": C + "simd_tests.rs:17-19"},
    "boundary_sizes": {"sizes": [7, 8, 9, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65],
                       "tol_rel": 1e-4, "src": C + "simd_avx512_tests.rs:225-277"},
    "ramp_graph": {"formula": "v_i[j]=32i+j, 100x32, Euclidean scalar engine, M16 efc100, q=v_0, k10 ef50",
                   "expect": "results[0].0==0, len<=10", "src": H + "graph_tests.rs:10-30"},
    "sinusoid_A": {"formula": "v_i[j]=sin(0.01(i+j)), 100x128 cosine simd, M16 efc100, k10 ef50",
                   "expect": "10 results, results[0].1<0.1", "src": H + "tests.rs:10-29"},
    "sinusoid_B": {"formula": "v_i[j]=sin(0.001(128i+j)), 200x128 cosine, M16 efc100, q=v_{0,40,80,120,160}, k10 ef128",
                   "expect": "mean recall>=0.8", "src": H + "tests.rs:32-91"},
    "sinusoid_C": {"formula": "v_i[j]=sin(0.01(127i+j)), 500x128 cosine M32 efc200, q=sin(0.01j), k10 ef100",
                   "expect": ">=5 results ascending", "src": H + "graph_tests.rs:169-199"},
    "index_recall": {"formula": "v_i[j]=sin(0.001(64i+j)), 500x64 cosine auto params, q=sin(0.001j), Accurate k10",
                     "expect": "recall>=0.8", "src": C + "index/hnsw/index_tests.rs:1106-1158"},
    "gpu_template": {"formula": "v_i[j]=sin(0.01(i+j)) 100x128 cosine, q=cos(0.02j), brute force k10",
                     "expect": "reference asks >=8/10 id overlap; we require 10/10 + rank",
                     "src": C + "index/hnsw/index_tests.rs:1551-1588"},
}
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
with open(out, "w") as f:
    json.dump({"reference": "cyberlife-coder/velesdb v1.4.1", "kats": kats, "datasets": datasets}, f, indent=1)
print(f"wrote {len(kats)} KATs to {out}")
