"""Stamped timeline of the selection kernel's k-tile: runs the headline batch (1 M x 768 cosine, 1 024
queries, k = 10) on a -DVDB_PP_STAMP variant of the library (tools/probes/pp_variants.sh) and prints, for one wave of each wave row
of block 8, the average shader cycles per k-tile spent in each segment of the four-phase schedule of sweep_topk_gemm_bf16_pp.
    python tools/probes/pp_stamp_probe.py tools/probes/out/libvelesdb_hip_stamp1.so [rows] [cosine|hamming|jaccard]
(hamming / jaccard: the four-bit instance, rows of uniform bits; a row tile is 3 k-tiles of 256 dimensions instead of 12 of 64)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from velesdb_amd import _ffi  # noqa: E402

lib_path = os.path.abspath(sys.argv[1])
_ffi.use_library(lib_path)
import torch  # noqa: E402,F401  (first: the HIP runtime it loads is the one the library binds to)
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])

N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
METRIC = sys.argv[3] if len(sys.argv) > 3 else "cosine"
BITS = METRIC in ("hamming", "jaccard")
KT_PER_ROW_TILE = 3 if BITS else 12
D, Q, K = 768, 1024, 10
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
ix = va.HnswIndex(D, {"cosine": va.DistanceMetric.Cosine, "hamming": va.DistanceMetric.Hamming, "jaccard": va.DistanceMetric.Jaccard}[METRIC], va.HnswParams(16, 100, N))
stream = torch.cuda.current_stream().cuda_stream
for base in range(0, N, 250_000):
    c = torch.randn((min(250_000, N - base), D), generator=g, device=dev)
    if BITS:
        c = (c > 0).float()
    torch.cuda.synchronize()
    ix.upload_dev(base, c.data_ptr(), c.shape[0], stream)
    torch.cuda.synchronize()
    del c
qs = torch.randn((Q, D), generator=g, device=dev)
if BITS:
    qs = (qs > 0).float()
qs = qs.cpu().numpy()
for _ in range(4):
    ids, sc, cnt = ix.search_batch_brute_force(qs, K)
assert BITS or ix.last_select_level() == 2, ix.last_select_level()
L = C.CDLL(lib_path)
buf = (C.c_ulonglong * 64)()
rc = L.vdb_hip_debug_pp_stamps(buf)
assert rc == 0, rc
names = ["ph1-2: reads + requests + waits + opening barrier", "ph1-2: 16 products", "ph1-2: closing barrier", "ph1-2: (issue of reads + requests only)",
         "ph3-4: requests (+ reads) + vmcnt/lgkmcnt waits + opening barrier", "ph3-4: 16 products", "ph3-4: closing barrier", "ph3-4: (issue only)",
         "epilogue (quick test + protocol + re-read), per ROW TILE", "k-tiles"]
for w, row in enumerate(("wave 0 (row 0)", "wave 4 (row 1)")):
    v = [int(buf[w * 28 + i]) for i in range(28)]
    kt = max(v[9], 1)
    print(f"{row}: {kt} k-tiles of the batch's last (largest) selection launch, block 8")
    tot = 0.0
    for i in range(8):
        per = v[i] / kt   # two phases of the pair per k-tile
        tot += per
        print(f"   {names[i]:70s} {per:9.1f} cycles per k-tile  ({per / 2:7.1f} per phase)")
    print(f"   {'sum of the segments':70s} {tot:9.1f} cycles per k-tile  (the 64 products of a wave need 64 x 16 = 1 024; both rows' = 2 048 per SIMD)")
    rtiles = kt / KT_PER_ROW_TILE
    print(f"   {names[8]:70s} {v[8] / rtiles:9.1f} cycles per row tile ({KT_PER_ROW_TILE} k-tiles)")
    for slot, nm in ((11, "epilogue: alignment barrier (waves 0-3 wait for the last products of waves 4-7)"), (10, "epilogue: quick test"),
                     (12, "epilogue: look phase (scan of hot lanes + finish), all rounds"), (13, "epilogue: sync-point barrier(s)"),
                     (16, "   look phase: the lazy group masks (stamped from the alignment barrier)"), (20, "   look phase: the dump path (few hot lanes)"), (17, "   look phase: scans"), (18, "   look phase: finishes"),
                     (14, "epilogue: compaction + its barrier (when one runs)"), (15, "epilogue: append"), (8, "epilogue: the rest (re-derive lane terms, re-read A fragments, fall-behind barrier)")):
        print(f"   {nm:70s} {v[slot] / rtiles:9.1f} cycles per row tile")
    print(f"   row tiles in which this wave had a hot lane: {v[19]} of {rtiles:.0f}; they began in the dump path: {v[21]}; hot lanes in all: {v[22]}; rounds: {v[23]}; dump passes: {v[24]}")
ix.close()
