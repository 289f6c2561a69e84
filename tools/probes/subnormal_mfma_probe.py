"""Subnormal products through the matrix-core sweeps: the reference's `test_very_small_values` input (components 1e-20:
products 1e-40, subnormal in f32) and mixed-scale rows, through the exact f32 sweep at three batch sizes (streaming matrix-core kernel,
GEMM-structured kernel, selection stage) — ids and score BITS against the oracle in the mode the index reports.  The vector-ALU
kernels are already pinned for this (tests/test_gpu_sweep.py::test_batch_distance_edge_cases); the matrix pipe is not.

STATE: written at the end of round 4 on the CPU, NOT YET RUN.  gpurun --timeout 300 -- 'python tools/probes/subnormal_mfma_probe.py'"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
from oracle import pyoracle as po  # noqa: E402

DM = va.DistanceMetric


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def check(name, metric, rows, queries, k=10):
    ix = va.HnswIndex(rows.shape[1], metric)
    ix.upload(np.arange(rows.shape[0], dtype=np.uint64), rows)
    ok_all = True
    for nq in (1, 4, 68, min(96, queries.shape[0])):
        q = queries[:nq]
        ids, sc, cnt = ix.search_batch_brute_force(q, k)
        mode = po.MODE_M if ix.sweep_arith_mode(k) == "M" else po.MODE_C
        eid, esc = po.scan_topk(int(metric), rows, q, k, mode, nthreads=po.host_threads())
        ok = bool(np.array_equal(ids, eid) and np.array_equal(bits(sc), bits(esc)))
        ok_all &= ok
        print(("PASS " if ok else "FAIL ") + f"{name} {metric.name} nq={nq} mode={'M' if mode == po.MODE_M else 'C'} level={ix.last_select_level()}"
              + ("" if ok else f" first diff: gpu {sc[0][:3]} oracle {esc[0][:3]}"), flush=True)
    ix.close()
    return ok_all


rng = np.random.default_rng(5)
n, d = 70_000, 768                      # >= 65 536 rows: large batches reach the selection stage
tiny_rows = (rng.standard_normal((n, d)) * 1e-20).astype(np.float32)
tiny_q = (rng.standard_normal((96, d)) * 1e-20).astype(np.float32)
mixed = rng.standard_normal((n, d)).astype(np.float32)
mixed[::7] *= np.float32(1e-25)         # every 7th row in the range where its products with a unit-scale query are subnormal-ish
mixed[::11] *= np.float32(1e-38)
unit_q = rng.standard_normal((96, d)).astype(np.float32)
ok = True
for metric in (DM.DotProduct, DM.Cosine):
    ok &= check("all-tiny", metric, tiny_rows, tiny_q)
    ok &= check("mixed-scale rows", metric, mixed, unit_q)
    ok &= check("tiny queries, unit rows", metric, rng.standard_normal((n, d)).astype(np.float32), tiny_q)
print("ALL PASS" if ok else "SOME FAIL")
