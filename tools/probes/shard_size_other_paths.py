"""One shard of BASELINE configs[4] (6 250 000 x 768 rows, 4.8e9 elements) through the paths the f32 cosine test of
tests/test_gpu_headline_sizes.py does not reach: Euclidean batches (augmented bf16 image + l2_rerank_verify) and the SQ8 storage
mode (4.8 GB of codes, the dequantised bf16 image, the reference's scalar chains; the one-lane-per-row exact sweep for small calls).
Rows are generated chunk-wise on the device (bench.py's sharded leg), every chunk is scanned once by the oracle on the host for a
few sampled queries and the per-chunk lists are merged in the canonical order.  Prints one PASS / FAIL line per check.

STATE: written at the end of round 4 and NOT YET RUN — the round's GPU minutes were spent (its f32 cosine sibling in tests/ did run:
profiles/r04_configs4_shard_size_parity.txt).  First thing to run in round 5:
    gpurun --timeout 900 -- 'python tools/probes/shard_size_other_paths.py > gpurun_out/shard_other.log 2>&1'
(SR=1000000 in the environment for a quick pass over the same code paths at a size the fixed tests already cover.)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import velesdb_amd as va  # noqa: E402
if __import__("os").environ.get("VELESDB_HIP_LIB"):  # a kernel-variant build: the package reads no environment, probe scripts bind it themselves
    from velesdb_amd import _ffi as _vffi  # noqa: E402
    _vffi.use_library(__import__("os").environ["VELESDB_HIP_LIB"])
from oracle import pyoracle as po  # noqa: E402

SR = int(os.environ.get("SR", 6_250_000))
D, K, BQ, chunk = 768, 10, 1024, 1_000_000
t00 = time.time()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def report(name, ok, extra=""):
    print(("PASS " if ok else "FAIL ") + name + (" " + extra if extra else ""), flush=True)


dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(4242)
stream = torch.cuda.current_stream().cuda_stream
gq = torch.Generator(device=dev)
gq.manual_seed(48)
qs = torch.randn((BQ, D), generator=gq, device=dev).cpu().numpy()
rng = np.random.default_rng(9)
s_l2 = np.unique(np.concatenate([[0, 255, 256, 1023], rng.integers(0, BQ, 8)]))
s_sq = np.unique(np.concatenate([[0, 1023], rng.integers(0, BQ, 4)]))
ixe = va.HnswIndex(D, va.DistanceMetric.Euclidean, va.HnswParams(16, 100, SR))
ixs = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(16, 100, SR))
ixs.set_storage_mode(va.StorageMode.SQ8)
nt = po.host_threads()
acc = {"l2": [np.empty((len(s_l2), 0), np.int64), np.empty((len(s_l2), 0), np.float32)],
       "sq": [np.empty((len(s_sq), 0), np.int64), np.empty((len(s_sq), 0), np.float32)]}


def fold(key, ei, es, base, ascending):
    bi = np.concatenate([acc[key][0], ei.astype(np.int64) + base], axis=1)
    bs = np.concatenate([acc[key][1], es], axis=1)
    s64 = bs.astype(np.float64)
    order = np.lexsort((bi, s64 if ascending else -s64), axis=1)[:, :K]
    acc[key] = [np.take_along_axis(bi, order, axis=1), np.take_along_axis(bs, order, axis=1)]


for base in range(0, SR, chunk):
    n_c = min(chunk, SR - base)
    c = torch.randn((n_c, D), generator=g, device=dev)
    torch.cuda.synchronize()
    ixe.upload_dev(base, c.data_ptr(), n_c, stream)
    ixs.upload_dev(base, c.data_ptr(), n_c, stream)
    torch.cuda.synchronize()
    host = c.cpu().numpy()
    del c
    ei, es = po.scan_topk(po.EUCLIDEAN, host, qs[s_l2], K, po.MODE_C, nthreads=nt)
    fold("l2", ei, es, base, True)
    ei, es = po.scan_topk_sq8(po.COSINE, host, qs[s_sq], K, nthreads=nt)
    fold("sq", ei, es, base, False)
    del host
print(f"corpus + oracle scans: {time.time() - t00:.1f} s", flush=True)

# Euclidean, 1 024 queries: the selection stage over the augmented image
gi, gs, gc = ixe.search_batch_brute_force(qs, K)
report("euclidean level", ixe.last_select_level() == 2, f"level {ixe.last_select_level()} unproven {ixe.last_split_stats()}")
report("euclidean 1024-query batch ids", np.array_equal(gi[s_l2].astype(np.int64), acc["l2"][0]) and bool(np.all(gc == K)))
report("euclidean 1024-query batch score bits", np.array_equal(bits(gs[s_l2]), bits(acc["l2"][1])))
i4, s4, _ = ixe.search_batch_brute_force(qs[s_l2[:4]], K)
report("euclidean 4-query call", np.array_equal(i4.astype(np.int64), acc["l2"][0][:4]) and np.array_equal(bits(s4), bits(acc["l2"][1][:4])))
ixe.close()
torch.cuda.empty_cache()

# SQ8 storage mode, 1 024 queries: selection over the dequantised image + the reference's chain; then the exact code sweep
gi, gs, gc = ixs.search_batch_sq8(qs, K)
report("sq8 level", ixs.last_select_level() == 3, f"level {ixs.last_select_level()} unproven {ixs.last_split_stats()}")
report("sq8 1024-query batch ids", np.array_equal(gi[s_sq].astype(np.int64), acc["sq"][0]) and bool(np.all(gc == K)))
report("sq8 1024-query batch score bits", np.array_equal(bits(gs[s_sq]), bits(acc["sq"][1])))
i4, s4, _ = ixs.search_batch_sq8(qs[s_sq[:4]], K)
report("sq8 4-query call (exact code sweep)", np.array_equal(i4.astype(np.int64), acc["sq"][0][:4]) and np.array_equal(bits(s4), bits(acc["sq"][1][:4])))
ixs.close()
print(f"total {time.time() - t00:.1f} s", flush=True)
