"""Probe (round 6): how long the chip takes to settle into the headline step.  bench.py's timed region (20 steps behind W warm-up
steps) reads ~5 % slower than the same 20 steps repeated at once; this prints the wall time of every single step (a synchronise behind
each: +launch latency, the same for all) from the first call on, and again after the GPU sat idle for 0.3 s and 3 s."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import velesdb_amd as va  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(42)
N, D, Q, K = 1_000_000, 768, 1024, 10
rows = torch.randn((N, D), generator=g, device=dev)
g.manual_seed(43)
qs = torch.randn((4 * Q, D), generator=g, device=dev)
ix = va.HnswIndex(D, va.DistanceMetric.Cosine, va.HnswParams(32, 400, N))
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
ix.upload_dev(0, rows.data_ptr(), N, st)
torch.cuda.synchronize()
ids = torch.empty((Q, K), dtype=torch.int64, device=dev)
sc = torch.empty((Q, K), dtype=torch.float32, device=dev)
n = torch.empty((Q,), dtype=torch.int32, device=dev)


def one(i):
    ix.search_batch_dev(qs[(i % 4) * Q:].data_ptr(), Q, K, 0, va.MODE_BRUTE, ids.data_ptr(), sc.data_ptr(), n.data_ptr(), st)


def series(tag, cnt):
    ts = []
    for i in range(cnt):
        t = time.perf_counter()
        one(i)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    print(tag, " ".join(f"{x:.3f}" for x in ts), flush=True)


def blocks(tag, nblk, per):
    out = []
    for b in range(nblk):
        t = time.perf_counter()
        for i in range(per):
            one(i)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / per * 1e3)
    print(tag, " ".join(f"{x:.4f}" for x in out), flush=True)


series("first calls, per step (sync each):", 40)
time.sleep(0.3)
blocks("after 0.3 s idle, blocks of 5 steps:", 16, 5)
time.sleep(3.0)
blocks("after 3 s idle, blocks of 5 steps:", 16, 5)
time.sleep(0.3)
blocks("after 0.3 s idle, blocks of 20 steps:", 6, 20)
