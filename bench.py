#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path on MI355X.

Headline workload (BASELINE.json configs[1]): 1M x 768-D f32 cosine, k=10, exact distance sweep + fused
GPU top-k (HnswIndex::search_brute_force semantics, recall@10 = 1.0 by construction).  A "step" = one batch
of --batch queries searched against the HBM-resident corpus through the C ABI's device-pointer entry point
(vdb_hip_index_search_batch_dev); value = whole-job queries/s, inputs resident in HBM.

The same run also measures configs[2] (the "hnsw" object): the HNSW graph is built on the GPU over the same
corpus (batched construction), then --hnsw-batch queries per step go through the traversal kernel at
ef = --ef; recall@10 against the exact result, distance evaluations / expansions counted by the kernel =>
algorithmic bytes => achieved HBM GB/s.  Rank 0 times the CPU restatement of the reference beside both legs
(oracle mode R brute force; oracle HNSW search over the very same graph, all host cores).

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Headline = replica mode (every GPU holds the corpus, the query stream is split:
no data-path collective, weak scaling).  The same run also measures the range-sharded mode (every GPU holds a
different 1M-row shard, per-shard top-k, ONE RCCL all-gather of k (id,score) pairs per query, merge) and
reports it under "sharded".
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--settle-steps", type=int, default=40,
                   help="untimed steps of the same workload IN FRONT of the W warm-up steps (reported as `settle_steps`): the chip needs ~25 ms of "
                        "this kernel mix before its clocks settle (tools/probes/step_settle_probe.py), more than W = 3-5 steps give it; 0 = none")
    p.add_argument("--rows", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--batch", type=int, default=1024, help="queries per step (exact sweep)")
    p.add_argument("--metric", default="cosine")
    p.add_argument("--tile", type=int, default=128, help="largest query tile of the sweep (1,2,4,8,16,32,48; 128 = "
                   "batches of >= 64 queries go to the GEMM-structured matrix-core kernel)")
    p.add_argument("--engine", type=int, default=1, help="1 = matrix-core sweep for cosine/dot (default), 0 = VALU")
    p.add_argument("--shard-rows", type=int, default=0, help="rows per GPU in the range-sharded leg (0 = --rows; BASELINE "
                   "configs[4] at 8 GPUs: 6250000)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-tiles", action="store_true", help="skip the per-tile-size table (profiling passes)")
    p.add_argument("--cpu-sample-rows", type=int, default=1_000_000)
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    p.add_argument("--check-queries", type=int, default=64, help="queries of the headline batch compared with the oracle in the run")
    # graph leg (configs[2])
    p.add_argument("--no-hnsw", action="store_true")
    p.add_argument("--no-int8", action="store_true", help="skip the dual-precision (int8 traversal) leg")
    p.add_argument("--no-embedding-leg", action="store_true", help="skip the graph leg on embedding-like data")
    p.add_argument("--dist-single", action="store_true", help="self-test: initialise torch's RCCL process group even with one rank")
    p.add_argument("--no-sharded-leg", action="store_true", help="skip the range-sharded leg (profiling passes)")
    p.add_argument("--no-split", action="store_true", help="headline on the exact f32 matrix-core kernel (no selection stage)")
    p.add_argument("--select-level", type=int, default=3, choices=[0, 1, 2, 3],
                   help="selection stage of large exact batches: 0 exact kernel, 1 split-bf16, 2 plain bf16 with block-local lists, 3 the WIDE selection at every k (library default)")
    p.add_argument("--no-traffic-pass", action="store_true", help="skip the rocprofv3 FETCH_SIZE child pass that fills roofline.traffic")
    # the child of a traffic pass (run under rocprofv3 --pmc FETCH_SIZE by the parent): "headline" = the headline steps only,
    # "hnsw" = the traversal leg over the graph files in --graph-dir, "bf16" = the configs[3] leg
    p.add_argument("--pmc-child", default="", help=argparse.SUPPRESS)
    p.add_argument("--lib", default="", help="another build of libvelesdb_hip.so to measure (kernel-variant A / B runs; default: the in-tree library)")
    p.add_argument("--graph-dir", default="", help=argparse.SUPPRESS)
    p.add_argument("--no-latency-legs", action="store_true", help="skip the graph-path latency legs and the configs[0] leg")
    p.add_argument("--ef-curve", default="64,128,256,512", help="ef_search values of the recall / QPS curve of the graph legs")
    p.add_argument("--no-metrics-leg", action="store_true", help="skip the per-metric table (Euclidean / dot / Hamming / Jaccard sweeps)")
    p.add_argument("--no-bf16-leg", action="store_true", help="skip the bf16 GEMM-distance leg (BASELINE configs[3])")
    p.add_argument("--no-sq8-leg", action="store_true", help="skip the SQ8 storage-mode leg")
    p.add_argument("--bf16-rows", type=int, default=10_000_000)
    p.add_argument("--bf16-steps", type=int, default=5)
    p.add_argument("--bf16-cpu-rows", type=int, default=1_000_000, help="rows of the slice the CPU restatement scans")
    p.add_argument("--latent", type=int, default=32, help="latent factors of the embedding-like corpus")
    p.add_argument("--latent-noise", type=float, default=0.25)
    p.add_argument("--hnsw-batch", type=int, default=8192, help="queries per step (graph traversal)")
    p.add_argument("--hnsw-steps", type=int, default=5)
    p.add_argument("--ef", type=int, default=128)
    p.add_argument("--M", type=int, default=32)
    p.add_argument("--efc", type=int, default=400)
    p.add_argument("--recall-queries", type=int, default=1000)
    p.add_argument("--no-m128-leg", action="store_true", help="skip the graph leg at the reference's own preset for this corpus (HnswParams::million_scale)")
    p.add_argument("--m128-rows", type=int, default=1_000_000, help="rows of the million_scale(768) leg (M 128 / ef_construction 1600)")
    p.add_argument("--cpu-hnsw-queries", type=int, default=20000)
    return p.parse_args()

# ---- the line the driver parses -------------------------------------------------------------------------------------------------
COMPACT_LIMIT = 4096   # bytes; the driver's record keeps a bounded tail of stdout (round 4's 25 KB line was cut: BENCH_r04.parsed = null)
_ROOF_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel", "kernel_ms", "launches_timed")
_CPU_KEYS = ("value", "unit", "cores", "kind", "cpu_model", "host_hardware_threads", "cores_of_socket", "sample")


def _pick(d, keys, clip=160):
    out = {}
    for k_ in keys:
        if isinstance(d, dict) and k_ in d:
            v = d[k_]
            out[k_] = v[:clip] if isinstance(v, str) else v
    return out


def _ok(pc):
    """one bool out of a parity_check object: every boolean field true (None when there is no check)"""
    if not isinstance(pc, dict):
        return None
    flags = [v for v in pc.values() if isinstance(v, bool)]
    return bool(flags) and all(flags)


def compact_line(full, legs_file="bench_legs.json"):
    """Pure function: the full record of a run -> the ONE line printed last on stdout.  Holds the contract's keys, the headline's
    roofline and cpu_baseline objects (trimmed of prose) and one-number summaries of the other legs; everything else (tile tables, ef
    curves, caller tables, notes) lives in `legs_file`.  Always <= COMPACT_LIMIT bytes: strings are clipped, and if a
    pathological input still overflows, the leg summaries are dropped before the contract keys are."""
    g = full.get
    cfg = dict(g("config") or {})
    if isinstance(cfg.get("workload"), str):
        cfg["workload"] = cfg["workload"][:240]
    line = {k_: g(k_) for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    line["config"] = cfg
    line["recall_at_10"] = g("recall_at_10")
    line["parity_check"] = _ok(g("parity_check"))
    line["frac_step"] = g("frac_step")
    line["settle_steps"] = g("settle_steps")
    line["repeat_ms_per_step"] = g("repeat_ms_per_step")
    line["roofline"] = _pick(g("roofline") or {}, _ROOF_KEYS)
    cpu = _pick(g("cpu_baseline") or {}, _CPU_KEYS, clip=200)
    line["cpu_baseline"] = cpu if cpu else None
    if g("error"):
        line["error"] = str(g("error"))[:300]
    legs = {}
    h = g("hnsw")
    if isinstance(h, dict):
        r_ = h.get("roofline") or {}
        legs["hnsw"] = {"qps": h.get("qps"), "recall": h.get("recall_at_10"), "frac": r_.get("frac"), "bound": r_.get("bound"),
                        "kernel_ms": r_.get("kernel_ms"), "parity": _ok(h.get("parity_check")),
                        "cpu_qps": (h.get("cpu_baseline") or {}).get("value"),
                        "build_inserts_per_s": h.get("build_inserts_per_s"),
                        "build_frac": ((h.get("build") or {}).get("roofline") or {}).get("frac"),
                        "int8_frac": (h.get("int8") or {}).get("hbm_frac"), "int8_qps": (h.get("int8") or {}).get("qps")}
        lm = h.get("latency_mode")
        if isinstance(lm, list) and lm and isinstance(lm[0], dict):
            legs["hnsw"]["one_query_us"] = lm[0].get("median_us_per_call")
    he_ = g("host_entry")
    if isinstance(he_, dict):  # what a VectorIndex caller gets: host pointers in, host pointers out (PCIe-inclusive; never `value`)
        line["host_entry_qps"] = {str(c_.get("queries_per_call")): c_.get("qps") for c_ in (he_.get("calls") or [])[:3] if isinstance(c_, dict)}
        for t_ in (he_.get("threads") or [])[:3]:
            if isinstance(t_, dict):
                line["host_entry_qps"]["%sx%s" % (t_.get("threads"), t_.get("queries_per_call"))] = t_.get("qps")
    kc = g("k_curve")
    if isinstance(kc, list):
        line["k_curve_qps"] = {str(p_.get("k")): p_.get("qps") for p_ in kc[:6] if isinstance(p_, dict)}
        for p_ in kc:
            if isinstance(p_, dict) and p_.get("k") == 50:
                line["k50_qps"] = p_.get("qps")
    he2 = g("hnsw_embedding_like")
    if isinstance(he2, dict):
        legs["hnsw_embedding_like"] = {"qps": he2.get("qps"), "recall": he2.get("recall_at_10"), "frac": (he2.get("roofline") or {}).get("frac"),
                                       "cpu_qps": (he2.get("cpu_baseline") or {}).get("value")}
    hm = g("hnsw_m128")
    if isinstance(hm, dict):
        lm_ = hm.get("latency_mode")
        legs["hnsw_m128"] = {"qps": hm.get("qps"), "recall": hm.get("recall_at_10"), "frac": (hm.get("roofline") or {}).get("frac"),
                             "build_inserts_per_s": hm.get("build_inserts_per_s"), "build_frac": ((hm.get("build") or {}).get("roofline") or {}).get("frac"),
                             "cpu_qps": (hm.get("cpu_baseline") or {}).get("value"), "cpu_recall": (hm.get("cpu_baseline") or {}).get("recall_at_10"),
                             "one_query_us": (lm_[0].get("median_us_per_call") if isinstance(lm_, list) and lm_ and isinstance(lm_[0], dict) else None),
                             "parity": _ok(hm.get("parity_check"))}
    b = g("bf16_gemm")
    if isinstance(b, dict):
        r_ = b.get("roofline") or {}
        legs["bf16_gemm"] = {"qps": b.get("qps"), "frac": r_.get("frac"), "kernel_ms": r_.get("kernel_ms"),
                             "traffic_over_algorithmic": r_.get("traffic_over_algorithmic"), "parity": _ok(b.get("parity_check"))}
    s_ = g("sharded")
    if isinstance(s_, dict):
        legs["sharded"] = {"qps": s_.get("qps"), "group_ok": s_.get("group_ok"), "ranks": s_.get("ranks"),
                           "rows_per_shard": s_.get("rows_per_shard"), "transport": s_.get("transport"),
                           "identical_across_ranks": s_.get("results_identical_across_ranks"),
                           "error": (str(s_["error"])[:120] if s_.get("error") else None)}
    c0 = g("config0_10k")
    if isinstance(c0, dict):
        pts = ((c0.get("concurrent_callers") or {}).get("points")) or []
        legs["config0_10k"] = {"search_median_us": c0.get("search_median_us"),
                               "reference_published_us": (c0.get("reference_published") or {}).get("search_us"),
                               "gpu_over_cpu_by_threads": {str(p_.get("threads")): p_.get("gpu_over_cpu") for p_ in pts[:6]}}
    q8 = g("sq8_storage_mode")
    if isinstance(q8, dict):
        legs["sq8"] = {"batch_qps": (q8.get("batch") or {}).get("qps"), "eight_queries_hbm_frac": (q8.get("eight_queries") or {}).get("hbm_frac"),
                       "one_query_hbm_frac": (q8.get("one_query") or {}).get("hbm_frac"),
                       "parity": _ok(q8.get("parity_check"))}
    om = g("other_metrics")
    if isinstance(om, list):
        legs["other_metrics"] = {}
        for m_ in om[:6]:
            if not isinstance(m_, dict):
                continue
            sq, bt = m_.get("single_query") or {}, m_.get("batch") or {}
            legs["other_metrics"][str(m_.get("metric"))[:12]] = {
                "one_query_ms": sq.get("ms_per_call"), "one_query_hbm_frac": sq.get("hbm_frac"), "batch_qps": bt.get("qps"),
                "batch_frac": (bt.get("roofline") or {}).get("frac"), "batch_traffic_over_algorithmic": (bt.get("roofline") or {}).get("traffic_over_algorithmic"),
                "parity": _ok(m_.get("parity_check"))}
    line["legs"] = legs
    line["legs_file"] = legs_file
    line["device"] = (g("device") or "")[:80]
    enc = json.dumps(line)
    for drop in ("other_metrics", "config0_10k", "sq8", "sharded", "bf16_gemm", "hnsw_embedding_like", "hnsw_m128", "hnsw"):   # never reached on a real run (tested)
        if len(enc) <= COMPACT_LIMIT:
            break
        line["legs"].pop(drop, None)
        enc = json.dumps(line)
    if len(enc) > COMPACT_LIMIT:
        line["roofline"] = _pick(line["roofline"], _ROOF_KEYS, clip=40)
        line["cpu_baseline"] = _pick(line["cpu_baseline"] or {}, _CPU_KEYS, clip=40)
        line["config"] = {"workload": str(cfg.get("workload"))[:120]}
    return line


def main():
    a = parse()
    # stdout carries ONE JSON line and nothing else: native libraries (RCCL prints a version banner to the C stdout at
    # communicator creation) are pointed at stderr for the life of the process; the line is written to the saved fd
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # multi-process GPU work on this pool needs dmabuf IPC; the runtime reads the variable when it initialises (the first HIP call
    # below), so it is set before torch is imported, not next to init_process_group
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch  # first: the HIP runtime it loads is the one libvelesdb_hip.so binds to
    import torch.distributed as dist
    import numpy as np
    if a.lib:
        from velesdb_amd import _ffi as _vffi
        _vffi.use_library(os.path.abspath(a.lib))
    import velesdb_amd as va

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    def events_on_last(i, n):
        """Inside a timed region of n steps: the library records the HIP events that time a kernel (roofline.kernel_ms, read back as
        last_kernel_ms / last_selection_ms = the LAST step's launches) on step n - 1 only.  A record is not free: the GPU idles ~6 us on
        either side of a launch bracketed by events (profiles/r04z_headline_step_timeline.txt: 8 records = 47 us of a 1.75 ms headline
        step, 2.7 %), and the library's default — what a caller gets — is no events at all."""
        if i == n - 1:
            va.set_kernel_timing(True)

    # ---- the reference's calling pattern (concurrent `search` from many threads, one query per call): T native host
    # threads over the C ABI's HOST-pointer entry point vdb_hip_index_search (tools/callers_bench.cpp; Python threads would
    # measure the interpreter lock).  Every answer is compared bit for bit with a batched reference call.
    _cb = {}

    def callers_leg(index, q_np, k_, ef_, mode_, ref, threads_list=(1, 4, 16, 64), seconds=1.2, nq_per_call=1):
        import ctypes as C
        if "lib" not in _cb:
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "libcallers_bench.so")
            if not os.path.exists(path):
                return {"skipped": f"{path} missing (built by __graft_entry__.build())"}
            L = C.CDLL(path)
            L.callers_run.restype = C.c_int
            L.callers_run.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int,
                                      C.c_double, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            _cb["lib"] = L
        rid, rsc, rn = [np.ascontiguousarray(x) for x in ref]
        pts = []
        for T in threads_list:
            out = np.zeros(8, dtype=np.float64)
            s0 = index.combine_stats()
            rc = _cb["lib"].callers_run(index._h, q_np.ctypes.data, q_np.shape[0], q_np.shape[1], k_, ef_, mode_, T, seconds, 20, nq_per_call,
                                        rid.ctypes.data, rsc.ctypes.data, rn.ctypes.data, out.ctypes.data)
            s1 = index.combine_stats()
            assert rc == 0
            launches, calls = s1[0] - s0[0], s1[1] - s0[1]
            pts.append({"threads": T, "queries_per_call": nq_per_call, "qps": round(out[0], 1), "p50_us": round(out[1], 1), "p99_us": round(out[2], 1),
                        "calls": int(out[4]), "calls_per_launch": round(calls / max(launches, 1), 2),
                        "bitwise_mismatches_vs_batched_call": int(out[5]), "failed_calls": int(out[6])})
        return pts
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available() and va.device_count() > 0, "bench.py needs a GPU"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.dist_single
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    metric = {"cosine": va.DistanceMetric.Cosine, "euclidean": va.DistanceMetric.Euclidean,
              "dot": va.DistanceMetric.DotProduct}[a.metric]
    N, D, K, Q = a.rows, a.dim, a.k, a.batch

    if a.pmc_child in ("hnsw", "bf16", "bits_hamming", "bits_jaccard"):  # traffic pass of another leg: the same launches, nothing else
        stream_c = torch.cuda.current_stream().cuda_stream
        gc_ = torch.Generator(device=dev)
        if a.pmc_child.startswith("bits_"):  # the other_metrics leg's packed-bit batch (x > 0.5 of the same random rows / queries)
            mm_c = va.DistanceMetric.Hamming if a.pmc_child == "bits_hamming" else va.DistanceMetric.Jaccard
            ixc = va.HnswIndex(D, mm_c, va.HnswParams(a.M, a.efc, N), device=local)
            gc_.manual_seed(42)
            for base in range(0, N, 250_000):
                n_c = min(250_000, N - base)
                c = (torch.randn((n_c, D), generator=gc_, device=dev) > 0.5).float()
                torch.cuda.synchronize()
                ixc.upload_dev(base, c.data_ptr(), n_c, stream_c)
                torch.cuda.synchronize()
                del c
            nqc = Q
            gc_.manual_seed(43)
            qc = (torch.randn((nqc, D), generator=gc_, device=dev, dtype=torch.float32) > 0.5).float()
            c_ids = torch.empty((nqc, K), dtype=torch.int64, device=dev)
            c_sc = torch.empty((nqc, K), dtype=torch.float32, device=dev)
            c_n = torch.empty((nqc,), dtype=torch.int32, device=dev)
            for _ in range(a.warmup + a.steps):
                ixc.search_batch_dev(qc.data_ptr(), nqc, K, 0, va.MODE_BRUTE, c_ids.data_ptr(), c_sc.data_ptr(), c_n.data_ptr(), stream_c)
            torch.cuda.synchronize()
            return
        if a.pmc_child == "hnsw":
            ixc = va.HnswIndex(D, metric, va.HnswParams(a.M, a.efc, N), device=local)
            ixc.load_reference_files(a.graph_dir, "native_hnsw")
            nqc = a.hnsw_batch
            mode_c, ef_c = va.MODE_HNSW, a.ef
        else:
            ixc = va.HnswIndex(D, metric, va.HnswParams(a.M, a.efc, a.bf16_rows), device=local)
            ixc.enable_bf16()
            gc_.manual_seed(45)
            for base in range(0, a.bf16_rows, 1_000_000):
                n_c = min(1_000_000, a.bf16_rows - base)
                c = torch.randn((n_c, D), generator=gc_, device=dev)
                torch.cuda.synchronize()
                ixc.upload_dev(base, c.data_ptr(), n_c, stream_c)
                del c
            nqc = 1024
            mode_c, ef_c = va.MODE_BRUTE_BF16, 0
        gc_.manual_seed(43)
        qc = torch.randn((nqc, D), generator=gc_, device=dev, dtype=torch.float32)
        c_ids = torch.empty((nqc, K), dtype=torch.int64, device=dev)
        c_sc = torch.empty((nqc, K), dtype=torch.float32, device=dev)
        c_n = torch.empty((nqc,), dtype=torch.int32, device=dev)
        for _ in range(a.warmup + a.steps):
            ixc.search_batch_dev(qc.data_ptr(), nqc, K, ef_c, mode_c, c_ids.data_ptr(), c_sc.data_ptr(), c_n.data_ptr(), stream_c)
        torch.cuda.synchronize()
        return

    # ---- synthetic corpus, generated on the device (same seed on every rank = replica) ----
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    corpus = torch.randn((N, D), generator=g, device=dev, dtype=torch.float32)
    g.manual_seed(43)
    n_query_pool = max(Q * 4, 256, 0 if a.no_hnsw else a.hnsw_batch)
    queries = torch.randn((n_query_pool, D), generator=g, device=dev, dtype=torch.float32)
    ix = va.HnswIndex(D, metric, va.HnswParams(a.M, a.efc, N), device=local)
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    va.set_max_query_tile(a.tile)
    va.set_sweep_engine(a.engine)
    va.set_split_selector(0 if a.no_split else a.select_level)
    ix.upload_dev(0, corpus.data_ptr(), N, stream)
    sample_rows = min(a.cpu_sample_rows, N)
    # (the host copies of the corpus the oracle checks need are taken BEHIND the headline's timed region: 3 GB over PCIe leave the GPU
    # idle for ~0.3 s, and W warm-up steps = 7.5 ms do not bring a chip back from that — round 6 measured the same 20 steps 2.5 %
    # faster when repeated at once, `repeat_ms_per_step`)

    out_ids = torch.empty((Q, K), dtype=torch.int64, device=dev)
    out_sc = torch.empty((Q, K), dtype=torch.float32, device=dev)
    out_n = torch.empty((Q,), dtype=torch.int32, device=dev)

    def step(i, mode=va.MODE_BRUTE):
        # each rank takes its own slice of the query stream (replica mode)
        off = ((i * world + rank) * Q) % (n_query_pool - Q + 1)
        ix.search_batch_dev(queries[off:off + Q].data_ptr(), Q, K, 0, mode, out_ids.data_ptr(),
                            out_sc.data_ptr(), out_n.data_ptr(), stream)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def gather_rows(vals, names):
        """one row of floats per rank -> list of dicts on every rank (rank order)"""
        mine_ = torch.tensor(vals, device=dev, dtype=torch.float64)
        rows_ = [torch.zeros_like(mine_) for _ in range(world)] if use_dist else [mine_]
        if use_dist:
            dist.all_gather(rows_, mine_)
        return [{n_: (int(v) if n_ in ("rank", "rows") else round(float(v), 4)) for n_, v in zip(names, r_.tolist())} for r_ in rows_]

    def max_over_ranks(x):
        if use_dist:
            t = torch.tensor([x], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # The chip does not run this kernel mix at its settled rate from the first call: measured per step from idle (0.3 s or 3 s alike, or
    # behind the upload's HBM-bound kernels) 1.75, 1.57, 1.53 ms per step over the first 15 steps against 1.47 settled
    # (profiles/r06s_step_settle_probe.log) — a serving process is in the settled state, a 20-step region behind 5 warm-up steps is not.
    # So `settle_steps` untimed steps of the same workload run IN FRONT of the W warm-up steps; they are reported in the line, W and K
    # are what the caller asked for, and `repeat_ms_per_step` (the same K steps once more) still shows what is left of the effect.
    # (one-time set-up of the kernel timing the LAST timed step switches on — the library creates its HIP events at first use and
    # the runtime turns the queue's profiling on at the first timed record: 1.8 ms inside the 30-ms timed window on one box of the
    # pool, twice, `profiles/r06sel3_*` / `r06ae_*` — is paid here, on an untimed step)
    va.set_kernel_timing(True)
    step(0)
    va.set_kernel_timing(False)
    for i in range(max(0, a.settle_steps)):
        step(i)
    for i in range(a.warmup):
        step(i)
    barrier()
    if a.pmc_child:  # "headline" traffic pass (run under rocprofv3 --pmc FETCH_SIZE by the parent): the headline steps, nothing else
        for i in range(a.steps):
            step(a.warmup + i)
        torch.cuda.synchronize()
        return
    if use_dist:  # RCCL's version banner sits in the C stdout buffer of every rank: push it out now, not after the JSON line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
    t0 = time.perf_counter()
    for i in range(a.steps):
        events_on_last(i, a.steps)
        step(a.warmup + i)
    torch.cuda.synchronize()
    dt_mine = time.perf_counter() - t0  # this rank's own steps (the timed region below ends behind the barrier: the slowest rank's)
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, kernel_launches = ix.last_kernel_ms()  # HIP events around the sweep kernel, last step
    sel_ms, sel_launches = ix.last_selection_ms()      # HIP events around every launch of the selection kernel, last step
    va.set_kernel_timing(False)
    dt = max_over_ranks(dt)
    # (diagnostic, not `value`: the same K steps once more, right behind the timed region — how much of ms_per_step is the chip still
    # settling into the workload behind W warm-up steps)
    t0r = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + a.steps + i)
    torch.cuda.synchronize()
    repeat_ms_per_step = (time.perf_counter() - t0r) / a.steps * 1e3
    host_sample = corpus[:sample_rows].cpu().numpy() if rank == 0 else None
    host_full = None
    if rank == 0 and a.check_queries > 0:
        host_full = host_sample if sample_rows == N else corpus.cpu().numpy()
    del corpus
    torch.cuda.empty_cache()
    qps = world * Q * a.steps / dt
    replicas_per_rank = gather_rows([float(rank), Q * a.steps / dt_mine], ("rank", "qps"))

    # ---- roofline of the dominant kernel (the sweep): algorithmic bytes / measured duration ----
    mfma = a.engine == 1 and a.metric in ("cosine", "dot")

    F32_MFMA_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak of MI355X (cdna_hip_programming.md section 3)

    def gemm_plan(nq):  # sweep_gemm_plan (sweep_gemm.hip): query tiles of <= 128, 16*nqf queries per wave
        nqt = (nq + 127) // 128
        qper = (nq + nqt - 1) // nqt
        return nqt, qper, max(2, (qper + 31) // 32)

    def tile_for(nq, max_tile, use_mfma):  # the library's tile choice (index.hip brute_dev)
        if use_mfma:
            if nq >= 64 and max_tile >= 128:
                return nq  # GEMM kernel: the whole batch in one launch
            return 48 if (nq > 32 and max_tile >= 48) else (32 if (nq > 16 and max_tile >= 32) else 16)
        lds_tiles = D % 256 == 0 and D <= 1024
        if lds_tiles and nq >= 24 and max_tile >= 32:
            return 32
        if lds_tiles and nq >= 12 and max_tile >= 16:
            return 16
        return min(8 if nq >= 8 else (4 if nq >= 4 else (2 if nq >= 2 else 1)), max_tile, 8)

    def kernel_name(t, use_mfma):
        if use_mfma:
            if t >= 64:
                return f"sweep_topk_gemm_f32<{a.metric},NQF={gemm_plan(t)[2]}>"
            return f"sweep_topk_mfma_f32<{a.metric},NQT={t // 16}>"
        cpl = D // 256 if D % 256 == 0 and D <= 1024 else 0
        return (f"sweep_topk_f32_qlds<{a.metric},B={t},CPL={cpl}>" if t >= 16
                else f"sweep_topk_f32<{a.metric},B={t},CPL={cpl}>")

    def alg_bytes_for(t):  # N*D*4 (+ N*4 precomputed norms) per corpus pass + the query tile
        return N * D * 4 + (N * 4 if a.metric == "cosine" else 0) + t * D * 4

    tile = tile_for(Q, a.tile, mfma)
    alg_bytes = alg_bytes_for(tile)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    flops = 2.0 * N * D * tile  # algorithmic: one multiply-add per (row, query, dimension)
    tflops = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md)
    # select_stage.hip split_path_ok: large exact cosine / dot batches select on the bf16 matrix cores (sweep_split.hip)
    split_active = (mfma and not a.no_split and a.select_level > 0 and a.tile >= 128 and Q >= 224 and Q * 8 >= ((Q + 255) // 256) * 256 * 7
                    and K <= 10 and N >= 65536 and D % 32 == 0 and D >= 64)

    def exact_roofline(kms, nl):
        tf = flops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        return {"bound": "mfma", "achieved": round(tf, 1), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "kernel": kernel_name(tile, mfma), "kernel_ms": round(kms, 4),
                "launches_timed": nl, "alg_flops_per_launch": flops,
                "queries_per_launch": tile, "alg_bytes_per_launch": alg_bytes,
                "hbm_gbs": round(alg_bytes / (kms * 1e-3) / 1e9, 1) if kms > 0 else 0.0,
                "note": "exact f32 contraction on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: 157.3 TFLOP/s "
                        "dense peak, 1/16 of the bf16 rate); algorithmic flop = 2*rows*dim*queries"}

    exact_leg = None
    if split_active:
        # exact f32 RESULTS (ids, ranks, score bits of the exact kernel; checked below against the oracle), produced by a
        # split-bf16 selection on the bf16 matrix cores + exact re-scoring + per-query proof.  The timed region (HIP
        # events) covers the whole batch: exact seed sweep, two selection launches, merges, re-scoring / proof, the
        # device-driven fallback launch.  achieved = ALGORITHMIC flop (2*rows*dim*queries) / that time against the bf16
        # dense peak; the selection issues 3 bf16 MFMAs per algorithmic product (hi.hi + hi.lo + lo.hi).
        nq_last, unproven = ix.last_split_stats()
        level = ix.last_select_level()
        mfmas_per_product = 3.0 if level == 1 else 1.0
        sel_kernel = (f"sweep_topk_gemm_bf16_glds<{a.metric},SPLIT>" if level == 1 else
                      (f"sweep_topk_gemm_bf16_pp<dot, WIDE> over normalised bf16 images of rows and queries ({a.metric}; ping-pong pipeline, candidates to global lists)"
                       if level == 4 else f"sweep_topk_gemm_bf16_pp<{a.metric}> (plain bf16 selection, ping-pong pipeline)"))
        # the dominant kernel = the selection kernel: its launches of one step sweep every row behind the seed prefix once
        # (algorithmic flop 2 * rows * dim * queries per step), their durations are summed from HIP events around each launch
        sel_tf = (2.0 * (N - 4096) * D * tile) / (sel_ms * 1e-3) / 1e12 if sel_ms > 0 else 0.0
        roofline = {"bound": "mfma", "achieved": round(sel_tf, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(sel_tf / BF16_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                    "kernel": sel_kernel,
                    "kernel_ms": round(sel_ms, 4), "launches_timed": sel_launches,
                    "kernel_ms_note": "sum over the step's launches of the selection kernel (growing row ranges, thresholds re-seeded in between); "
                                      "HIP events on the last of the timed steps only (a record idles the GPU ~6 us either side of the launch)",
                    "select_level": level,
                    "alg_flops_per_launch": 2.0 * (N - 4096) * D * tile, "queries_per_launch": tile, "alg_bytes_per_launch": alg_bytes,
                    "mfmas_per_product": mfmas_per_product,
                    "matrix_pipe_utilisation": round(mfmas_per_product * sel_tf / BF16_MFMA_PEAK_TFLOPS, 4),
                    "whole_batch": {"ms": round(kernel_ms, 4), "achieved": round(tflops, 1), "frac": round(tflops / BF16_MFMA_PEAK_TFLOPS, 4),
                                    "note": "seed sweep + selection launches + merges + exact re-scoring / proof + fallback check: "
                                            "algorithmic flop (2*rows*dim*queries) / the batch's HIP-event time / 2.5 PFLOP/s"},
                    "unproven_queries_last_batch": unproven, "queries_last_batch": nq_last,
                    "hbm_gbs": round(achieved, 1), "hbm_frac": round(achieved / HBM_PEAK_GBS, 4),
                    "note": "exact f32 RESULTS (bit-identical to exact_f32_kernel) from a selection on the bf16 matrix cores "
                            "(level 2: one MFMA per product over the bf16 copy of the rows; level 1: split-bf16, three) + "
                            "exact re-scoring + per-query proof; frac = the selection kernel's algorithmic flop / its launches' "
                            "time / 2.5 PFLOP/s (bf16 dense); the chip runs this kernel at ~1.93 GHz (profiles/r02s_pmc_clock_*)"}
        va.set_split_selector(False)
        for i in range(2):
            step(i)
        torch.cuda.synchronize()
        te = time.perf_counter()
        ne = max(2, min(a.steps, 5))
        for i in range(ne):
            events_on_last(i, ne)
            step(a.warmup + i)
        torch.cuda.synchronize()
        e_dt = (time.perf_counter() - te) / ne
        e_kms, e_nl = ix.last_kernel_ms()
        va.set_kernel_timing(False)
        va.set_split_selector(a.select_level)
        exact_leg = {"qps": round(Q / e_dt, 1), "ms_per_step": round(e_dt * 1e3, 4), "roofline": exact_roofline(e_kms, e_nl)}
        roofline["exact_f32_kernel"] = exact_leg
    elif mfma and tile >= 64:
        # the GEMM-structured kernel serves the whole batch with one corpus pass: bound by the exact-f32 matrix pipe
        roofline = exact_roofline(kernel_ms, kernel_launches)
        roofline["hbm_frac"] = round(achieved / HBM_PEAK_GBS, 4)
    else:
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "kernel": kernel_name(tile, mfma),
                    "kernel_ms": round(kernel_ms, 4), "launches_timed": kernel_launches,
                    "alg_bytes_per_launch": alg_bytes, "queries_per_launch": tile,
                    "f32_tflops": round(tflops, 1),
                    "note": "one corpus pass serves `queries_per_launch` queries, so HBM bytes per QUERY are "
                            "alg_bytes/queries_per_launch; `tiles` lists every tile size"}

    # HBM traffic, measured in THIS run: rank 0 re-runs a leg's launches in a child process under `rocprofv3 --pmc FETCH_SIZE`
    # (its own pass, counters only + kernel trace; MI355X_MICROARCH.md HBM section: the counter is in KiB and reports half of
    # the bytes of wide coalesced reads on gfx950 => x 1024 x 2) and sums the kernels named in `wanted`.
    def traffic_pass(mode, wanted, extra, child_steps=3, child_warm=1, timeout=600):
        """-> (bytes per step, {kernel: bytes per step}, source text); bytes is None when the pass failed"""
        import csv
        import glob
        import subprocess
        tdir = tempfile.mkdtemp(prefix="vdb_bench_pmc_")
        try:
            cmd = ["rocprofv3", "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", tdir, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", mode, "--steps", str(child_steps), "--warmup", str(child_warm), "--settle-steps", "0",
                   "--rows", str(N), "--dim", str(D), "--k", str(K), "--batch", str(Q), "--metric", a.metric,
                   "--tile", str(a.tile), "--engine", str(a.engine), "--select-level", str(a.select_level)] + (["--no-split"] if a.no_split else []) + extra
            pr = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(tdir, "**", "*counter_collection.csv"), recursive=True)
            if pr.returncode != 0 or not files:
                return None, {}, "traffic pass failed: " + (pr.stderr or "")[-200:]
            per_kernel = {}
            for r in csv.DictReader(open(files[0])):
                if r.get("Counter_Name") != "FETCH_SIZE":
                    continue
                name = r["Kernel_Name"]
                if not any(t in name for t in wanted):
                    continue  # corpus upload / conversion, torch kernels
                short = name.split("(")[0].replace("void ", "")
                per_kernel[short] = per_kernel.get(short, 0.0) + float(r["Counter_Value"]) * 1024.0 * 2.0
            nsteps = child_steps + child_warm
            return (round(sum(per_kernel.values()) / nsteps), {k2: round(v / nsteps) for k2, v in sorted(per_kernel.items(), key=lambda kv: -kv[1])[:4]},
                    f"rocprofv3 --pmc FETCH_SIZE child pass of this run ({nsteps} steps of the leg), bytes per step, x2 gfx950 correction applied")
        except Exception as e:  # noqa: BLE001 - the traffic pass is diagnostic: never fail the bench line for it
            return None, {}, f"traffic pass failed: {e!r}"[:300]
        finally:
            shutil.rmtree(tdir, ignore_errors=True)

    if rank == 0 and world == 1 and not a.no_traffic_pass:
        tb_, tk_, tsrc_ = traffic_pass("headline", ("sweep_topk", "merge_topk", "split_rerank", "split_seed", "split_reseed", "seed_scores_bf16", "sel16_prep_queries",
                                                "l2_seed", "select_finish", "wide_seed", "wide_reseed", "wide_rerank", "seln_prep_queries"), [])
        roofline["traffic_source"] = tsrc_
        if tb_ is not None:
            roofline["traffic"] = tb_
            roofline["traffic_by_kernel"] = tk_
            roofline["traffic_over_algorithmic"] = round(tb_ / alg_bytes, 3)

    tiles = []
    if rank == 0 and not a.no_tiles:
        va.set_split_selector(False)  # the table shows the exact kernels at every batch size
        plans = [(1, t) for t in (16, 32, 48, 64, 96, 128, 192, 256)] if a.metric in ("cosine", "dot") else []
        plans = [(1, 1)] + plans if plans else plans
        plans += [(0, t) for t in (1, 8, 16, 32)]
        for eng, t in plans:
            if t > a.tile and not (eng == 1 and (t == 1 or (t >= 64 and a.tile >= 128))):
                continue
            if t > n_query_pool:
                continue
            use_m = eng == 1
            va.set_sweep_engine(eng)
            mt = (128 if t >= 64 else max(t, 16)) if use_m else t
            va.set_max_query_tile(mt if mt in (1, 2, 4, 8, 16, 32, 48, 128) else 48)
            nq_t = t
            eff = tile_for(nq_t, mt, use_m)
            t_ids = torch.empty((nq_t, K), dtype=torch.int64, device=dev)
            t_sc = torch.empty((nq_t, K), dtype=torch.float32, device=dev)
            t_n = torch.empty((nq_t,), dtype=torch.int32, device=dev)
            for _ in range(2):
                ix.search_batch_dev(queries.data_ptr(), nq_t, K, 0, va.MODE_BRUTE, t_ids.data_ptr(),
                                    t_sc.data_ptr(), t_n.data_ptr(), stream)
            torch.cuda.synchronize()
            reps = 10
            tt = time.perf_counter()
            for r_ in range(reps):
                events_on_last(r_, reps)
                ix.search_batch_dev(queries.data_ptr(), nq_t, K, 0, va.MODE_BRUTE, t_ids.data_ptr(),
                                    t_sc.data_ptr(), t_n.data_ptr(), stream)
            torch.cuda.synchronize()
            t_dt = (time.perf_counter() - tt) / reps
            kms, nl = ix.last_kernel_ms()
            va.set_kernel_timing(False)
            gbs = alg_bytes_for(eff) / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
            tf = 2.0 * N * D * min(eff, nq_t) / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
            tiles.append({"engine": "mfma" if use_m else "valu", "queries": nq_t, "kernel": kernel_name(eff, use_m),
                          "kernel_ms": round(kms, 4), "hbm_gbs": round(gbs, 1),
                          "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "f32_tflops": round(tf, 1),
                          "qps": round(nq_t / t_dt, 1)})
        va.set_sweep_engine(a.engine)
        va.set_max_query_tile(a.tile)
        va.set_split_selector(0 if a.no_split else a.select_level)

    # ---- the same batch sizes through the DEFAULT path (from 16 queries up: the selection stage; same result bits) ----
    batch_sizes = []
    if rank == 0 and not a.no_tiles and not a.no_split and a.metric in ("cosine", "dot"):
        for nq_t in (16, 32, 64, 128, 256, 384, 512):
            if nq_t > n_query_pool:
                continue
            t_ids = torch.empty((nq_t, K), dtype=torch.int64, device=dev)
            t_sc = torch.empty((nq_t, K), dtype=torch.float32, device=dev)
            t_n = torch.empty((nq_t,), dtype=torch.int32, device=dev)
            for _ in range(2):
                ix.search_batch_dev(queries.data_ptr(), nq_t, K, 0, va.MODE_BRUTE, t_ids.data_ptr(), t_sc.data_ptr(), t_n.data_ptr(), stream)
            torch.cuda.synchronize()
            reps = 10
            tt = time.perf_counter()
            for _ in range(reps):
                ix.search_batch_dev(queries.data_ptr(), nq_t, K, 0, va.MODE_BRUTE, t_ids.data_ptr(), t_sc.data_ptr(), t_n.data_ptr(), stream)
            torch.cuda.synchronize()
            t_dt = (time.perf_counter() - tt) / reps
            batch_sizes.append({"queries": nq_t, "ms_per_call": round(t_dt * 1e3, 4), "qps": round(nq_t / t_dt, 1),
                                "select_level": int(ix.last_select_level())})

    # ---- the headline batch at the k the reference also benches (benches/hnsw_benchmark.rs:152-159: k = 10, 50, 100; its brute-force /
    # Perfect / rerank calls take any k, search.rs:118-160,176-219): device-resident, Q queries per step.  k = 10: the selection stage
    # with block-local lists; 10 < k <= 128: the WIDE selection (csrc/sweep_wide.hip); same exact bits either way (tests/test_gpu_wide_k.py)
    k_curve = None
    if rank == 0 and not a.no_tiles and not a.no_split and a.metric in ("cosine", "dot"):
        k_curve = []
        for k_c in (10, 11, 50, 100):
            kc_ids = torch.empty((Q, k_c), dtype=torch.int64, device=dev)
            kc_sc = torch.empty((Q, k_c), dtype=torch.float32, device=dev)
            kc_n = torch.empty((Q,), dtype=torch.int32, device=dev)

            def kstep(i_):
                off_ = (i_ * Q) % (n_query_pool - Q + 1)
                ix.search_batch_dev(queries[off_:off_ + Q].data_ptr(), Q, k_c, 0, va.MODE_BRUTE, kc_ids.data_ptr(), kc_sc.data_ptr(), kc_n.data_ptr(), stream)
            for i_ in range(3):
                kstep(i_)
            torch.cuda.synchronize()
            reps = 10
            tk0 = time.perf_counter()
            for i_ in range(reps):
                events_on_last(i_, reps)
                kstep(3 + i_)
            torch.cuda.synchronize()
            kdt = (time.perf_counter() - tk0) / reps
            ksel_ms, ksel_n = ix.last_selection_ms()
            va.set_kernel_timing(False)
            nq_k, unp_k = ix.last_split_stats()
            k_curve.append({"k": k_c, "queries": Q, "ms_per_step": round(kdt * 1e3, 4), "qps": round(Q / kdt, 1), "select_level": int(ix.last_select_level()),
                            "selection_kernel_ms": round(ksel_ms, 4), "selection_launches": ksel_n,
                            "selection_frac_of_bf16_pipe": round(2.0 * N * D * Q / (ksel_ms * 1e-3) / 1e12 / 2500.0, 4) if ksel_ms > 0 else None,
                            "unproven_queries_last_batch": unp_k})
        if host_full is not None and a.check_queries > 0:   # parity of the k = 50 and k = 100 answers: ids + score bits against the oracle
            from oracle import pyoracle as po_k
            om_k = {"cosine": po_k.COSINE, "dot": po_k.DOT}[a.metric]
            for pt in k_curve:
                if pt["k"] not in (50, 100):
                    continue
                qk = queries[:Q].cpu().numpy()
                gi_k, gs_k, _ = ix.search_batch_brute_force(qk, pt["k"])
                pick = np.unique(np.linspace(0, Q - 1, 16).astype(np.int64))
                ei_k, es_k = po_k.scan_topk(om_k, host_full, qk[pick], pt["k"], po_k.MODE_M if ix.sweep_arith_mode(pt["k"]) == "M" else po_k.MODE_C,
                                            nthreads=po_k.host_threads())
                pt["parity_check"] = {"queries": int(pick.size), "ids_equal_oracle": bool(np.array_equal(gi_k[pick], ei_k)),
                                      "scores_bit_equal_oracle": bool(np.array_equal(gs_k[pick].view(np.uint32), es_k.view(np.uint32)))}

    # ---- the HOST-pointer entry points on the headline workload (PCIe-inclusive; never `value`): what the shim's
    # VectorIndex::search / search_batch_parallel bind (velesdb-hip/src/lib.rs) — queries from host memory, results back to host
    # memory — and the reference's calling pattern, many threads x one query per call, through the combining front
    host_entry = None
    if rank == 0 and world == 1 and not a.no_tiles:  # (--no-tiles: profiling passes of the headline step alone)
        hq_np = queries[:min(n_query_pool, 4096)].cpu().numpy()
        he = []
        for nq_h in (Q, 256, 64, 16, 1):
            if nq_h > hq_np.shape[0]:
                continue
            for _ in range(2):
                ix._search_raw(hq_np[:nq_h], K, 0, va.MODE_BRUTE)
            reps = 10
            th_ = time.perf_counter()
            for r_ in range(reps):
                ix._search_raw(hq_np[(r_ * nq_h) % (hq_np.shape[0] - nq_h + 1):][:nq_h], K, 0, va.MODE_BRUTE)
            hd_ = (time.perf_counter() - th_) / reps
            he.append({"queries_per_call": nq_h, "ms_per_call": round(hd_ * 1e3, 4), "qps": round(nq_h / hd_, 1),
                       "h2d_bytes": nq_h * D * 4, "d2h_bytes": nq_h * (K * 12 + 4)})
        b_ref = ix._search_raw(hq_np, K, 0, va.MODE_BRUTE)
        host_entry = {"entry_point": "vdb_hip_index_search_batch (host pointers), exact sweep over the headline corpus", "calls": he,
                      "concurrent_callers": callers_leg(ix, hq_np, K, 0, va.MODE_BRUTE, b_ref)}
        # T threads, each issuing whole Q-query batches from host memory (search_batch_parallel from several request handlers): every
        # call leases its own search context and stream, so one call's copies run beside another's kernels
        if hq_np.shape[0] >= Q:
            ix.set_option(va.OPT_COMBINE_MAX_BATCH, 0)   # (whole batches: nothing to combine)
            thr = callers_leg(ix, hq_np, K, 0, va.MODE_BRUTE, b_ref, threads_list=(1, 2, 4), seconds=0.8, nq_per_call=Q)
            ix.set_option(va.OPT_COMBINE_MAX_BATCH, -1)
            host_entry["threads"] = thr if isinstance(thr, list) else [thr]

    # ---- single-query latency mode (one corpus pass per query) ----
    lat = {}
    if rank == 0:
        torch.cuda.synchronize()
        reps = 20
        t1 = time.perf_counter()
        for i in range(reps):
            events_on_last(i, reps)
            ix.search_batch_dev(queries[i:i + 1].data_ptr(), 1, K, 0, va.MODE_BRUTE, out_ids.data_ptr(),
                                out_sc.data_ptr(), out_n.data_ptr(), stream)
        torch.cuda.synchronize()
        l_dt = (time.perf_counter() - t1) / reps
        kms, _ = ix.last_kernel_ms()
        va.set_kernel_timing(False)
        b1 = alg_bytes_for(16 if mfma else 1)  # the matrix-core streaming kernel stages a 16-query tile
        lat = {"ms_per_query": round(l_dt * 1e3, 4), "qps": round(1.0 / l_dt, 1), "sweep_kernel_ms": round(kms, 4),
               "hbm_gbs": round(b1 / (kms * 1e-3) / 1e9, 1) if kms > 0 else 0.0,
               "hbm_frac": round(b1 / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if kms > 0 else 0.0}

    # ---- graph leg (configs[2]): GPU construction + traversal kernel ----
    hnsw = None
    graph_dir = None
    if not a.no_hnsw:
        HQ = min(a.hnsw_batch, n_query_pool)
        h_ids = torch.empty((HQ, K), dtype=torch.int64, device=dev)
        h_sc = torch.empty((HQ, K), dtype=torch.float32, device=dev)
        h_n = torch.empty((HQ,), dtype=torch.int32, device=dev)
        barrier()
        tb = time.perf_counter()
        ix.build_graph(0)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - tb
        # construction roofline: the insert kernel counts the rows whose distance it evaluates (search_layer at
        # ef_construction + select_neighbors, graph.rs:158-237, 526-581); algorithmic bytes = rows x dim x 4 (random 3-KB gathers), over
        # the WALL time of build_graph (insert + sort + link kernels and the host's batch loop: the insert kernel is ~97 % of it,
        # profiles/r05*_build_kernel_stats.csv)
        b_rows, b_phases, b_nodes, b_sel = ix.build_stats()
        # (the rows of select_neighbors are the node's own <= ef_construction candidates, evaluated again against every selected
        # neighbour: they come out of L2, not out of HBM — the fraction is quoted on the search_layer rows alone, the total beside it)
        build_bytes = (b_rows - b_sel) * D * 4
        build_roof = {"bound": "hbm", "achieved": round(build_bytes / build_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": round(build_bytes / build_s / 1e9 / HBM_PEAK_GBS, 4), "alg_bytes": build_bytes,
                      "frac_with_select_neighbors_rereads": round(b_rows * D * 4 / build_s / 1e9 / HBM_PEAK_GBS, 4),
                      "rows_evaluated_per_insert": round(b_rows / max(b_nodes, 1), 1),
                      "select_neighbors_rows_per_insert": round(b_sel / max(b_nodes, 1), 1),
                      "distance_phases_per_insert": round(b_phases / max(b_nodes, 1), 1), "nodes": b_nodes, "seconds": round(build_s, 2),
                      "kernel": "hnsw_insert_kernel + hnsw_link_kernel + radix sort of the link requests (wall time of build_graph)",
                      "alg_bytes_rule": "rows whose distance to the NEW node the insert kernel evaluated (search_layer at ef_construction) x dim x 4; counters from the kernel"}

        def hstep():
            ix.search_batch_dev(queries.data_ptr(), HQ, K, a.ef, va.MODE_HNSW, h_ids.data_ptr(), h_sc.data_ptr(),
                                h_n.data_ptr(), stream)

        hstep()
        barrier()
        th = time.perf_counter()
        for i_ in range(a.hnsw_steps):
            events_on_last(i_, a.hnsw_steps)
            hstep()
        barrier()
        hdt_mine = time.perf_counter() - th
        hdt = max_over_ranks(hdt_mine)
        hk_ms, hk_n = ix.last_kernel_ms()
        va.set_kernel_timing(False)
        n_dist, n_expand = ix.last_search_stats()  # of the last batch
        hbytes = n_dist * D * 4 + n_expand * 2 * a.M * 4
        hgbs = hbytes / (hk_ms * 1e-3) / 1e9 if hk_ms > 0 else 0.0
        # recall@10 against the exact top-k (the sweep, itself bit-checked against the oracle)
        RQ = min(a.recall_queries, HQ)
        rq = queries[:RQ].cpu().numpy()
        gt, _, _ = ix.search_batch_brute_force(rq, K)
        hi = h_ids[:RQ].cpu().numpy()
        recall_h = float(np.mean([len(set(hi[i].tolist()) & set(gt[i].tolist())) / K for i in range(RQ)]))
        hnsw = {"workload": f"{N}x{D} f32 {a.metric}, HNSW graph in HBM (M={a.M}, M0={2 * a.M}, ef_construction={a.efc}, "
                            f"built on the GPU), k={K}, ef={a.ef}, {HQ} queries/step (BASELINE configs[2])",
                "qps": round(world * HQ * a.hnsw_steps / hdt, 1), "ms_per_step": round(hdt / a.hnsw_steps * 1e3, 3),
                "recall_at_10": round(recall_h, 4), "recall_queries": RQ,
                "build_seconds": round(build_s, 2), "build_inserts_per_s": round(N / build_s, 1), "build": {"roofline": build_roof},
                "n_dist_per_query": round(n_dist / HQ, 1), "n_expand_per_query": round(n_expand / HQ, 1),
                "roofline": {"bound": "hbm", "achieved": round(hgbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(hgbs / HBM_PEAK_GBS, 4), "traffic": None,
                             "kernel": f"hnsw_search_kernel<{a.metric},CPL={D // 256 if D % 256 == 0 else 0}>",
                             "kernel_ms": round(hk_ms, 4), "launches_timed": hk_n,
                             "alg_bytes_per_launch": hbytes,
                             "alg_bytes_rule": "n_dist*dim*4 + n_expand*M0*4, counters from the kernel"}}
        # the replica leg of the graph path (the 3.3 GB index fits every GPU; the query stream is split, no collective):
        # every rank's own rate and recall, so that a SCALE run shows the curve per GPU and not only the aggregate
        hnsw["replicas"] = {"ranks": world, "parallelism": "replicas x%d (every rank holds the whole graph and answers its own %d queries per step; no collective)" % (world, HQ),
                            "per_rank": gather_rows([float(rank), HQ * a.hnsw_steps / hdt_mine, recall_h, float(ix.len())],
                                                    ("rank", "qps", "recall_at_10", "rows"))}
        # recall / QPS curve over ef_search (SearchQuality presets Fast 64 / Balanced 128 / ... , params.rs:309-319): the
        # "QPS @ recall@10" metric as a curve; the CPU baseline fills in its side of every point below
        ef_list = [int(x) for x in a.ef_curve.split(",") if x.strip()]

        def gpu_ef_curve(index, q_t, gt_ids, nq_c, RQc):
            pts = []
            c_ids = torch.empty((nq_c, K), dtype=torch.int64, device=dev)
            c_sc = torch.empty((nq_c, K), dtype=torch.float32, device=dev)
            c_n = torch.empty((nq_c,), dtype=torch.int32, device=dev)
            for ef_c in ef_list:
                def cstep():
                    index.search_batch_dev(q_t.data_ptr(), nq_c, K, ef_c, va.MODE_HNSW, c_ids.data_ptr(), c_sc.data_ptr(),
                                           c_n.data_ptr(), stream)
                cstep()
                torch.cuda.synchronize()
                tc0 = time.perf_counter()
                for i_ in range(2):
                    events_on_last(i_, 2)
                    cstep()
                torch.cuda.synchronize()
                cdt_ = (time.perf_counter() - tc0) / 2
                ck_ms, _ = index.last_kernel_ms()
                va.set_kernel_timing(False)
                c_nd, c_ne = index.last_search_stats()
                cbytes = c_nd * D * 4 + c_ne * 2 * a.M * 4
                ci_ = c_ids[:RQc].cpu().numpy()
                rec_ = float(np.mean([len(set(ci_[i].tolist()) & set(gt_ids[i].tolist())) / K for i in range(RQc)]))
                pts.append({"ef": ef_c, "qps": round(nq_c / cdt_, 1), "ms_per_step": round(cdt_ * 1e3, 3),
                            "recall_at_10": round(rec_, 4), "n_dist_per_query": round(c_nd / nq_c, 1),
                            "n_expand_per_query": round(c_ne / nq_c, 1),
                            "hbm_gbs": round(cbytes / (ck_ms * 1e-3) / 1e9, 1) if ck_ms > 0 else 0.0,
                            "hbm_frac": round(cbytes / (ck_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ck_ms > 0 else 0.0})
            return pts

        if rank == 0:
            hnsw["ef_curve"] = gpu_ef_curve(ix, queries, gt, HQ, RQ)
        # graph-path latency: VectorIndex::search is a one-query call (trait_impl.rs:38-42); 1 / 8 / 64 queries per call
        if rank == 0 and not a.no_latency_legs:
            lat_h = []
            for nq_l in (1, 8, 64):
                for _ in range(3):
                    ix.search_batch_dev(queries.data_ptr(), nq_l, K, a.ef, va.MODE_HNSW, h_ids.data_ptr(), h_sc.data_ptr(),
                                        h_n.data_ptr(), stream)
                torch.cuda.synchronize()
                samples = []
                for r_ in range(50):
                    qoff = (r_ * nq_l) % (n_query_pool - nq_l + 1)
                    tl0 = time.perf_counter()
                    ix.search_batch_dev(queries[qoff:qoff + nq_l].data_ptr(), nq_l, K, a.ef, va.MODE_HNSW, h_ids.data_ptr(),
                                        h_sc.data_ptr(), h_n.data_ptr(), stream)
                    torch.cuda.synchronize()
                    samples.append(time.perf_counter() - tl0)
                med = float(np.median(samples))
                l_nd, l_ne = ix.last_search_stats()  # of the last call
                lat_h.append({"queries_per_call": nq_l, "median_us_per_call": round(med * 1e6, 1),
                              "qps": round(nq_l / med, 1),
                              "neighbour_list_prediction_hit_rate": round(ix.last_prefetch_hits() / max(l_ne, 1), 3),
                              "alg_bytes_per_query": int((l_nd * D * 4 + l_ne * 2 * a.M * 4) / nq_l)})
            hnsw["latency_mode"] = lat_h
            # many host threads, one query per vdb_hip_index_search call (host pointers): the combining front (search_front.hip)
            cq_np = queries[:4096].cpu().numpy()
            c_ref = ix._search_raw(cq_np, K, a.ef, va.MODE_HNSW)   # ONE batched call (launches alone)
            hnsw["concurrent_callers"] = {
                "entry_point": "vdb_hip_index_search (host pointers, one query per call, ef %d) from T native threads" % a.ef,
                "points": callers_leg(ix, cq_np, K, a.ef, va.MODE_HNSW, c_ref),
                "front": {"max_batch": ix.get_option(va.OPT_COMBINE_MAX_BATCH), "inflight": ix.get_option(va.OPT_COMBINE_INFLIGHT),
                          "window_us": ix.get_option(va.OPT_COMBINE_WINDOW_US)}}
            ix.set_option(va.OPT_COMBINE_MAX_BATCH, 0)
            hnsw["concurrent_callers"]["without_the_front"] = callers_leg(ix, cq_np, K, a.ef, va.MODE_HNSW, c_ref, threads_list=(1, 16), seconds=0.8)
            ix.set_option(va.OPT_COMBINE_MAX_BATCH, -1)
            # the batch entry point from HOST memory (PCIe-inclusive; what the shim's search_batch_parallel binds)
            he = []
            for nq_h in (1024, 64):
                ix._search_raw(cq_np[:nq_h], K, a.ef, va.MODE_HNSW)
                th_ = time.perf_counter()
                for r_ in range(10):
                    ix._search_raw(cq_np[(r_ * nq_h) % (4096 - nq_h + 1):][:nq_h], K, a.ef, va.MODE_HNSW)
                hd_ = (time.perf_counter() - th_) / 10
                he.append({"queries_per_call": nq_h, "ms_per_call": round(hd_ * 1e3, 3), "qps": round(nq_h / hd_, 1)})
            hnsw["host_entry"] = he
            hstep()
            torch.cuda.synchronize()
        # dual-precision leg: int8 graph walk (integer L2^2 between u8 codes) + exact f32 re-rank of
        # the k * 4 best, same graph, same queries, same ef
        if not a.no_int8:
            ix.train_quantizer(0)
            torch.cuda.synchronize()

            def istep():
                ix.search_batch_dev(queries.data_ptr(), HQ, K, a.ef, va.MODE_HNSW_INT8, h_ids.data_ptr(), h_sc.data_ptr(),
                                    h_n.data_ptr(), stream)

            istep()
            barrier()
            ti = time.perf_counter()
            for i_ in range(a.hnsw_steps):
                events_on_last(i_, a.hnsw_steps)
                istep()
            barrier()
            idt = max_over_ranks(time.perf_counter() - ti)
            ik_ms, _ = ix.last_kernel_ms()
            va.set_kernel_timing(False)
            i_nd, i_ne = ix.last_search_stats()
            ibytes = i_nd * (D + 4) + i_ne * 2 * a.M * 4 + HQ * K * 4 * D * 4
            ii = h_ids[:RQ].cpu().numpy()
            rec_i = float(np.mean([len(set(ii[i].tolist()) & set(gt[i].tolist())) / K for i in range(RQ)]))
            hnsw["int8"] = {"qps": round(world * HQ * a.hnsw_steps / idt, 1), "ms_per_step": round(idt / a.hnsw_steps * 1e3, 3),
                            "recall_at_10": round(rec_i, 4), "n_dist_per_query": round(i_nd / HQ, 1),
                            "n_expand_per_query": round(i_ne / HQ, 1), "kernel_ms": round(ik_ms, 4),
                            "alg_bytes_per_launch": ibytes,
                            "alg_bytes_rule": "n_dist*(dim+4) + n_expand*M0*4 + k*oversampling(4)*dim*4 per query",
                            "hbm_gbs": round(ibytes / (ik_ms * 1e-3) / 1e9, 1) if ik_ms > 0 else 0.0,
                            "hbm_frac": round(ibytes / (ik_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ik_ms > 0 else 0.0}
            hstep()  # leave the f32 results in h_ids for the parity check below
            torch.cuda.synchronize()
        if rank == 0 and (not a.no_cpu_baseline or (world == 1 and not a.no_traffic_pass)):
            graph_dir = tempfile.mkdtemp(prefix="vdb_bench_")
            ix.save(graph_dir, "native_hnsw")
        if rank == 0 and world == 1 and not a.no_traffic_pass and graph_dir:
            tb_, _, tsrc_ = traffic_pass("hnsw", ("hnsw_search_kernel",), ["--graph-dir", graph_dir, "--hnsw-batch", str(HQ), "--ef", str(a.ef),
                                                                        "--M", str(a.M), "--efc", str(a.efc)], child_steps=2, child_warm=1)
            hnsw["roofline"]["traffic"] = tb_
            hnsw["roofline"]["traffic_source"] = tsrc_
            if tb_ is not None:
                hnsw["roofline"]["traffic_over_algorithmic"] = round(tb_ / hbytes, 3)

    # ---- exactness / recall check against the oracle on the full corpus (rank 0) ----
    recall = None
    check = {}
    cpu = None
    if rank == 0:
        from oracle import pyoracle as po
        om = {"cosine": po.COSINE, "euclidean": po.EUCLIDEAN, "dot": po.DOT}[a.metric]
        ncores = po.host_threads()
        if host_full is not None:
            # the headline launch itself (Q queries in ONE call = the GEMM kernel, every query tile), of which
            # `--check-queries` spread over the whole batch are compared with the oracle (ids + score bits)
            nchk = min(a.check_queries, Q)
            qall = queries[:Q].cpu().numpy()
            gi_all, gs_all, _ = ix.search_batch_brute_force(qall, K)
            sel = np.unique(np.linspace(0, Q - 1, nchk).astype(np.int64))
            qh, gi, gs = qall[sel], gi_all[sel], gs_all[sel]
            ci, cs = po.scan_topk(om, host_full, qh, K, po.MODE_M if ix.sweep_arith_mode(K) == "M" else po.MODE_C,
                                  nthreads=ncores)
            ri, rs = po.scan_topk(om, host_full, qh, K, po.MODE_R, nthreads=ncores)
            check = {"queries": int(sel.size), "of_a_batch_of": Q, "kernel": roofline["kernel"],
                     "ids_equal_oracle_canonical": bool(np.array_equal(gi, ci)),
                     "scores_bit_equal_oracle_canonical": bool(np.array_equal(gs.view(np.uint32), cs.view(np.uint32))),
                     "ids_equal_reference_order": bool(np.array_equal(gi, ri)),
                     "max_rel_diff_vs_reference_order": float(np.max(np.abs(gs - rs) / np.abs(rs)))}
            recall = float(np.mean([len(set(gi[i].tolist()) & set(ri[i].tolist())) / K for i in range(sel.size)]))
            if a.no_sq8_leg or world != 1:
                host_full = None  # (the SQ8 leg checks its batch against the oracle over the same rows)
        if not a.no_cpu_baseline:
            # bounded sample of the SAME workload on every host core through a persistent thread pool (rayon in the
            # reference), the corpus pages placed by the threads that scan them (NUMA); shape A = the production engine
            # (wide16: 4 x f32x8 FMA accumulators, oracle mode R), shape B = the true AVX-512F kernel of simd_native.rs
            # (one zmm accumulator, MODE_NATIVE); single-thread latency as criterion measures it (one query at a time)
            cpu_model, cpu_sockets, cpu_cores_per_socket = "", 1, ncores
            try:
                phys, cores_ = set(), 0
                with open("/proc/cpuinfo") as f:
                    for ln in f:
                        if ln.startswith("model name") and not cpu_model:
                            cpu_model = ln.split(":", 1)[1].strip()
                        elif ln.startswith("physical id"):
                            phys.add(ln.split(":", 1)[1].strip())
                        elif ln.startswith("cpu cores") and not cores_:
                            cores_ = int(ln.split(":", 1)[1])
                cpu_sockets = max(1, len(phys))
                cpu_cores_per_socket = cores_ or max(1, (os.cpu_count() or ncores) // cpu_sockets)
            except (OSError, ValueError):
                pass
            spread = po.SpreadRows(host_sample, ncores)
            hs = spread.array
            cal = min(max(8, ncores // 8), n_query_pool)
            qh = queries[:cal].cpu().numpy()
            po.scan_topk(om, hs, qh[:2], K, po.MODE_R, nthreads=ncores)  # pool start-up
            t3 = time.perf_counter()
            po.scan_topk(om, hs, qh, K, po.MODE_R, nthreads=ncores)
            cal_dt = time.perf_counter() - t3
            shapes = {}
            for shape, mode_c in (("shape_a", po.MODE_R), ("shape_b", po.MODE_NATIVE)):
                sq = int(max(cal, min(n_query_pool, 0.5 * a.cpu_seconds / max(cal_dt / cal, 1e-6))))
                qh = queries[:sq].cpu().numpy()
                t3 = time.perf_counter()
                po.scan_topk(om, hs, qh, K, mode_c, nthreads=ncores)
                cdt = time.perf_counter() - t3
                shapes[shape] = {"qps": round(sq / cdt * sample_rows / N, 3), "queries": sq, "seconds": round(cdt, 2)}
            shapes["shape_b"]["avx512f"] = bool(po.cpu_has_avx512f())
            st_samples = []
            for i in range(3):
                t3 = time.perf_counter()
                po.scan_topk(om, hs, queries[i:i + 1].cpu().numpy(), K, po.MODE_R, nthreads=1)
                st_samples.append(time.perf_counter() - t3)
            cpu = {"value": shapes["shape_a"]["qps"], "unit": "queries/s", "cores": ncores, "kind": "port",
                   "cpu_model": cpu_model, "host_hardware_threads": os.cpu_count(),
                   "cores_of_socket": f"{ncores} CPUs (cgroup quota) of a host with {os.cpu_count()} hardware threads ({cpu_sockets} x {cpu_model}): one whole "
                                      f"{cpu_cores_per_socket}-core socket is at best x{max(1, cpu_cores_per_socket // max(ncores, 1))} of `value`, the whole host x{max(1, cpu_sockets * cpu_cores_per_socket // max(ncores, 1))}",
                   "cores_note": "cores = CPUs this process may use (affinity mask capped by the cgroup cpu.max quota); "
                                 "one pool thread per CPU", "shape_a": shapes["shape_a"], "shape_b": shapes["shape_b"],
                   "single_thread_us": round(float(np.median(st_samples)) * 1e6 * N / sample_rows, 1),
                   "sample": f"oracle restatement of brute_force_search_parallel (batch.rs:223-244) on the first {sample_rows} of "
                             f"{N} rows, {ncores} pool threads, pages first-touched by the scanning threads; value = shape A "
                             f"(mode R, wide16 4 x f32x8 FMA, the production engine) x {sample_rows}/{N}; shape B = "
                             f"simd_native.rs AVX-512F kernel; single_thread_us = median of 3 one-query scans on one thread"}
            del spread, hs
            if graph_dir is not None:
                # the reference's graph search on the host cores, over the SAME graph (loaded from the files the
                # GPU index wrote in the reference's format), mode R arithmetic, reference tie order
                tl = time.perf_counter()
                og = po.NativeHnsw.file_load(graph_dir, "native_hnsw", om, po.MODE_R)
                og.spread(ncores)  # vectors re-placed round-robin over the pool's threads (NUMA)
                load_s = time.perf_counter() - tl
                cq = min(a.cpu_hnsw_queries, n_query_pool)
                qh = queries[:cq].cpu().numpy()
                og.search_batch(qh[:ncores], K, a.ef, po.TIE_REFERENCE, nthreads=ncores)  # pool start-up, page warm-up
                t4 = time.perf_counter()
                oi, od, oc, ond, one = og.search_batch(qh, K, a.ef, po.TIE_REFERENCE, nthreads=ncores)
                hdt_cpu = time.perf_counter() - t4
                # single-thread latency as criterion measures it (benches/hnsw_benchmark.rs:139-163: one search per
                # iteration, >= 100 samples, median)
                st = []
                for i in range(120):
                    t5 = time.perf_counter()
                    og.search(qh[i % cq], K, a.ef, po.TIE_REFERENCE)
                    st.append(time.perf_counter() - t5)
                st_med = float(np.median(st[20:]))
                # the CPU side of the ef curve (fewer queries per point: the whole curve stays within ~10 s)
                cqc = min(cq, 4096)
                for pt in hnsw.get("ef_curve", []):
                    t6 = time.perf_counter()
                    ci_, _, _, cnd_, _ = og.search_batch(qh[:cqc], K, pt["ef"], po.TIE_REFERENCE, nthreads=ncores)
                    cdt_ = time.perf_counter() - t6
                    rqc = min(RQ, cqc)
                    pt["cpu_qps"] = round(cqc / cdt_, 1)
                    pt["cpu_recall_at_10"] = round(float(np.mean([len(set(ci_[i].tolist()) & set(gt[i].tolist())) / K
                                                                 for i in range(rqc)])), 4)
                    pt["gpu_over_cpu"] = round(pt["qps"] / max(cqc / cdt_, 1e-9), 1)
                # parity of the GPU traversal with the canonical oracle on a few queries of the same graph
                og_c = po.NativeHnsw.file_load(graph_dir, "native_hnsw", om, po.MODE_C)
                pq = min(8, cq)
                ci, cd, cc, _, _ = og_c.search_batch(qh[:pq], K, a.ef, po.TIE_CANONICAL, nthreads=min(pq, ncores))
                gi = h_ids[:pq].cpu().numpy().astype(np.uint64)
                gsim = h_sc[:pq].cpu().numpy()
                exp_sim = np.array([[po.transform_score(om, float(x)) for x in row] for row in cd], dtype=np.float32)
                rq_n = min(RQ, cq)
                rec_cpu = float(np.mean([len(set(oi[i].tolist()) & set(gt[i].tolist())) / K for i in range(rq_n)]))
                hnsw["cpu_baseline"] = {
                    "value": round(cq / hdt_cpu, 1), "unit": "queries/s", "cores": ncores, "kind": "port",
                    "recall_at_10": round(rec_cpu, 4),
                    "single_thread_us": round(st_med * 1e6, 1), "per_thread_ms_per_query": round(hdt_cpu / cq * ncores * 1e3, 2),
                    "sample": f"oracle NativeHnsw::search (mode R, reference heap/tie order, reference's neighbour "
                              f"prefetch) over the same {N}-node graph, {cq} queries, ef={a.ef}, {ncores} pool threads, vectors "
                              f"placed round-robin over the threads, {hdt_cpu:.2f} s (+{load_s:.1f} s loading the graph files); "
                              f"single_thread_us = median of 100 one-query searches on one thread (criterion style); the "
                              f"reference publishes 4.8 ms at 1M x 768 (docs/guides/SEARCH_MODES.md:458-463)",
                    "n_dist_per_query": round(ond / cq, 1)}
                hnsw["parity_check"] = {"queries": pq, "ids_equal_oracle_canonical": bool(np.array_equal(gi, ci)),
                                        "scores_bit_equal_oracle_canonical":
                                            bool(np.array_equal(gsim.view(np.uint32), exp_sim.view(np.uint32)))}
                hnsw["gpu_over_cpu"] = round(hnsw["qps"] / (cq / hdt_cpu), 2)
                if "concurrent_callers" in hnsw and isinstance(hnsw["concurrent_callers"].get("points"), list):
                    # the oracle's search on the same number of host threads over the same graph (capped by the box's cores)
                    for pt in hnsw["concurrent_callers"]["points"]:
                        tc = min(pt["threads"], ncores)
                        nqc_ = min(cq, 512 * tc)
                        tcp = time.perf_counter()
                        og.search_batch(qh[:nqc_], K, a.ef, po.TIE_REFERENCE, nthreads=tc)
                        pt["cpu_threads"] = tc
                        pt["cpu_qps"] = round(nqc_ / (time.perf_counter() - tcp), 1)
                        pt["gpu_over_cpu"] = round(pt["qps"] / max(pt["cpu_qps"], 1e-9), 2)
                del og, og_c
        if graph_dir is not None:
            shutil.rmtree(graph_dir, ignore_errors=True)

    # ---- SQ8 storage mode (N = 1 only): exact top-k of f32 queries over the one-byte codes with the reference's
    # asymmetric distances (quantization.rs:410-554).  Large batches select on the bf16 matrix cores over the dequantised
    # rows and re-score the candidates with the reference's chain (bit-exact); small batches sweep the codes.
    sq8_leg = None
    if world == 1 and not a.no_sq8_leg and a.metric in ("cosine", "dot"):
        ix.set_storage_mode(va.StorageMode.SQ8)

        def sq8_run(nq, reps, warm=2):
            for _ in range(warm):
                ix.search_batch_dev(queries[:nq].data_ptr(), nq, K, 0, va.MODE_BRUTE_SQ8, out_ids.data_ptr(), out_sc.data_ptr(),
                                    out_n.data_ptr(), stream)
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            for _ in range(reps):
                ix.search_batch_dev(queries[:nq].data_ptr(), nq, K, 0, va.MODE_BRUTE_SQ8, out_ids.data_ptr(), out_sc.data_ptr(),
                                    out_n.data_ptr(), stream)
            torch.cuda.synchronize()
            return (time.perf_counter() - t_s) / reps

        nq_big = min(Q, 1024)
        dt_sel = sq8_run(nq_big, 10, warm=20)  # (20 untimed calls first: the chip settles into the matrix-core mix, see `settle_steps`)
        lvl = ix.last_select_level() if nq_big >= 224 else 0
        nq_l, unp = ix.last_split_stats()
        sel_ids = out_ids[:nq_big].cpu().numpy().astype(np.uint64)
        sel_sc = out_sc[:nq_big].cpu().numpy().copy()
        va.set_split_selector(0)
        dt_exact = sq8_run(min(nq_big, 64), 2) / min(nq_big, 64) * nq_big  # the exact sweep serves 8 queries per pass: linear in the batch
        ex_ids = out_ids[:min(nq_big, 64)].cpu().numpy().astype(np.uint64)
        ex_sc = out_sc[:min(nq_big, 64)].cpu().numpy().copy()
        dt_8 = sq8_run(8, 5)
        va.set_split_selector(0 if a.no_split else a.select_level)
        dt_8d = sq8_run(8, 5)  # the default path: from 6 queries up the selection stage (one partly filled query tile) is ahead of the exact sweep
        lvl_8d = ix.last_select_level()
        # one query per call — what a single search() on a StorageMode::SQ8 collection gets (sweep_topk_sq8<B=1>: the codes pass once,
        # 5 vector instructions per code: this one IS bandwidth-bound); kernel time from HIP events on the last call
        va.set_kernel_timing(True)
        dt_1 = sq8_run(1, 10)
        k1_ms, _ = ix.last_kernel_ms()
        va.set_kernel_timing(False)
        code_bytes = N * (D + 12 + (4 if a.metric == "cosine" else 0))
        # the exact sweep at B queries per pass is bound by the vector ALUs, not by HBM: per code 3 instructions of dequantisation +
        # (multiply, add) per query, separately rounded and in the reference's left-to-right order (quantization.rs:452-466) —
        # N * D * (3 + 2 B) lane operations against 78.6 T lane-op/s (1 024 SIMDs x 32 lanes x 2.4 GHz; half the fma-counted 157.3 TF)
        valu_ops_8 = N * D * (3 + 2 * 8)
        sq8_leg = {"workload": f"{N}x{D} SQ8 codes ({a.metric}, asymmetric f32-query distances), k={K}",
                   "batch": {"queries": nq_big, "qps": round(nq_big / dt_sel, 1), "ms_per_batch": round(dt_sel * 1e3, 3),
                             "select_level": lvl, "unproven_queries_last_batch": unp,
                             "kernel": ("sweep_topk_gemm_bf16_pp<WIDE> over the dequantised bf16 image + wide_rerank_sq8 + " if lvl == 4 else
                                        "sweep_topk_gemm_bf16_pp over the dequantised bf16 image + split_rerank_verify<SQ8> + ") +
                                       "gathered sweep_topk_sq8 for unproven queries"},
                   "exact_sweep_same_batch": {"qps": round(nq_big / dt_exact, 1), "ms_per_batch": round(dt_exact * 1e3, 3),
                                              "note": "sweep_topk_sq8<B=8> for every query (selection off), extrapolated from 64 queries"},
                   "batch_equals_exact_sweep_bitwise": bool(np.array_equal(sel_ids[:len(ex_ids)], ex_ids) and
                                                            np.array_equal(sel_sc[:len(ex_sc)].view(np.uint32), ex_sc.view(np.uint32))),
                   "one_query": {"qps": round(1 / dt_1, 1), "ms_per_call": round(dt_1 * 1e3, 4), "sweep_kernel_ms": round(k1_ms, 4),
                                 "hbm_gbs": round(code_bytes / (k1_ms * 1e-3) / 1e9, 1) if k1_ms > 0 else 0.0,
                                 "hbm_frac": round(code_bytes / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k1_ms > 0 else 0.0},
                   "eight_queries": {"qps": round(8 / dt_8, 1), "ms_per_call": round(dt_8 * 1e3, 4),
                                     "hbm_gbs": round(code_bytes / dt_8 / 1e9, 1), "hbm_frac": round(code_bytes / dt_8 / 1e9 / HBM_PEAK_GBS, 4),
                                     "valu_frac": round(valu_ops_8 / dt_8 / 78.6e12, 4),
                                     "bound": "vector ALU (3 + 2 B separately rounded lane operations per code in the reference's order), not HBM: "
                                              "at B = 8 the pass is 14.6 G lane operations = 0.19 ms at the vector peak, 0.10 ms at HBM peak"},
                   "eight_queries_default_path": {"qps": round(8 / dt_8d, 1), "ms_per_call": round(dt_8d * 1e3, 4), "select_level": lvl_8d},
                   "alg_bytes_per_pass": code_bytes}
        if rank == 0 and host_full is not None and a.check_queries > 0:
            from oracle import pyoracle as po_s
            nchk = min(16, nq_big)
            pick = np.unique(np.linspace(0, nq_big - 1, nchk).astype(np.int64))
            eid_s, esc_s = po_s.scan_topk_sq8({"cosine": po_s.COSINE, "dot": po_s.DOT}[a.metric], host_full, queries[:nq_big].cpu().numpy()[pick],
                                              K, nthreads=po_s.host_threads())
            sq8_leg["parity_check"] = {"queries": int(len(pick)), "of_a_batch_of": nq_big,
                                       "ids_equal_oracle": bool(np.array_equal(sel_ids[pick], eid_s.astype(np.uint64))),
                                       "scores_bit_equal_oracle": bool(np.array_equal(sel_sc[pick].view(np.uint32), esc_s.view(np.uint32)))}
        ix.set_storage_mode(va.StorageMode.Full)
    host_full = None

    # ---- graph leg on embedding-like data (N = 1 only): iid N(0,1) in 768-D has no neighbourhood structure, so HNSW
    # recall there is ~0.04 for the reference and the GPU alike.  Real embeddings have low intrinsic
    # dimension: `--latent` Gaussian factors through a random projection + noise.  Same build, same traversal kernel,
    # same CPU baseline (the oracle's restatement of NativeHnsw::search over the very same graph) — this is the
    # "QPS at recall@10" the metric asks for.
    hnsw_emb = None
    if world == 1 and not a.no_hnsw and not a.no_embedding_leg:
        from oracle import pyoracle as po
        om = {"cosine": po.COSINE, "euclidean": po.EUCLIDEAN, "dot": po.DOT}[a.metric]
        g.manual_seed(44)
        proj = torch.randn((a.latent, D), generator=g, device=dev)
        rows2 = torch.randn((N, a.latent), generator=g, device=dev) @ proj
        rows2 += a.latent_noise * torch.randn((N, D), generator=g, device=dev)
        HQ = min(a.hnsw_batch, n_query_pool)
        q2 = torch.randn((HQ, a.latent), generator=g, device=dev) @ proj
        q2 += a.latent_noise * torch.randn((HQ, D), generator=g, device=dev)
        ix2 = va.HnswIndex(D, metric, va.HnswParams(a.M, a.efc, N), device=local)
        torch.cuda.synchronize()
        ix2.upload_dev(0, rows2.data_ptr(), N, stream)
        torch.cuda.synchronize()
        del rows2
        torch.cuda.empty_cache()
        tb = time.perf_counter()
        ix2.build_graph(0)
        torch.cuda.synchronize()
        build2 = time.perf_counter() - tb
        e_ids = torch.empty((HQ, K), dtype=torch.int64, device=dev)
        e_sc = torch.empty((HQ, K), dtype=torch.float32, device=dev)
        e_n = torch.empty((HQ,), dtype=torch.int32, device=dev)

        def estep():
            ix2.search_batch_dev(q2.data_ptr(), HQ, K, a.ef, va.MODE_HNSW, e_ids.data_ptr(), e_sc.data_ptr(),
                                 e_n.data_ptr(), stream)

        estep()
        torch.cuda.synchronize()
        te = time.perf_counter()
        for i_ in range(a.hnsw_steps):
            events_on_last(i_, a.hnsw_steps)
            estep()
        torch.cuda.synchronize()
        edt = time.perf_counter() - te
        ek_ms, _ = ix2.last_kernel_ms()
        va.set_kernel_timing(False)
        e_nd, e_ne = ix2.last_search_stats()
        ebytes = e_nd * D * 4 + e_ne * 2 * a.M * 4
        RQ = min(a.recall_queries, HQ)
        gt2, _, _ = ix2.search_batch_brute_force(q2[:RQ].cpu().numpy(), K)
        ei = e_ids[:RQ].cpu().numpy()
        rec2 = float(np.mean([len(set(ei[i].tolist()) & set(gt2[i].tolist())) / K for i in range(RQ)]))
        hnsw_emb = {"workload": f"{N}x{D} f32 {a.metric}, {a.latent} Gaussian latent factors x random projection + "
                                f"{a.latent_noise} noise (embedding-like), M={a.M}, ef_construction={a.efc}, built on the "
                                f"GPU, k={K}, ef={a.ef}, {HQ} queries/step",
                    "qps": round(HQ * a.hnsw_steps / edt, 1), "ms_per_step": round(edt / a.hnsw_steps * 1e3, 3),
                    "recall_at_10": round(rec2, 4), "recall_queries": RQ, "build_seconds": round(build2, 2),
                    "n_dist_per_query": round(e_nd / HQ, 1), "n_expand_per_query": round(e_ne / HQ, 1),
                    "roofline": {"bound": "hbm", "achieved": round(ebytes / (ek_ms * 1e-3) / 1e9, 1) if ek_ms > 0 else 0.0,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(ebytes / (ek_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ek_ms > 0 else 0.0,
                                 "traffic": None, "kernel_ms": round(ek_ms, 4), "alg_bytes_per_launch": ebytes}}
        hnsw_emb["ef_curve"] = gpu_ef_curve(ix2, q2, gt2, HQ, RQ)
        if not a.no_traffic_pass:  # FETCH_SIZE of the same launch over the same graph, in a child pass of this run
            gdt_ = tempfile.mkdtemp(prefix="vdb_bench_embt_")
            try:
                ix2.save(gdt_, "native_hnsw")
                tb_, _, tsrc_ = traffic_pass("hnsw", ("hnsw_search_kernel",), ["--graph-dir", gdt_, "--hnsw-batch", str(HQ), "--ef", str(a.ef),
                                                                            "--M", str(a.M), "--efc", str(a.efc)], child_steps=2, child_warm=1)
                hnsw_emb["roofline"]["traffic"] = tb_
                hnsw_emb["roofline"]["traffic_source"] = tsrc_
                if tb_ is not None:
                    hnsw_emb["roofline"]["traffic_over_algorithmic"] = round(tb_ / ebytes, 3)
            finally:
                shutil.rmtree(gdt_, ignore_errors=True)
        if not a.no_cpu_baseline:
            gd = tempfile.mkdtemp(prefix="vdb_bench_emb_")
            try:
                ix2.save(gd, "native_hnsw")
                og = po.NativeHnsw.file_load(gd, "native_hnsw", om, po.MODE_R)
                ncores = po.host_threads()
                og.spread(ncores)
                cq = min(a.cpu_hnsw_queries, HQ)
                qh2 = q2[:cq].cpu().numpy()
                og.search_batch(qh2[:ncores], K, a.ef, po.TIE_REFERENCE, nthreads=ncores)  # pool / page warm-up
                cqc = min(cq, 4096)
                for pt in hnsw_emb["ef_curve"]:
                    t6 = time.perf_counter()
                    ci_, _, _, _, _ = og.search_batch(qh2[:cqc], K, pt["ef"], po.TIE_REFERENCE, nthreads=ncores)
                    cdt_ = time.perf_counter() - t6
                    rqc = min(RQ, cqc)
                    pt["cpu_qps"] = round(cqc / cdt_, 1)
                    pt["cpu_recall_at_10"] = round(float(np.mean([len(set(ci_[i].tolist()) & set(gt2[i].tolist())) / K
                                                                 for i in range(rqc)])), 4)
                    pt["gpu_over_cpu"] = round(pt["qps"] / max(cqc / cdt_, 1e-9), 1)
                t5 = time.perf_counter()
                oi, od, oc, ond, one = og.search_batch(qh2, K, a.ef, po.TIE_REFERENCE, nthreads=ncores)
                cdt2 = time.perf_counter() - t5
                rqn = min(RQ, cq)
                rec_c = float(np.mean([len(set(oi[i].tolist()) & set(gt2[i].tolist())) / K for i in range(rqn)]))
                hnsw_emb["cpu_baseline"] = {"value": round(cq / cdt2, 1), "unit": "queries/s", "cores": ncores, "kind": "port",
                                            "recall_at_10": round(rec_c, 4),
                                            "sample": f"oracle NativeHnsw::search (mode R, reference tie order) over the same "
                                                      f"graph, {cq} queries, ef={a.ef}, {ncores} threads, {cdt2:.2f} s"}
                hnsw_emb["gpu_over_cpu"] = round(hnsw_emb["qps"] / (cq / cdt2), 2)
                del og
            finally:
                shutil.rmtree(gd, ignore_errors=True)
        ix2.close()

    # ---- configs[2] as the REFERENCE would run it (N = 1 only): HnswParams::for_dataset_size / million_scale(768) (params.rs:72-157) =
    # max_connections 128 (M0 = 256), ef_construction 1600 — the preset the reference picks for a 768-D corpus beyond 10 000 vectors.
    # Same iid N(0,1) corpus and queries as the headline, same traversal kernel (throughput instance; lists of 256 neighbours), the
    # construction roofline at ef_construction 1600, small calls (latency mode, test-first form) and the CPU restatement over the
    # very same graph.  tests/test_gpu_m128.py holds the same path to the oracle (build link for link, traversal ids + bits + counters).
    hnsw_m128 = None
    if world == 1 and rank == 0 and not a.no_m128_leg:
        from oracle import pyoracle as po
        om = {"cosine": po.COSINE, "euclidean": po.EUCLIDEAN, "dot": po.DOT}[a.metric]
        pm = va.HnswParams.million_scale(D)
        MN = a.m128_rows
        M1, EFC1 = pm.max_connections, pm.ef_construction
        g.manual_seed(42)
        rows3 = torch.randn((MN, D), generator=g, device=dev, dtype=torch.float32)  # (the headline corpus again when MN == N)
        ix4 = va.HnswIndex(D, metric, va.HnswParams(M1, EFC1, MN), device=local)
        torch.cuda.synchronize()
        ix4.upload_dev(0, rows3.data_ptr(), MN, stream)
        torch.cuda.synchronize()
        del rows3
        torch.cuda.empty_cache()
        tb = time.perf_counter()
        ix4.build_graph(0)
        torch.cuda.synchronize()
        build4 = time.perf_counter() - tb
        b_rows, b_phases, b_nodes, b_sel = ix4.build_stats()
        bb4 = (b_rows - b_sel) * D * 4
        HQ4 = min(a.hnsw_batch, n_query_pool)
        m_ids = torch.empty((HQ4, K), dtype=torch.int64, device=dev)
        m_sc = torch.empty((HQ4, K), dtype=torch.float32, device=dev)
        m_n = torch.empty((HQ4,), dtype=torch.int32, device=dev)

        def mstep(nq_=HQ4, off_=0):
            ix4.search_batch_dev(queries[off_:off_ + nq_].data_ptr(), nq_, K, a.ef, va.MODE_HNSW, m_ids.data_ptr(), m_sc.data_ptr(), m_n.data_ptr(), stream)

        mstep()
        torch.cuda.synchronize()
        tm = time.perf_counter()
        for i_ in range(a.hnsw_steps):
            events_on_last(i_, a.hnsw_steps)
            mstep()
        torch.cuda.synchronize()
        mdt = time.perf_counter() - tm
        mk_ms, _ = ix4.last_kernel_ms()
        va.set_kernel_timing(False)
        m_nd, m_ne = ix4.last_search_stats()
        mbytes = m_nd * D * 4 + m_ne * 2 * M1 * 4
        RQ4 = min(a.recall_queries, HQ4)
        gt4, _, _ = ix4.search_batch_brute_force(queries[:RQ4].cpu().numpy(), K)
        mi = m_ids[:RQ4].cpu().numpy()
        rec4 = float(np.mean([len(set(mi[i].tolist()) & set(gt4[i].tolist())) / K for i in range(RQ4)]))
        hnsw_m128 = {"workload": f"{MN}x{D} f32 {a.metric} iid N(0,1), HnswParams::million_scale({D}) = M {M1} (M0 {2 * M1}), ef_construction {EFC1} "
                                 f"(params.rs:124-157), built on the GPU, k={K}, ef={a.ef}, {HQ4} queries/step",
                     "qps": round(HQ4 * a.hnsw_steps / mdt, 1), "ms_per_step": round(mdt / a.hnsw_steps * 1e3, 3), "recall_at_10": round(rec4, 4),
                     "recall_queries": RQ4, "build_seconds": round(build4, 2), "build_inserts_per_s": round(MN / build4, 1),
                     "build": {"roofline": {"bound": "hbm", "achieved": round(bb4 / build4 / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": round(bb4 / build4 / 1e9 / HBM_PEAK_GBS, 4),
                                            "frac_with_select_neighbors_rereads": round(b_rows * D * 4 / build4 / 1e9 / HBM_PEAK_GBS, 4),
                                            "rows_evaluated_per_insert": round(b_rows / max(b_nodes, 1), 1),
                                            "select_neighbors_rows_per_insert": round(b_sel / max(b_nodes, 1), 1),
                                            "distance_phases_per_insert": round(b_phases / max(b_nodes, 1), 1), "seconds": round(build4, 2)}},
                     "n_dist_per_query": round(m_nd / HQ4, 1), "n_expand_per_query": round(m_ne / HQ4, 1),
                     "roofline": {"bound": "hbm", "achieved": round(mbytes / (mk_ms * 1e-3) / 1e9, 1) if mk_ms > 0 else 0.0, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(mbytes / (mk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if mk_ms > 0 else 0.0, "traffic": None,
                                  "kernel_ms": round(mk_ms, 4), "alg_bytes_per_launch": mbytes,
                                  "alg_bytes_rule": "n_dist*dim*4 + n_expand*M0*4, counters from the kernel"}}
        # small calls: 1 / 8 / 64 queries per call (device-resident), the latency-mode walk in its test-first form
        lat4 = []
        for nq_l in (1, 8, 64):
            for _ in range(3):
                mstep(nq_l)
            torch.cuda.synchronize()
            smp = []
            for r_ in range(30):
                t_l = time.perf_counter()
                mstep(nq_l, (r_ * nq_l) % (n_query_pool - nq_l + 1))
                torch.cuda.synchronize()
                smp.append(time.perf_counter() - t_l)
            lat4.append({"queries_per_call": nq_l, "median_us_per_call": round(float(np.median(smp)) * 1e6, 1), "qps": round(nq_l / float(np.median(smp)), 1)})
        hnsw_m128["latency_mode"] = lat4
        if not a.no_cpu_baseline:
            gd4 = tempfile.mkdtemp(prefix="vdb_bench_m128_")
            try:
                ix4.save(gd4, "native_hnsw")
                og4 = po.NativeHnsw.file_load(gd4, "native_hnsw", om, po.MODE_R)
                nc4 = po.host_threads()
                og4.spread(nc4)
                cq4 = min(4096, n_query_pool)
                qh4 = queries[:cq4].cpu().numpy()
                og4.search_batch(qh4[:nc4], K, a.ef, po.TIE_REFERENCE, nthreads=nc4)  # pool / page warm-up
                t4_ = time.perf_counter()
                oi4, _, _, ond4, _ = og4.search_batch(qh4, K, a.ef, po.TIE_REFERENCE, nthreads=nc4)
                cdt4 = time.perf_counter() - t4_
                rq4 = min(RQ4, cq4)
                rec_c4 = float(np.mean([len(set(oi4[i].tolist()) & set(gt4[i].tolist())) / K for i in range(rq4)]))
                st4 = []
                for i in range(40):
                    t5_ = time.perf_counter()
                    og4.search(qh4[i], K, a.ef, po.TIE_REFERENCE)
                    st4.append(time.perf_counter() - t5_)
                del og4
                # parity: the canonical oracle over the same graph, a few queries of the timed batch (ids + score bits)
                og4c = po.NativeHnsw.file_load(gd4, "native_hnsw", om, po.MODE_C)
                pq4 = 8
                mstep()
                torch.cuda.synchronize()
                ci4, cd4, _, _, _ = og4c.search_batch(qh4[:pq4], K, a.ef, po.TIE_CANONICAL, nthreads=pq4)
                exp4 = np.array([[po.transform_score(om, float(x)) for x in row] for row in cd4], dtype=np.float32)
                hnsw_m128["parity_check"] = {"queries": pq4,
                                             "ids_equal_oracle_canonical": bool(np.array_equal(m_ids[:pq4].cpu().numpy().astype(np.uint64), ci4)),
                                             "scores_bit_equal_oracle_canonical": bool(np.array_equal(m_sc[:pq4].cpu().numpy().view(np.uint32), exp4.view(np.uint32)))}
                del og4c
                hnsw_m128["cpu_baseline"] = {"value": round(cq4 / cdt4, 1), "unit": "queries/s", "cores": nc4, "kind": "port", "recall_at_10": round(rec_c4, 4),
                                             "single_thread_us": round(float(np.median(st4[8:])) * 1e6, 1), "n_dist_per_query": round(ond4 / cq4, 1),
                                             "sample": f"oracle NativeHnsw::search (mode R, reference tie order) over the same {MN}-node M {M1} graph, {cq4} queries, "
                                                       f"ef={a.ef}, {nc4} pool threads, {cdt4:.2f} s"}
                hnsw_m128["gpu_over_cpu"] = round(hnsw_m128["qps"] / (cq4 / cdt4), 2)
            finally:
                shutil.rmtree(gd4, ignore_errors=True)
        ix4.close()
        torch.cuda.empty_cache()

    # ---- BASELINE configs[0] on the GPU (N = 1 only): the reference's criterion workload — 10 000 x 768 `generate_vector`
    # rows, cosine, M 32 / ef_construction 400 — `index.search(query, 10)` one query per call through the HOST entry point
    # (what VectorIndex::search is), beside the reference's published 56.8 us / 9.2 K q/s (bench_hnsw_results.txt:86-87,
    # 114-116, i9-14900KF).  The graph is built with the batched GPU construction (the sequential-insert graph of the
    # parity test takes a minute to build and searches the same way).
    config0 = None
    if world == 1 and rank == 0 and not a.no_hnsw and not a.no_latency_legs:
        def generate_vector(dim, seed):
            i = np.arange(dim, dtype=np.float32)
            return ((np.sin(np.float32(seed) * np.float32(0.1) + i * np.float32(0.01)) + np.float32(1.0)) / np.float32(2.0)).astype(np.float32)
        n0 = 10_000
        rows0 = np.stack([generate_vector(768, s_) for s_ in range(n0)])
        ix0 = va.HnswIndex(768, va.DistanceMetric.Cosine, va.HnswParams(32, 400, n0), device=local)
        tb0 = time.perf_counter()
        ix0.upload(np.arange(n0, dtype=np.uint64), rows0)
        ix0.build_graph(0)
        b0 = time.perf_counter() - tb0
        q0_ = generate_vector(768, 99_999)
        for _ in range(5):
            ix0.search(q0_, 10)
        lat0 = []
        for _ in range(200):
            t0_ = time.perf_counter()
            ix0.search(q0_, 10)
            lat0.append(time.perf_counter() - t0_)
        qs0 = np.stack([generate_vector(768, 100_000 + j) for j in range(100)])
        t0_ = time.perf_counter()
        for j in range(100):
            ix0.search(qs0[j], 10)
        seq0 = time.perf_counter() - t0_
        t0_ = time.perf_counter()
        ix0.search_batch_parallel(qs0, 10, va.SearchQuality.Balanced)
        bat0 = time.perf_counter() - t0_
        config0 = {"workload": "10000x768 cosine, generate_vector rows, M=32 ef_construction=400, index.search(query, 10) (ef 128)",
                   "build_seconds_batched": round(b0, 2),
                   "search_median_us": round(float(np.median(lat0)) * 1e6, 1), "search_p99_us": round(float(np.percentile(lat0, 99)) * 1e6, 1),
                   "sequential_100_queries_qps": round(100 / seq0, 1), "one_call_100_queries_qps": round(100 / bat0, 1),
                   "reference_published": {"search_us": 56.8, "qps": 9200, "host": "i9-14900KF, criterion (bench_hnsw_results.txt:86-87,114-116)"},
                   "note": "a single query cannot fill a GPU: the graph path pays kernel launch + synchronisation + a serial "
                           "walk per call; the GPU path overtakes one CPU thread from a few queries per call (one_call_100_queries_qps)"}
        qs0b = np.stack([generate_vector(768, 100_000 + j) for j in range(512)])
        ref0 = ix0._search_raw(qs0b, 10, 128, va.MODE_AUTO)
        config0["concurrent_callers"] = {"entry_point": "vdb_hip_index_search (host pointers, one query per call, Balanced) from T native threads",
                                         "points": callers_leg(ix0, qs0b, 10, 128, va.MODE_AUTO, ref0, seconds=0.8)}
        if not a.no_cpu_baseline:
            from oracle import pyoracle as po0
            gd0 = tempfile.mkdtemp(prefix="vdb_bench0_")
            try:
                ix0.save(gd0, "native_hnsw")
                og0 = po0.NativeHnsw.file_load(gd0, "native_hnsw", po0.COSINE, po0.MODE_R)
                nc0 = po0.host_threads()
                og0.search_batch(qs0b[:nc0], 10, 128, po0.TIE_REFERENCE, nthreads=nc0)
                for pt in config0["concurrent_callers"]["points"]:
                    tc = min(pt["threads"], nc0)
                    tcp = time.perf_counter()
                    for _ in range(4):
                        og0.search_batch(qs0b, 10, 128, po0.TIE_REFERENCE, nthreads=tc)
                    pt["cpu_threads"] = tc
                    pt["cpu_qps"] = round(4 * 512 / (time.perf_counter() - tcp), 1)
                    pt["gpu_over_cpu"] = round(pt["qps"] / max(pt["cpu_qps"], 1e-9), 2)
                del og0
            finally:
                shutil.rmtree(gd0, ignore_errors=True)
        ix0.close()

    # ---- bf16 GEMM distance (BASELINE configs[3], N = 1 only): 10 M x 768 bf16 rows, 1 024 queries per batch contracted
    # on the bf16 matrix cores with f32 accumulation (half_precision.rs:199-255 semantics), fused top-k.  Bound: MFMA.
    bf16_leg = None
    if world == 1 and not a.no_bf16_leg and a.metric in ("cosine", "dot") and D % 64 == 0:
        BR, BQ = a.bf16_rows, 1024
        ix3 = va.HnswIndex(D, metric, va.HnswParams(a.M, a.efc, BR), device=local)
        ix3.enable_bf16()
        g.manual_seed(45)
        first_chunk = None
        chunk = 1_000_000
        for base in range(0, BR, chunk):
            n_c = min(chunk, BR - base)
            c = torch.randn((n_c, D), generator=g, device=dev)
            torch.cuda.synchronize()
            ix3.upload_dev(base, c.data_ptr(), n_c, stream)
            if base == 0 and not a.no_cpu_baseline:
                first_chunk = c[:min(a.bf16_cpu_rows, n_c)].cpu().numpy()
            del c
        bq = torch.randn((BQ, D), generator=g, device=dev)
        b_ids = torch.empty((BQ, K), dtype=torch.int64, device=dev)
        b_sc = torch.empty((BQ, K), dtype=torch.float32, device=dev)
        b_n = torch.empty((BQ,), dtype=torch.int32, device=dev)

        def bstep(mode=va.MODE_BRUTE_BF16):
            ix3.search_batch_dev(bq.data_ptr(), BQ, K, 0, mode, b_ids.data_ptr(), b_sc.data_ptr(), b_n.data_ptr(), stream)

        bstep()
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        for i_ in range(a.bf16_steps):
            events_on_last(i_, a.bf16_steps)
            bstep()
        torch.cuda.synchronize()
        bdt = (time.perf_counter() - tb0) / a.bf16_steps
        bk_ms, b_nl = ix3.last_kernel_ms()
        va.set_kernel_timing(False)
        bf_ids = b_ids.cpu().numpy().copy()
        bstep(va.MODE_BRUTE)  # exact f32 sweep over the same rows: ground truth of the recall figure
        torch.cuda.synchronize()
        ex_ids = b_ids.cpu().numpy()
        rec_b = float(np.mean([len(set(bf_ids[i].tolist()) & set(ex_ids[i].tolist())) / K for i in range(BQ)]))
        bflop = 2.0 * BR * D * BQ
        b_tf = bflop / (bk_ms * 1e-3) / 1e12 if bk_ms > 0 else 0.0
        bf16_leg = {"workload": f"{BR}x{D} bf16 {a.metric} (rows and queries rounded to nearest even, f32 accumulate), "
                                f"{BQ} queries per batch as one MFMA GEMM distance with fused top-k, k={K} (BASELINE configs[3])",
                    "qps": round(BQ / bdt, 1), "ms_per_batch": round(bdt * 1e3, 3), "recall_at_10_vs_exact_f32": round(rec_b, 4),
                    "roofline": {"bound": "mfma", "achieved": round(b_tf, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                 "frac": round(b_tf / 2500.0, 4), "traffic": None, "kernel_ms": round(bk_ms, 4),
                                 "launches_timed": b_nl, "alg_flops_per_launch": bflop,
                                 "alg_bytes_per_launch": BR * D * 2 + BR * 4 + BQ * D * 2,
                                 "kernel": "sweep_topk_gemm_bf16_pp<%s> (256x256 LDS-DMA tile, ping-pong pipeline; timed region = seed sweep + launches of <= 2 M rows + merges)" % a.metric,
                                 "note": "dense bf16 MFMA peak 2.5 PFLOP/s (v_mfma_f32_16x16x32_bf16); algorithmic flop = 2*rows*dim*queries"}}
        if first_chunk is not None:
            from oracle import pyoracle as po
            om = {"cosine": po.COSINE, "dot": po.DOT}[a.metric]
            ncores = po.host_threads()
            qh = bq.cpu().numpy()
            nqc = min(ncores, BQ)
            t6 = time.perf_counter()
            oi, osc = po.scan_topk_bf16(om, first_chunk, qh[:nqc], K, nthreads=ncores)   # calibration pass + the parity check's oracle side
            t_c = time.perf_counter() - t6
            # parity of configs[3]'s kernel IN this run: the same 1 024-query batch over an index of the slice the oracle just
            # scanned (>= 65 536 rows: served by sweep_topk_gemm_bf16_glds, asserted), first nqc queries against the oracle by
            # the rule of tests/test_gpu_bf16.py (scores within 1e-5 of the f64 value — relative to |q||v| for the dot product —
            # and ids equal wherever neighbouring oracle scores are further apart than that)
            ixp = va.HnswIndex(D, metric, va.HnswParams(a.M, a.efc, first_chunk.shape[0]), device=local)
            ixp.upload(np.arange(first_chunk.shape[0]), first_chunk)
            ixp.enable_bf16()
            pi, ps, _ = ixp.search_batch_brute_force_bf16(qh, K)
            glds = bool(ixp.last_kernels() & va.KERNEL_GEMM_BF16_GLDS)
            ixp.close()
            qq64 = po.round_bf16(qh[:nqc]).astype(np.float64)
            ids_equal = bool(np.array_equal(pi[:nqc], oi))
            ok_q, worst = 0, 0.0
            for i in range(nqc):
                r64 = po.round_bf16(first_chunk[pi[i].astype(np.int64)]).astype(np.float64)
                tv = r64 @ qq64[i]
                sc_ = np.linalg.norm(r64, axis=1) * np.linalg.norm(qq64[i])
                if a.metric == "cosine":
                    tv, sc_ = tv / sc_, np.ones_like(sc_)
                err = float(np.max(np.abs(ps[i].astype(np.float64) - tv) / sc_))
                worst = max(worst, err)
                good = err <= 1e-5
                for r in range(K):
                    if pi[i, r] != oi[i, r]:  # a different row at this rank: only inside a near-tie of the oracle's scores
                        good &= abs(float(osc[i, r]) - tv[r]) <= 2e-5 * sc_[r]
                ok_q += int(good)
            bf16_leg["parity_check"] = {"rows": int(first_chunk.shape[0]), "queries": int(nqc), "served_by_gemm_bf16_glds": glds,
                                        "queries_within_rule": ok_q, "ids_identical_to_oracle": ids_equal,
                                        "max_score_err_rel": worst, "ok": bool(glds and ok_q == nqc),
                                        "rule": "scores within 1e-5 (cosine: absolute; dot: relative to |q||v|) of the f64 score of the "
                                                "bf16-rounded inputs; ids equal to oracle scan_topk_bf16 except inside near-ties <= 2e-5"}
            nq2, t_c2 = nqc, t_c
            if t_c < 0.5 * a.cpu_seconds:  # one query per thread was short: a second, longer pass sized to the target
                nq2 = int(max(nqc, min(BQ, nqc * (a.cpu_seconds / max(t_c, 1e-3)))))
                t7 = time.perf_counter()
                po.scan_topk_bf16(om, first_chunk, qh[:nq2], K, nthreads=ncores)
                t_c2 = time.perf_counter() - t7
            slice_qps = nq2 / t_c2
            bf16_leg["cpu_baseline"] = {"value": round(slice_qps * first_chunk.shape[0] / BR, 2), "unit": "queries/s", "cores": ncores,
                                        "kind": "port",
                                        "sample": f"oracle scan_topk_bf16 (half_precision.rs semantics: bf16-rounded inputs, f32 "
                                                  f"accumulate) on a {first_chunk.shape[0]}-row slice x {nq2} queries, {ncores} threads, "
                                                  f"{t_c2:.2f} s = {slice_qps:.1f} q/s on the slice; value = that x "
                                                  f"{first_chunk.shape[0]}/{BR} (extrapolated to the full corpus)"}
            bf16_leg["gpu_over_cpu"] = round(bf16_leg["qps"] / max(bf16_leg["cpu_baseline"]["value"], 1e-9), 1)
        ix3.close()
        torch.cuda.empty_cache()
        if not a.no_traffic_pass:  # FETCH_SIZE of the timed region (seed sweep, LDS-DMA launches, merges) in a child pass of this run
            tb_, tk_, tsrc_ = traffic_pass("bf16", ("sweep_topk_gemm", "merge_topk", "seed_tau", "round_queries_bf16"),
                                           ["--bf16-rows", str(BR)], child_steps=2, child_warm=1)
            bf16_leg["roofline"]["traffic"] = tb_
            bf16_leg["roofline"]["traffic_source"] = tsrc_
            if tb_ is not None:
                bf16_leg["roofline"]["traffic_by_kernel"] = tk_
                bf16_leg["roofline"]["traffic_over_algorithmic"] = round(tb_ / (BR * D * 2 + BR * 4 + BQ * D * 2), 3)

    # ---- the exact sweep for the other metrics of the path (N = 1 only): same N x D corpus (Hamming / Jaccard: the
    # reference's x > 0.5 threshold of it, swept as packed bits), one query per launch and --batch queries per launch
    metrics_leg = None
    if world == 1 and not a.no_metrics_leg:
        metrics_leg = []
        g.manual_seed(42)
        base = torch.randn((N, D), generator=g, device=dev, dtype=torch.float32)
        for mname, mm in (("euclidean", va.DistanceMetric.Euclidean), ("dot", va.DistanceMetric.DotProduct),
                          ("hamming", va.DistanceMetric.Hamming), ("jaccard", va.DistanceMetric.Jaccard)):
            if mname == a.metric:
                continue
            bits_metric = mname in ("hamming", "jaccard")
            src = (base > 0.5).float() if bits_metric else base
            qsrc = (queries[:Q] > 0.5).float() if bits_metric else queries[:Q].contiguous()
            ixm = va.HnswIndex(D, mm, va.HnswParams(a.M, a.efc, N), device=local)
            torch.cuda.synchronize()
            ixm.upload_dev(0, src.data_ptr(), N, stream)
            torch.cuda.synchronize()
            row = {"metric": mname}
            # (the batch: 20 untimed calls first — the chip settles into a matrix-core kernel mix over ~15 calls, see `settle_steps`; the
            # five calls behind two warm-ups that rounds 2-5 timed here read 10 % slow for that reason alone)
            for nq_m, reps, warm_m in ((1, 20, 2), (Q, 10, 20)):
                for _ in range(warm_m):
                    ixm.search_batch_dev(qsrc.data_ptr(), nq_m, K, 0, va.MODE_BRUTE, out_ids.data_ptr(), out_sc.data_ptr(),
                                         out_n.data_ptr(), stream)
                torch.cuda.synchronize()
                tm = time.perf_counter()
                for r_ in range(reps):
                    events_on_last(r_, reps)
                    ixm.search_batch_dev(qsrc.data_ptr(), nq_m, K, 0, va.MODE_BRUTE, out_ids.data_ptr(), out_sc.data_ptr(),
                                         out_n.data_ptr(), stream)
                torch.cuda.synchronize()
                mdt = (time.perf_counter() - tm) / reps
                kms_m, nl_m = ixm.last_kernel_ms()
                va.set_kernel_timing(False)
                key = "single_query" if nq_m == 1 else "batch"
                row[key] = {"queries": nq_m, "ms_per_call": round(mdt * 1e3, 4), "qps": round(nq_m / mdt, 1),
                            "sweep_kernel_ms": round(kms_m, 4), "launches": nl_m, "untimed_calls_first": warm_m, "timed_calls": reps}
            if bits_metric and row["batch"]["sweep_kernel_ms"] > 0 and (ixm.last_kernels() & va.KERNEL_BITS_GEMM):
                # the batch as a four-bit GEMM distance on the matrix cores (bits_gemm.hip): exact integer dot products
                bops = 2.0 * N * D * Q
                btops = bops / (row["batch"]["sweep_kernel_ms"] * 1e-3) / 1e12
                row["batch"]["roofline"] = {"bound": "mfma", "achieved": round(btops, 1), "peak": 10000.0, "unit": "TFLOP/s (fp4)", "frac": round(btops / 10000.0, 4),
                                            "traffic": None, "alg_bytes_per_batch": N * ((D + 255) // 256 * 128), "alg_ops_per_batch": bops,
                                            "kernel": "sweep_topk_gemm_bf16_pp<%s, FP4> (v_mfma_scale_f32_16x16x128_f8f6f4, unit scales; timed region = sample seed + launches + merges)" % mname,
                                            "note": "dense FP4 MFMA peak ~10 PFLOP/s (MI355X_MICROARCH.md); algorithmic operations = 2*rows*dim*queries"}
            if bits_metric and "roofline" in row["batch"] and not a.no_traffic_pass:
                # HBM bytes of the batch's kernels (seed, selection launches over the four-bit image, merges), in a child pass of this run
                tb_m, tk_m, tsrc_m = traffic_pass("bits_" + mname, ("sweep_topk_gemm_bf16_pp", "merge_topk", "seed_scores_fp4", "select_finish"), [])
                row["batch"]["roofline"]["traffic_source"] = tsrc_m
                if tb_m is not None:
                    row["batch"]["roofline"]["traffic"] = tb_m
                    row["batch"]["roofline"]["traffic_by_kernel"] = tk_m
                    row["batch"]["roofline"]["traffic_over_algorithmic"] = round(tb_m / row["batch"]["roofline"]["alg_bytes_per_batch"], 3)
            pass_bytes = N * ((D + 127) // 128 * 16) if bits_metric else N * D * 4
            sk = row["single_query"]["sweep_kernel_ms"]
            row["single_query"]["hbm_gbs"] = round(pass_bytes / (sk * 1e-3) / 1e9, 1) if sk > 0 else 0.0
            row["single_query"]["hbm_frac"] = round(pass_bytes / (sk * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sk > 0 else 0.0
            row["alg_bytes_per_pass"] = pass_bytes
            row["note"] = {"euclidean": "batch: WIDE selection on the bf16 matrix cores over s = q.v - |v|^2/2 (augmented DotProduct form), canonical "
                                        "(q - v)^2 re-scoring of the rows above the final bound, per-query proof, gathered exact pass for unproven queries",
                           "dot": "batch: the headline's selection stage (WIDE bf16 selection + exact re-scoring, proof by construction)",
                           "hamming": "packed bits (x > 0.5), 96 B/row; batches of >= 32 queries: +-1 four-bit image (48 B per 96 values), dim - 2 |q ^ v| as a four-bit GEMM on the "
                                      "matrix cores with the fused top-k (exact integers); smaller batches: 32 queries per corpus pass, AND+popcount",
                           "jaccard": "packed bits (x > 0.5), 96 B/row; batches of >= 32 queries: {0,1} four-bit image, |q & v| as a four-bit GEMM on the matrix "
                                      "cores, per-element bound d - c|v| >= c|q| in the epilogue (exact integers); smaller batches: AND+popcount"}[mname]
            if not a.no_cpu_baseline:
                # the reference's exact path for this metric restated on the host cores (mode R: wide16 kernels; Hamming /
                # Jaccard: the f32-threshold kernels of simd_explicit.rs), one query per thread, bounded sample
                from oracle import pyoracle as po
                pm = {"euclidean": po.EUCLIDEAN, "dot": po.DOT, "hamming": po.HAMMING, "jaccard": po.JACCARD}[mname]
                ncores = po.host_threads()
                host_rows = src.cpu().numpy()
                nq_c = min(ncores, Q)
                qh = qsrc[:nq_c].cpu().numpy()
                tc = time.perf_counter()
                r_ids, r_sc = po.scan_topk(pm, host_rows, qh, K, po.MODE_R, nthreads=ncores)
                cdt = time.perf_counter() - tc
                row["cpu_baseline"] = {"value": round(nq_c / cdt, 2), "unit": "queries/s", "cores": ncores, "kind": "port",
                                       "sample": f"oracle mode R exact scan of {N} rows x {nq_c} queries, {ncores} threads, {cdt:.2f} s"}
                row["gpu_over_cpu_batch"] = round(row["batch"]["qps"] / max(nq_c / cdt, 1e-9), 1)
                # parity of the batch just timed (its results are still in the output buffers): ids AND score bits against the oracle
                # in the arithmetic the index declares (C / M; integer work for the bit metrics), and the distance to the reference's own
                # summation order (mode R) — the north-star's 1e-5
                c_mode = po.MODE_M if ixm.sweep_arith_mode(K) == "M" else po.MODE_C
                c_ids, c_sc = po.scan_topk(pm, host_rows, qh, K, c_mode, nthreads=ncores)
                g_ids = out_ids[:nq_c].cpu().numpy().astype(np.int64)
                g_sc = out_sc[:nq_c].cpu().numpy()
                denom = np.maximum(np.abs(r_sc.astype(np.float64)), 1e-30)
                row["parity_check"] = {
                    "queries": int(nq_c), "of_a_batch_of": int(Q),
                    "ids_equal_oracle": bool(np.array_equal(g_ids, c_ids.astype(np.int64))),
                    "scores_bit_equal_oracle": bool(np.array_equal(g_sc.view(np.uint32), np.ascontiguousarray(c_sc, dtype=np.float32).view(np.uint32))),
                    "within_1e5_of_reference_order": bool(np.all(np.abs(g_sc.astype(np.float64) - r_sc.astype(np.float64)) / denom <= 1e-5)),
                    "max_rel_diff_vs_reference_order": float(np.max(np.abs(g_sc.astype(np.float64) - r_sc.astype(np.float64)) / denom))}
                del host_rows
            metrics_leg.append(row)
            ixm.close()
            del src
            torch.cuda.empty_cache()
        del base
        torch.cuda.empty_cache()

    # ---- range-sharded mode (BASELINE configs[4]), every run: per-shard top-k + ONE all-gather of k 12-byte (id, score)
    # records per query and shard + merge kernel, all behind the C ABI (csrc/shard_group.hip).
    #   * one process per GPU (torchrun, also the 1-rank default run): every rank's index joins the RCCL group
    #     (vdb_hip_index_join_group); default: rank r's N rows are shard r of a world*N-row corpus; `--shard-rows
    #     6250000` at 8 GPUs = configs[4] (50 M rows, every rank generates its own shard on the device).
    #   * at one GPU additionally: ONE handle over two co-located range shards of the same corpus
    #     (vdb_hip_index_create(devices=[0, 0], VDB_SHARD_RANGE)): the merged result must equal the unsharded one bit for bit.
    sharded = None
    if not a.no_sharded_leg:
        from velesdb_amd.sharded import join_process_group
        sharded = {}
        ref_ids = ref_sc = None
        if world == 1:
            step(0)
            torch.cuda.synchronize()
            ref_ids, ref_sc = out_ids.cpu().numpy().copy(), out_sc.cpu().numpy().copy()
            if host_sample is not None and sample_rows == N:
                gx = va.HnswIndex(D, metric, va.HnswParams(a.M, a.efc, N), devices=[local, local], shard_mode=va.SHARD_RANGE)
                gx.upload(np.arange(N, dtype=np.uint64), host_sample)
                off0 = (0 * world + rank) * Q % (n_query_pool - Q + 1)

                def gstep():
                    gx.search_batch_dev(queries[off0:off0 + Q].data_ptr(), Q, K, 0, va.MODE_BRUTE, out_ids.data_ptr(),
                                        out_sc.data_ptr(), out_n.data_ptr(), stream)

                gstep()
                torch.cuda.synchronize()
                same = bool(np.array_equal(out_ids.cpu().numpy(), ref_ids) and
                            np.array_equal(out_sc.cpu().numpy().view(np.uint32), ref_sc.view(np.uint32)))
                tg = time.perf_counter()
                for _ in range(a.steps):
                    gstep()
                torch.cuda.synchronize()
                gdt = (time.perf_counter() - tg) / a.steps
                sharded["one_handle_two_range_shards"] = {
                    "qps": round(Q / gdt, 1), "ms_per_step": round(gdt * 1e3, 4), "shards": 2, "rows_per_shard": N // 2,
                    "transport": gx.shard_info()["transport"], "equals_unsharded_bitwise": same,
                    "note": "co-located on one GPU: the two shard sweeps run one after the other"}
                gx.close()
                torch.cuda.empty_cache()
        SR = a.shard_rows if a.shard_rows > 0 else N
        ix_sh = ix
        ok = 1
        try:
            if SR != N or world > 1:  # every rank generates its own shard (ids = global rows rank * SR ...)
                gs = torch.Generator(device=dev)
                gs.manual_seed(4242 + rank)
                ix_sh = va.HnswIndex(D, metric, va.HnswParams(a.M, a.efc, SR), device=local)
                chunk = 1_000_000
                for base in range(0, SR, chunk):  # bounded staging: 3 GB at a time
                    n_c = min(chunk, SR - base)
                    c = torch.randn((n_c, D), generator=gs, device=dev, dtype=torch.float32)
                    torch.cuda.synchronize()
                    ix_sh.upload_dev(rank * SR + base, c.data_ptr(), n_c, stream)
                    del c
        except Exception as e:  # noqa: BLE001 - a shard that cannot be set up must not hang the collective
            print(f"[rank {rank}] shard setup failed: {e}", file=sys.stderr)
            ok = 0
        if use_dist:
            t_ok = torch.tensor([ok], device=dev, dtype=torch.int32)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)  # every rank agrees before the first data-path collective
            ok = int(t_ok.item())
        if ok == 1:
            # a group that cannot be formed (RCCL not loadable, id exchange refused ...) must cost the sharded leg, not the line: the
            # replica headline above needs no collective.  (The agreement below runs on torch's own group, not on the library's.)
            try:
                join_process_group(ix_sh, rank, world, dev)
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] joining the shard group failed: {e}", file=sys.stderr)
                sharded["join_error"] = str(e)[:300]
                ok = -1
            if use_dist:
                t_ok = torch.tensor([ok], device=dev, dtype=torch.int32)
                dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
                ok = int(t_ok.item())
        if ok == 1:
            # self-diagnosis of the first real multi-GPU run: what every rank's handle says about itself
            inf = ix_sh.shard_info()
            mine = torch.tensor([rank, inf["world"], inf["rank"], {"none": 0, "rccl": 1, "d2d-copy": 2}.get(inf["transport"], -1),
                                 ix_sh.len()], device=dev, dtype=torch.int64)
            allinf = [torch.zeros_like(mine) for _ in range(world)] if use_dist else [mine]
            if use_dist:
                dist.all_gather(allinf, mine)
            per_rank = [{"rank": int(t[0]), "group_world": int(t[1]), "group_rank": int(t[2]),
                         "transport": {0: "none", 1: "rccl", 2: "d2d-copy"}.get(int(t[3]), "?"), "rows": int(t[4])} for t in allinf]
            sharded["per_rank"] = per_rank
            sharded["group_ok"] = bool(all(r_["transport"] == "rccl" and r_["group_world"] == world and r_["group_rank"] == r_["rank"]
                                           for r_ in per_rank))
            if not sharded["group_ok"]:
                sharded["error"] = "the shard group is not what --gpus asked for (see per_rank): every rank must report transport rccl, group_world == n_gpus"

            def sharded_step(i):
                off = (i * Q) % (n_query_pool - Q + 1)  # every rank searches the SAME queries against ITS shard
                ix_sh.search_batch_dev(queries[off:off + Q].data_ptr(), Q, K, 0, va.MODE_BRUTE, out_ids.data_ptr(),
                                       out_sc.data_ptr(), out_n.data_ptr(), stream)

            for i in range(a.warmup):
                sharded_step(i)
            barrier()
            t2 = time.perf_counter()
            for i in range(a.steps):
                sharded_step(a.warmup + i)
            barrier()
            sdt_mine = time.perf_counter() - t2
            sdt = max_over_ranks(sdt_mine)
            # every rank ends every step with the GLOBAL top-k: the ranks' results must be identical (a checksum of checksums —
            # the size-independent parity property of the sharded path; the world-1 run also compares with the unsharded bits)
            torch.cuda.synchronize()
            csum = float((out_ids.to(torch.float64).sum() + out_sc.to(torch.float64).sum()).item())
            rows_sh = gather_rows([float(rank), Q * a.steps / sdt_mine, csum], ("rank", "qps", "result_checksum"))
            sharded["per_rank_rate"] = rows_sh
            sharded["results_identical_across_ranks"] = bool(all(r_["result_checksum"] == rows_sh[0]["result_checksum"] for r_ in rows_sh))
            sharded.update({"qps": round(Q * a.steps / sdt, 1), "corpus_rows": world * SR, "rows_per_shard": SR,
                            "ranks": world, "ms_per_step": round(sdt / a.steps * 1e3, 4),
                            "transport": ix_sh.shard_info()["transport"],
                            "collective": "one ncclAllGather of %d B/query/GPU (12-byte (u64 id, f32 score) records) + "
                                          "merge_shards_topk, inside libvelesdb_hip.so" % (K * 12)})
            if world == 1 and SR == N and ref_ids is not None:
                sharded_step(0)
                torch.cuda.synchronize()
                sharded["equals_unsharded_bitwise"] = bool(
                    np.array_equal(out_ids.cpu().numpy(), ref_ids) and
                    np.array_equal(out_sc.cpu().numpy().view(np.uint32), ref_sc.view(np.uint32)))
        else:
            sharded["error"] = ("shard setup failed on at least one rank (see stderr)" if ok == 0 else
                                "the shard group could not be joined on at least one rank (see stderr / join_error)")
            sharded["group_ok"] = False
        if ix_sh is not ix:
            ix_sh.close()
            torch.cuda.empty_cache()

    device_state = None
    if rank == 0:
        # which box is this?  The HBM-bound traversal leg differs by up to 15 % between boxes of the pool and by < 1 % within one
        # (profiles/r02ag_hnsw_spread.log): memory / fabric clock levels and the partition modes are recorded with the line
        try:
            import subprocess as sp_
            out = sp_.run(["rocm-smi", "--showclocks", "--showmemorypartition", "--showcomputepartition", "--showpower"],
                          capture_output=True, text=True, timeout=30).stdout
            device_state = [ln.split(":", 1)[1].strip() if ln.startswith("GPU[0]") else ln.strip()
                            for ln in out.splitlines() if ln.startswith("GPU[0]")]
        except Exception as ex:  # noqa: BLE001
            device_state = [f"rocm-smi unavailable: {ex}"]
    if rank == 0:
        line = {
            "metric": "qps_at_recall10_1Mx768_k10", "value": round(qps, 1), "unit": "queries/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{N}x{D} f32 {a.metric}, k={K}, exact distance sweep + fused GPU top-k "
                                   f"(BASELINE configs[1]); {Q} queries/step, {tile} queries per corpus pass "
                                   f"({roofline['kernel']})",
                       "rows": N, "dim": D, "k": K, "queries_per_step": Q,
                       "parallelism": "replicas x%d (query stream split, no collective)" % world},
            "replicas_per_rank": replicas_per_rank, "repeat_ms_per_step": round(repeat_ms_per_step, 4), "settle_steps": max(0, a.settle_steps),
            "recall_at_10": recall, "parity_check": check,
            # the headline's algorithmic flop over the WHOLE step's time / peak — from ms_per_step (the timed region's wall clock over its
            # steps, what `value` is computed from), not from the last step's event pair (roofline.whole_batch keeps that one)
            "frac_step": (round(2.0 * N * D * Q / (dt / a.steps) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4) if split_active else (roofline.get("whole_batch") or {}).get("frac")),
            "roofline": roofline, "cpu_baseline": cpu, "latency_mode": lat, "tiles": tiles, "batch_sizes_default_path": batch_sizes, "k_curve": k_curve, "host_entry": host_entry, "sharded": sharded,
            "hnsw": hnsw, "hnsw_embedding_like": hnsw_emb, "hnsw_m128": hnsw_m128, "config0_10k": config0, "bf16_gemm": bf16_leg, "sq8_storage_mode": sq8_leg, "other_metrics": metrics_leg,
            "device": va.device_name(local), "device_state": device_state,
        }
    # a multi-GPU run that is not what --gpus asked for must not pass for one: rank 0 checks the group the library reports
    bad_group = None
    if rank == 0 and world > 1:
        if a.gpus != world:
            bad_group = f"--gpus {a.gpus} but the process group has {world} ranks"
        elif sharded is not None and not sharded.get("group_ok", False):
            bad_group = "the shard group is not world == n_gpus over transport rccl on every rank: " + json.dumps(sharded.get("per_rank") or sharded.get("error"))
        elif sharded is not None and sharded.get("results_identical_across_ranks") is False:
            bad_group = "the ranks of the sharded leg ended with different merged results"
        if bad_group:
            line["error"] = bad_group
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # the full record: a file next to the line (and under gpurun_out/ when that exists, so a gpurun call brings it home);
        # stdout carries ONE compact line, last
        full_txt = json.dumps(line)
        here = os.path.dirname(os.path.abspath(__file__))
        legs_file = "bench_legs.json" if world == 1 else "bench_legs_%dgpu.json" % world
        for d_ in (here, os.path.join(here, "gpurun_out")):
            try:
                if os.path.isdir(d_):
                    with open(os.path.join(d_, legs_file), "w") as f:
                        f.write(full_txt + "\n")
            except OSError as ex:
                print(f"bench.py: could not write {legs_file} in {d_}: {ex}", file=sys.stderr)
        # (not echoed to stderr: the driver's record keeps ONE bounded tail of stdout + stderr, and 25 KB of stderr would push the line
        # out of it just as the 25 KB line pushed its own head out in round 4)
        print(f"bench.py: full record ({len(full_txt)} bytes) in {legs_file}", file=sys.stderr)
        sys.stderr.flush()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(compact_line(line, legs_file)) + "\n").encode())
        if bad_group:
            raise SystemExit("bench.py: " + bad_group)


if __name__ == "__main__":
    main()
