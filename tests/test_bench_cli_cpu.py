"""bench.py's contract, as far as a CPU can check it: with no flags it is the 1-GPU run of BASELINE's configuration (1 000 000 x 768 f32
cosine, k = 10, 1 024 queries per step) with a step / warm-up count that finishes in minutes; and without a GPU it fails LOUDLY — no
CPU fallback, no JSON line that could be mistaken for a measurement."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_defaults_are_the_baseline_configuration(monkeypatch):
    b = load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (1, 20, 3)
    assert (a.rows, a.dim, a.k, a.batch, a.metric) == (1_000_000, 768, 10, 1024, "cosine")     # BASELINE.json configs[1]
    assert (a.hnsw_batch, a.ef, a.M, a.efc) == (8192, 128, 32, 400)                            # configs[2]: HnswParams::auto(768), Balanced
    assert a.bf16_rows == 10_000_000 and a.shard_rows == 0                                     # configs[3]; configs[4] = --shard-rows 6250000 at 8 GPUs
    assert a.select_level == 3 and not a.no_split and a.engine == 1 and a.tile == 128          # the library's default path, nothing forced
    assert b.HBM_PEAK_GBS == 8000.0
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 5, 2)


def test_without_a_gpu_the_bench_fails_loudly_and_prints_no_line():
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr
    assert r.stdout.strip() == ""          # the JSON line goes to stdout and only after a measurement


def test_world_size_must_match_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300, env=env,
                       cwd=ROOT)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


def test_graft_entry_builds_and_smoke_refuses_to_run_without_a_gpu():
    """__graft_entry__.build() is idempotent on a built tree (every product / checker / helper object present afterwards);
    smoke() without a GPU stops at its first line instead of checking anything against anything"""
    import pytest
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    for rel in ("velesdb_amd/lib/libvelesdb_hip.so", "oracle/libvdb_oracle.so", "tests/stub_rccl/libstub_rccl.so", "tests/abi_driver",
                "tools/libcallers_bench.so"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
    with pytest.raises(AssertionError, match="needs a GPU"):
        g.smoke()


def _maximal_record(blow=1):
    """a synthetic full record with every leg present and every prose field `blow` times longer than a real run's"""
    prose = "x" * (400 * blow)
    roof = {"bound": "mfma", "achieved": 1078.9, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.4316, "traffic": 2301561600,
            "traffic_over_algorithmic": 0.747, "kernel": "sweep_topk_gemm_bf16_pp<cosine> " + prose, "kernel_ms": 1.4518, "launches_timed": 4,
            "note": prose, "kernel_ms_note": prose, "traffic_by_kernel": {("k%d" % i) + prose: i for i in range(8)},
            "exact_f32_kernel": {"roofline": {"note": prose}}}
    cpu = {"value": 78.8, "unit": "queries/s", "cores": 16, "kind": "port", "cpu_model": "AMD EPYC 9575F 64-Core Processor " + prose,
           "sample": prose, "shape_a": {"qps": 1.0}, "shape_b": {"qps": 1.0}, "cores_note": prose}
    pts = [{"threads": t, "qps": 1.0, "gpu_over_cpu": 0.5, "note": prose} for t in (1, 4, 16, 64)]
    return {
        "metric": "qps_at_recall10_1Mx768_k10", "value": 598577.0, "unit": "queries/s", "n_gpus": 8, "steps": 20, "warmup": 3,
        "ms_per_step": 1.7107, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "1000000x768 f32 cosine " + prose, "rows": 1000000, "dim": 768, "k": 10, "queries_per_step": 1024,
                   "parallelism": "replicas x8 (query stream split, no collective)"},
        "replicas_per_rank": [{"rank": r, "qps": 1.0} for r in range(8)], "recall_at_10": 1.0,
        "parity_check": {"queries": 64, "ids_equal_oracle_canonical": True, "scores_bit_equal_oracle_canonical": True, "kernel": prose},
        "frac_step": 0.38, "settle_steps": 40, "repeat_ms_per_step": 1.7001, "roofline": roof, "cpu_baseline": cpu, "latency_mode": {"note": prose},
        "tiles": [{"tile": t, "note": prose} for t in range(12)], "batch_sizes_default_path": [{"n": prose}] * 8,
        "host_entry": {"note": prose, "calls": [{"queries_per_call": n, "qps": 1000.0 * n, "note": prose} for n in (1024, 256, 64, 16, 1)],
                       "threads": [{"threads": t, "queries_per_call": 1024, "qps": 2.0 * t, "note": prose} for t in (2, 4)]},
        "k_curve": [{"k": k, "qps": 100.0 * k, "note": prose} for k in (10, 11, 50, 100)],
        "sharded": {"qps": 1.0, "group_ok": True, "ranks": 8, "rows_per_shard": 6250000, "transport": "rccl", "results_identical_across_ranks": True,
                    "per_rank": [{"rank": r, "note": prose} for r in range(8)], "collective": prose},
        "hnsw": {"qps": 167342.2, "recall_at_10": 0.99, "roofline": dict(roof, bound="hbm"), "parity_check": {"ok": True, "n": prose},
                 "cpu_baseline": cpu, "build_inserts_per_s": 44566.8, "build": {"roofline": {"frac": 0.2, "note": prose}},
                 "int8": {"qps": 1.0, "hbm_frac": 0.6, "note": prose}, "ef_curve": [{"ef": e, "note": prose} for e in (64, 128, 256, 512)],
                 "latency_mode": [{"queries_per_call": 1, "median_us_per_call": 1051.8}], "concurrent_callers": {"points": pts}},
        "hnsw_embedding_like": {"note": prose, "ef_curve": [{"note": prose}] * 4, "qps": 232000.0, "recall_at_10": 0.985, "roofline": dict(roof, bound="hbm")},
        "hnsw_m128": {"workload": prose, "qps": 51460.5, "recall_at_10": 0.1775, "roofline": dict(roof, bound="hbm"), "build_inserts_per_s": 4666.6,
                      "build": {"roofline": {"frac": 0.43, "note": prose}}, "cpu_baseline": dict(cpu, recall_at_10=0.1775),
                      "latency_mode": [{"queries_per_call": 1, "median_us_per_call": 2655.0}], "parity_check": {"ok": True, "n": prose}},
        "config0_10k": {"search_median_us": 304.1, "reference_published": {"search_us": 56.8}, "concurrent_callers": {"points": pts}, "note": prose},
        "bf16_gemm": {"qps": 1.0, "roofline": roof, "parity_check": {"ok": True, "rule": prose}, "cpu_baseline": cpu},
        "sq8_storage_mode": {"batch": {"qps": 1.0, "kernel": prose}, "eight_queries": {"hbm_frac": 0.12}, "parity_check": {"ids_equal_oracle": True}},
        "other_metrics": [{"metric": m, "single_query": {"ms_per_call": 0.5, "hbm_frac": 0.7}, "batch": {"qps": 1.0, "roofline": roof},
                           "parity_check": {"ids_equal_oracle": True}, "note": prose, "cpu_baseline": cpu}
                          for m in ("euclidean", "dot", "hamming", "jaccard")],
        "device": "AMD Radeon Graphics (gfx950:sramecc+:xnack-, 256 CUs) " + prose, "device_state": [prose] * 7, "error": prose,
    }


def test_the_printed_line_is_bounded_and_carries_the_contract():
    """round 4's 25 KB line was cut by the driver's bounded record (BENCH_r04.parsed = null).  The printed line is built by a pure
    function; whatever the legs hold it stays under 4 KB, round-trips through json, and keeps the contract's keys plus the headline's
    roofline and cpu_baseline objects"""
    import json
    b = load_bench()
    for blow in (1, 10, 100):
        full = _maximal_record(blow)
        assert len(json.dumps(full)) > 25_000
        line = b.compact_line(full, "bench_legs_8gpu.json")
        enc = json.dumps(line)
        assert len(enc) <= b.COMPACT_LIMIT < 8192, (blow, len(enc))
        assert "\n" not in enc
        back = json.loads(enc)
        assert back == line
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                  "data", "config", "roofline", "cpu_baseline", "recall_at_10", "parity_check"):
            assert k in back, k
        assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"] and back["n_gpus"] == 8
        assert back["config"]["workload"].startswith("1000000x768 f32 cosine")
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert back["roofline"][k] == full["roofline"][k]
        for k in ("value", "unit", "cores", "kind"):
            assert back["cpu_baseline"][k] == full["cpu_baseline"][k]
        assert back["parity_check"] is True and back["legs_file"] == "bench_legs_8gpu.json"
    # at a real run's sizes nothing is dropped: every leg's one-number summary is on the line
    back = b.compact_line(_maximal_record(1))
    assert set(back["legs"]) == {"hnsw", "hnsw_embedding_like", "hnsw_m128", "bf16_gemm", "sharded", "config0_10k", "sq8", "other_metrics"}
    # round 6: what a VectorIndex caller gets (host pointers), the k the reference also benches, the graph legs that say something
    assert back["host_entry_qps"] == {"1024": 1024000.0, "256": 256000.0, "64": 64000.0, "2x1024": 4.0, "4x1024": 8.0}
    assert back["k50_qps"] == 5000.0 and back["k_curve_qps"] == {"10": 1000.0, "11": 1100.0, "50": 5000.0, "100": 10000.0}
    # the untimed steps in front of the W warm-up steps are SAID on the line, with the diagnostic that shows what they are for
    assert "settle_steps" in back and "repeat_ms_per_step" in back
    assert back["settle_steps"] == 40 and back["repeat_ms_per_step"] == 1.7001
    assert back["legs"]["hnsw_embedding_like"] == {"qps": 232000.0, "recall": 0.985, "frac": 0.4316, "cpu_qps": None}
    assert back["legs"]["hnsw_m128"]["build_frac"] == 0.43 and back["legs"]["hnsw_m128"]["one_query_us"] == 2655.0 and back["legs"]["hnsw_m128"]["parity"] is True
    assert back["legs"]["hnsw"]["frac"] == 0.4316 and back["legs"]["hnsw"]["build_frac"] == 0.2 and back["legs"]["sharded"]["group_ok"] is True
    assert back["legs"]["other_metrics"]["hamming"]["parity"] is True
    # a failed parity flag anywhere in a check object reads as False on the line; no check object reads as None
    bad = _maximal_record(1)
    bad["parity_check"]["scores_bit_equal_oracle_canonical"] = False
    assert b.compact_line(bad)["parity_check"] is False
    bad.pop("parity_check")
    assert b.compact_line(bad)["parity_check"] is None
    # a minimal record (every optional leg skipped) still makes a line
    mini = {k: v for k, v in _maximal_record(1).items() if k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                                 "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline")}
    m = b.compact_line(mini)
    assert m["legs"] == {} and m["cpu_baseline"] is None and m["roofline"]["frac"] == 0.4316


def test_committed_full_records_compact_under_the_limit():
    """every full bench record the builder committed under profiles/ goes through compact_line under the limit"""
    import glob
    import json
    b = load_bench()
    seen = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_line_1gpu.json"))):
        full = json.loads(open(path).read().strip().splitlines()[-1])
        if "legs_file" in full:     # already a compact line
            continue
        enc = json.dumps(b.compact_line(full))
        assert len(enc) <= b.COMPACT_LIMIT, (path, len(enc))
        assert json.loads(enc)["value"] == full["value"]
        seen += 1
    assert seen >= 10
