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
    assert a.select_level == 2 and not a.no_split and a.engine == 1 and a.tile == 128          # the library's default path, nothing forced
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
