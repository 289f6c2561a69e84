"""The launch schedule of the 256 x 256 selection kernel, checked on the CPU at sizes no GPU test reaches.

`velesdb_amd/csrc/vdb_gemm_schedule.hpp` (host arithmetic, the text the library compiles) cuts a corpus into launches for the
selection stage, the bf16 result path (BASELINE configs[3], 10 M rows) and the bit metrics; `tests/gemm_schedule_model.cpp`
walks ~195 000 schedules — 1 row ... 2^32 - 512 rows (the per-index limit), 6.25 M rows per shard (configs[4]), 1 ... 4 096
queries, several chip sizes and head / launch-length settings — and checks that every (row tile, query tile) pair is reached
exactly once by the kernel's block map, that launches are consecutive and tile-aligned, that every partial list has its own
slot, and that nothing wraps in 32 bits (built with ASan + UBSan).  A schedule that skips a tile is a silently wrong top-k:
`HnswIndex::search_brute_force` looks at every vector (index/hnsw/index/search.rs:176-219).  The model found one wrap — the
tile count of a row range within 255 rows of 2^32 — fixed in the header, and the per-index row limit now keeps whole-tile
arithmetic inside 32 bits (`kMaxRowsPerIndex`).
"""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
def test_every_schedule_covers_every_tile_exactly_once(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "gemm_schedule_model")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wall", "-Wextra",
                           "-Werror", "-I", os.path.join(ROOT, "velesdb_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "gemm_schedule_model.cpp")])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    env.pop("LD_PRELOAD", None)   # the binary links its own sanitizer runtime
    r = subprocess.run([exe], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["ok"] and line["violations"] == 0
    assert line["cases"] > 150_000 and line["launches_walked_tile_by_tile"] > 1_000_000
