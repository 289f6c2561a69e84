"""The diagnostic environment switches (A / B probes) exist only in the PROBE build of the library (libvelesdb_hip_probe.so, the same
sources with -DVDB_PROBE_SWITCHES: csrc/vdb_probe_env.hpp); the shipped libvelesdb_hip.so reads no environment variable
(tests/test_abi_exports.py::test_shipped_library_reads_no_environment_variable).  DESIGN §6b claims that no result depends on a
switch.  This file PROVES it for the switches that select another kernel or another schedule: the parity tests of the path a
switch touches are run again, in a child process with the switch set — the same assertions (ids, ranks, score bits, counters
against the oracle) must hold.

  VELESDB_BF16_PP=0               lock-step LDS-DMA kernel instead of the ping-pong one      -> split / bf16 result-mode tests
  VELESDB_BF16_SEED=0             exact f32 seed sweep instead of the bf16 sample seed       -> split tests
  VELESDB_SEL_STEPS=2,0,0         another launch schedule of the selection stage             -> split tests
  (VELESDB_SELECT_MIN_QUERIES only moves the size from which the stage takes a batch — tests/test_gpu_split.py compares the stage
   with the exact kernels on both sides of it, and asserts WHICH path served a batch, so it is not re-run under another value)
  VELESDB_HNSW_LATENCY_MODE=0|2|3 throughput kernel only / latency-mode kernel forced with and without row speculation -> graph tests
  VELESDB_HNSW_VIS_LDS=1          LDS visited set in the throughput walk                      -> graph tests
  VELESDB_HNSW_PREFETCH_IDS=1     latency-mode walk with the neighbour-list prediction over cache-resident corpora too -> graph tests
  VELESDB_INT8_VIS_LDS=1, VELESDB_I8_WAVES2=0   the int8 walk's variants                      -> int8 tests
  VELESDB_BITS_FUSED_BLOCKS=37|3  the one-launch packed-bit search with another block count (every wave several batches of chunks, a
                                  ragged block grid; 3 blocks: fewer lists than k)           -> the one-launch tests
  VELESDB_BITS_FUSED=0            the three-launch form for one or two packed-bit queries     -> the one-launch tests
  VELESDB_MERGE_EXTRACT=0         the small merges on merge_topk_select instead of merge_topk_extract -> split tests
  VELESDB_COSINE_NORMALISED=0     round 5's Cosine selection (plain bf16 copy + row norms in the kernel) instead of the normalised images
                                                                                              -> split tests, the WIDE tests (k > 10)
  VELESDB_WIDE_SMALL_K=0          k <= 10 stays on the block-local lists at selector level 3 (the default) too           -> split tests
  VELESDB_WIDE_FUSE=0             the WIDE selection's final bound / pool by a wide_reseed launch of its own             -> wide-k tests
  VELESDB_POOL_SELECT=0           bounds / final pool of a selection batch by merge_topk_* instead of radix selection -> split tests
  VELESDB_GATHER_ALL=0            unproven queries: gathered pass up to 96 + the GEMM-structured fallback beyond (rounds 2-5) -> split tests
  (VELESDB_BITS_FUSED_SKIP is an ablation that returns WRONG results by design — probe timing only, nothing to re-run)
"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

va = pytest.importorskip("velesdb_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (a subset per switch keeps the file to a few minutes: the small random-data cases at both selection levels, the tie / duplicate /
# non-finite / soft-delete cases, the Euclidean form)
SPLIT = ["tests/test_gpu_split.py", "-k",
         "(random_data and (70000-128 or 66000-64)) or massive_exact_ties or zero_huge or soft_deleted or (euclidean_batches and 70000-128-256)"]
BF16 = ["tests/test_gpu_bf16.py", "-k", "glds_exact_products and 70077"]
GRAPH = ["tests/test_gpu_hnsw.py"]
INT8 = ["tests/test_gpu_int8.py"]
ONE_LAUNCH = ["tests/test_gpu_round5_parity.py", "-k", "one_launch and (300000 or 70001 or 4100 or 130 or 64-128)"]
# (everything of test_gpu_split.py that leaves queries unproven as well: near-duplicate clusters, NaN / inf rows, level parking)
SPLIT_ALL = ["tests/test_gpu_split.py", "-k", "not 1000000 and not 300000 and not 150001"]
WIDE = ["tests/test_gpu_wide_k.py", "-k", "vs_oracle or adversarial"]

CASES = [
    ({"VELESDB_BF16_PP": "0"}, SPLIT),
    ({"VELESDB_BF16_PP": "0"}, BF16),
    ({"VELESDB_BF16_SEED": "0"}, SPLIT),
    ({"VELESDB_SEL_STEPS": "2,0,0"}, SPLIT),
    ({"VELESDB_HNSW_LATENCY_MODE": "0"}, GRAPH),
    ({"VELESDB_HNSW_LATENCY_MODE": "2"}, GRAPH),
    ({"VELESDB_HNSW_LATENCY_MODE": "3"}, GRAPH),
    ({"VELESDB_HNSW_VIS_LDS": "1"}, GRAPH),
    ({"VELESDB_HNSW_PREFETCH_IDS": "1"}, GRAPH),
    ({"VELESDB_INT8_VIS_LDS": "1"}, INT8),
    ({"VELESDB_I8_WAVES2": "0"}, INT8),
    ({"VELESDB_BITS_FUSED_BLOCKS": "37"}, ONE_LAUNCH),
    ({"VELESDB_BITS_FUSED_BLOCKS": "3"}, ONE_LAUNCH),
    ({"VELESDB_BITS_FUSED": "0"}, ONE_LAUNCH),
    ({"VELESDB_MERGE_EXTRACT": "0"}, SPLIT),
    ({"VELESDB_COSINE_NORMALISED": "0"}, SPLIT),
    ({"VELESDB_COSINE_NORMALISED": "0"}, WIDE),
    ({"VELESDB_WIDE_SMALL_K": "0"}, SPLIT),
    ({"VELESDB_WIDE_FUSE": "0"}, WIDE),
    ({"VELESDB_POOL_SELECT": "0"}, SPLIT),
    ({"VELESDB_GATHER_ALL": "0"}, SPLIT_ALL),
]


@pytest.mark.parametrize("env,target", CASES, ids=[",".join(f"{k}={v}" for k, v in e.items()) + ":" + t[0].split("/")[-1] for e, t in CASES])
def test_parity_holds_under_the_switch(gpu_required, env, target):
    child_env = dict(os.environ, VDB_TEST_PROBE_LIB="1", **env)   # (tests/conftest.py binds the child to libvelesdb_hip_probe.so)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", *target], cwd=ROOT, env=child_env,
                       capture_output=True, text=True, timeout=900)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, f"parity broke under {env}:\n{tail}"
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
