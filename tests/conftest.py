import os
import sys

import pytest

# torch first (as bench.py does): PyTorch-ROCm bundles its own HIP runtime, and a process that loaded ROCm's copy through
# libvelesdb_hip.so before torch initialised reports "No HIP GPUs are available" from torch afterwards.  Tests that hand
# torch device pointers to the C ABI need both in one process.
try:
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
except Exception:  # noqa: BLE001 - torch is optional for everything but the device-pointer tests
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# A child pytest of tests/test_gpu_switches.py runs the parity tests under an environment switch: switches exist only in the probe build
# of the library (csrc/vdb_probe_env.hpp), which this harness — not the package — selects.
# (VDB_TEST_LIB=<path>: the same for a kernel-variant build under tools/probes/out/ — A / B runs of the parity tests)
if os.environ.get("VDB_TEST_PROBE_LIB") == "1" or os.environ.get("VDB_TEST_LIB"):
    from velesdb_amd import _ffi as _vdb_ffi
    _vdb_ffi.use_library(os.environ.get("VDB_TEST_LIB") or _vdb_ffi.PROBE_LIB_PATH)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present() -> bool:
    try:
        from velesdb_amd import device_count
        return device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    if not _gpu_present():
        pytest.skip("no HIP device visible")


@pytest.fixture(autouse=True)
def _library_defaults_for_gpu_tests(request):
    """Every `-m gpu` test starts from the library's process-wide defaults: several tests pin a selection level / engine / tile and leave
    it pinned (the knobs are process-wide), and what a later test asserts about WHICH path served a call must not depend on the order
    of the files."""
    if request.node.get_closest_marker("gpu") is not None and _gpu_present():
        import velesdb_amd as va
        va.set_split_selector(3)
        va.set_sweep_engine(1)
        va.set_max_query_tile(128)
    yield
