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
