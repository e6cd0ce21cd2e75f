import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _gpu_count():
    # PyTorch-ROCm bundles its own HIP runtime; when a test uses torch tensors as device buffers, torch has
    # to initialise the GPU BEFORE libmathaudio_hip.so pulls in a runtime of the same soname (the loader then
    # shares torch's copy). bench.py has the same order.
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass
    try:
        import math_audio_amd as ma
        return ma.device_count()
    except Exception:
        return 0


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")
    config._ma_gpu_count = _gpu_count()      # also fixes the HIP runtime load order (see _gpu_count)


@pytest.fixture(scope="session")
def gpu():
    """Skip (loudly) when no device is visible; GPU tests never fall back to a CPU path."""
    n = _gpu_count()
    if n <= 0:
        pytest.skip("no gfx950 device visible")
    return n
