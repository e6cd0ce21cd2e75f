import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def _gpu_count():
    try:
        import math_audio_amd as ma
        return ma.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    """Skip (loudly) when no device is visible; GPU tests never fall back to a CPU path."""
    n = _gpu_count()
    if n <= 0:
        pytest.skip("no gfx950 device visible")
    return n
