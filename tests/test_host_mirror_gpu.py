"""Runs the C++ host-mirror test program (the reference's own seam tests restated in C++)."""
import os
import subprocess
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror(gpu):
    exe = os.path.join(ROOT, "math_audio_amd", "host", "test_host_mirror")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout
