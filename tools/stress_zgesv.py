"""Diagnostic: many one-shot solves of random sizes (plan create / factor / solve / destroy each time); every residual is checked."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import math_audio_amd as ma
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); worst = 0.0; count = 0
while time.time() - t0 < float(sys.argv[2] if len(sys.argv) > 2 else 40):
    n = int(rng.integers(1, 900))
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)); b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    x = ma.zgesv(A, b)
    res = np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x))
    assert res < 1e-13 * max(n, 10), (n, res)
    worst = max(worst, res / max(n, 10)); count += 1
print("solves", count, "worst residual / n", worst)
