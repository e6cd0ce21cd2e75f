import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import math_audio_amd as ma
import ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 129
rng = np.random.default_rng(1)
A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)); b = rng.standard_normal(n) + 0j
def zgesv(A, b):
    Af = np.ascontiguousarray(A.copy()); bf = b.copy(); ip = np.zeros(n, dtype=np.int32)
    rc = ma.lib().ma_zgesv(n, Af.ctypes.data_as(C.c_void_p), bf.ctypes.data_as(C.c_void_p), ip.ctypes.data_as(C.c_void_p))
    return rc, Af, bf, ip
os.environ["MA_LU_REG_PANEL"] = "0"
rc0, F0, x0, ip0 = zgesv(A, b)
os.environ["MA_LU_REG_PANEL"] = "2"
rc, F, x, ip = zgesv(A, b)
print("n", n, "residual old", np.linalg.norm(A @ x0 - b) / np.linalg.norm(b), "pair", np.linalg.norm(A @ x - b) / np.linalg.norm(b))
print("pivots equal", np.array_equal(ip, ip0), np.nonzero(ip != ip0)[0][:8])
D = np.abs(F - F0)
rows = np.nonzero(D.max(axis=1) > 1e-9)[0]; cols = np.nonzero(D.max(axis=0) > 1e-9)[0]
print("factor diff max", D.max(), "bad rows", rows[:12], len(rows), "bad cols", cols[:12], len(cols))
print("x diff", np.abs(x - x0).max())
