"""Time per iteration of bicgstab / cgs / gmres over a dense device operator (the step scalars of the first two stay on the host where
the reference branches: one host round trip per inner product). usage: python tools/krylov_step_time.py"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math_audio_amd as ma
rng = np.random.default_rng(1)
for n in (1280, 5000):
    A = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) * (0.3 / np.sqrt(n)) + np.eye(n) * (1.0 + 0.2j)
    b = np.ones(n, dtype=complex)
    op = ma.LinearOperator.dense(A)
    for name, fn in (("bicgstab", ma.bicgstab), ("cgs", ma.cgs), ("gmres", lambda o, bb, it, tol: ma.gmres(o, bb, restart=50, max_iterations=it, tol=tol))):
        fn(op, b, 5, 1e-30)
        t0 = time.perf_counter(); x, info = fn(op, b, 200, 1e-300); dt = time.perf_counter() - t0
        print(n, name, info.iterations, "%.1f us/iter" % (dt / max(info.iterations, 1) * 1e6))
    op.close()
