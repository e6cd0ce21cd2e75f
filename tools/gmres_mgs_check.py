"""GMRES with the Gram-Schmidt step as one launch (default) against an inner-product and an update kernel per basis vector
(MA_GMRES_FUSED_MGS=0): time per iteration and the difference of the iterates. usage: python tools/gmres_mgs_check.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math_audio_amd as ma
from math_audio_amd import fem
nodes, rp, ci, K, M = fem.helmholtz_box(32, 32, 32)
n = len(rp) - 1
op = ma.CsrOperator(rp, ci, K=K, M=M); op.set_wavenumber(1.832 + 0.01j)
lin = ma.LinearOperator.csr(op)
i = np.arange(n); b = op.matvec(np.sin(0.1 * i) + 1j * np.cos(0.2 * i))
res = {}
for mode in ("1", "0"):
    os.environ["MA_GMRES_FUSED_MGS"] = mode
    t0 = time.perf_counter(); x, info = ma.gmres(lin, b, restart=50, max_iterations=40, tol=1e-10); dt = time.perf_counter() - t0
    res[mode] = (x, info.iterations, dt)
    print("fused" if mode == "1" else "separate", info.iterations, info.converged, "%.3f s" % dt, "%.3f ms/iter" % (dt / max(info.iterations, 1) * 1e3))
print("same iteration count:", res["1"][1] == res["0"][1], " max relative difference of the iterates: %.2e" % (np.abs(res["1"][0] - res["0"][0]).max() / np.abs(res["0"][0]).max()))
