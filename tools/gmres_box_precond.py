"""Diagnostic: BASELINE.json configs[4] (50 172-panel box, 1 kHz, matrix-free operator) with GMRES(50), plain and with the
diagonal preconditioner (DiagonalPreconditioner::from_diagonal, fmm_interface.rs:177-212). usage: python tools/gmres_box_precond.py [scale]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import math_audio_amd as ma
from math_audio_amd import mesh as mm
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nx, ny, nz = max(2, int(46 * scale)), max(2, int(61 * scale)), max(2, int(91 * scale))
m = mm.generate_box_mesh(0.30, 0.40, 0.60, nx, ny, nz)
k = mm.wave_number(1000.0); beta = mm.burton_miller_beta_scaled(k, 4.0)
plan = ma.BemPlan(m)
op = ma.LinearOperator.tbem(plan, k, beta)
b = ma.incident_rhs(m.center, m.normal, k, beta, kind=1, vec=(0.15, 0.20, 1.0), amp=1.0)
Mp = ma.Preconditioner(op, kind="diagonal")
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 300
t0 = time.perf_counter(); x1, i1 = ma.gmres_preconditioned(op, Mp, b, restart=50, max_iterations=maxit, tol=1e-6); t1 = time.perf_counter() - t0
print(json.dumps({"panels": m.n_elem, "solver": "gmres_preconditioned(diagonal)", "seconds": t1, "iterations": i1.iterations, "restarts": i1.restarts, "converged": i1.converged, "residual": i1.residual}))
