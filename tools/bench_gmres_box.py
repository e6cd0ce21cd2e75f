"""BASELINE.json configs[4] on ONE GPU: closed box 0.30 x 0.40 x 0.60 m, 46 x 61 x 91 cells -> 50 172 Tri3,
f = 1 kHz, monopole at (0.15, 0.20, 1.0), GMRES(50) tol 1e-6 with the matrix-free TBEM operator."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math_audio_amd as ma
from math_audio_amd import mesh as mm
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nx, ny, nz = max(2, int(46 * scale)), max(2, int(61 * scale)), max(2, int(91 * scale))
t0 = time.perf_counter(); m = mm.generate_box_mesh(0.30, 0.40, 0.60, nx, ny, nz); t_mesh = time.perf_counter() - t0
n = m.n_elem
k = mm.wave_number(1000.0); beta = mm.burton_miller_beta_scaled(k, 4.0)
t0 = time.perf_counter(); plan = ma.BemPlan(m); t_plan = time.perf_counter() - t0
t0 = time.perf_counter(); op = ma.LinearOperator.tbem(plan, k, beta); t_op = time.perf_counter() - t0
b = ma.incident_rhs(m.center, m.normal, k, beta, kind=1, vec=(0.15, 0.20, 1.0), amp=1.0)
x = np.ones(n, dtype=complex)
op.apply(x)
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    op.apply(x)
t_apply = (time.perf_counter() - t0) / reps
t0 = time.perf_counter(); xs, info = ma.gmres(op, b, restart=50, max_iterations=20, tol=1e-6); t_gm = time.perf_counter() - t0
print(json.dumps({"panels": n, "near_pairs": plan.num_near_pairs, "mesh_s": t_mesh, "plan_s": t_plan, "operator_setup_s": t_op,
                  "apply_s": t_apply, "pairs_per_s": n * n / t_apply, "gmres_s": t_gm, "iterations": info.iterations, "restarts": info.restarts,
                  "converged": info.converged, "residual": info.residual}))
