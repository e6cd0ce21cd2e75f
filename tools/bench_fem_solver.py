"""Every SolverType of the FEM dispatcher (math_audio_amd/fem_solver.py = math-fem/src/solver/mod.rs) on the F1M family: setup + solve
wall time, iterations, true residual. k = 1.832 + 0.01i, right-hand side of a unit nodal field.
usage: python tools/bench_fem_solver.py [cells_per_side] [tolerance]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
import math_audio_amd as ma
from math_audio_amd import fem_solver as fs
nside = int(sys.argv[1]) if len(sys.argv) > 1 else 48
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
k = 1.832 + 0.01j
p = fs.HelmholtzProblem.box(nside, nside, nside, k)
n = p.num_dofs()
A = sp.csr_matrix((p.stiffness - (k * k) * p.mass, p.col_indices, p.row_ptrs), shape=(n, n))
out = {"dofs": n, "nnz": int(p.row_ptrs[-1]), "k": [k.real, k.imag], "tolerance": tol, "gmres": {"restart": 50, "max_iterations": 400}, "solvers": {}}
skip = {fs.SolverType.Direct} if n > 30000 else set()
for t in fs.SolverType:
    if t in skip:
        continue
    cfg = fs.SolverConfig(solver_type=t, gmres=fs.GmresConfig(400, 50, tol), wavenumber=abs(k))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        s = fs.solve(p, cfg)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out["solvers"][t.name] = {"seconds": dt, "iterations": s.iterations, "converged": s.converged,
                                  "true_residual": float(np.linalg.norm(A @ s.values - p.rhs) / np.linalg.norm(p.rhs))}
    except fs.SolverError as e:
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out["solvers"][t.name] = {"seconds": dt, "error": e.kind, "text": str(e)[:120]}
    print(t.name, out["solvers"][t.name], file=sys.stderr, flush=True)
print(json.dumps(out))
