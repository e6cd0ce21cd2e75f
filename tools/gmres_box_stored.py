"""Config #5 with the system STORED in HBM: 50 172 panels = 40 GB of complex128, a seventh of one MI355X's 288 GB. The matrix
is assembled on the device once (the same kernels as the sweep) and every GMRES iteration is a dense matvec at HBM speed
instead of a re-evaluation of the 13-point rule. usage: python tools/gmres_box_stored.py [scale] [max_iterations]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import math_audio_amd as ma
from math_audio_amd import mesh as mm
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 300
m = mm.generate_box_mesh(0.30, 0.40, 0.60, max(2, int(46 * scale)), max(2, int(61 * scale)), max(2, int(91 * scale)))
n = m.n_elem
k = mm.wave_number(1000.0); beta = mm.burton_miller_beta_scaled(k, 4.0)
dev = torch.device("cuda", 0)
plan = ma.BemPlan(m)
A = torch.empty(n * n, dtype=torch.complex128, device=dev)
rhs0 = torch.empty(n, dtype=torch.complex128, device=dev)
st = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize(); t0 = time.perf_counter()
plan.assemble_dev(k, beta, A.data_ptr(), rhs0.data_ptr(), stream=st)
torch.cuda.synchronize(); t_asm = time.perf_counter() - t0
op = ma.LinearOperator.dense_dev(n, A.data_ptr(), keep=A)
x = torch.ones(n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
op.apply_dev(x.data_ptr(), y.data_ptr(), st); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    op.apply_dev(x.data_ptr(), y.data_ptr(), st)
torch.cuda.synchronize(); t_apply = (time.perf_counter() - t0) / 5
b = ma.incident_rhs(m.center, m.normal, k, beta, kind=1, vec=(0.15, 0.20, 1.0), amp=1.0)
Mp = ma.Preconditioner(op, kind="diagonal")
t0 = time.perf_counter(); xs, info = ma.gmres_preconditioned(op, Mp, b, restart=50, max_iterations=maxit, tol=1e-6); t_gm = time.perf_counter() - t0
res = None
if n <= 20000:
    Ah = A.reshape(n, n).cpu().numpy(); res = float(np.linalg.norm(Ah @ xs - b) / np.linalg.norm(b))
print(json.dumps({"panels": n, "matrix_GB": 16.0 * n * n / 1e9, "assemble_s": t_asm, "assembly_pairs_per_s": n * n / t_asm, "apply_ms": t_apply * 1e3,
                  "apply_GBs": 16.0 * n * n / t_apply / 1e9, "gmres_s": t_gm, "iterations": info.iterations, "restarts": info.restarts,
                  "converged": info.converged, "residual_preconditioned": info.residual, "true_relative_residual": res}))
