"""AmgPreconditioner::from_csr on the F1M family (BASELINE configs[3]: 96^3 cells, 912 673 DoF, k = 1.832 + 0.01i): host setup time
inside the library, hierarchy sizes, one cycle, and GMRES with it against plain GMRES.
usage: python tools/bench_amg_from_csr.py [cells_per_side] [preset]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import math_audio_amd as ma
from math_audio_amd import fem
nside = int(sys.argv[1]) if len(sys.argv) > 1 else 96
preset = sys.argv[2] if len(sys.argv) > 2 else "for_parallel"
t0 = time.perf_counter(); nodes, rp, ci, K, M = fem.helmholtz_box(nside, nside, nside); t_asm = time.perf_counter() - t0
n = len(rp) - 1
k = 1.832 + 0.01j
op = ma.CsrOperator(rp, ci, K=K, M=M); op.set_wavenumber(k)
lin = ma.LinearOperator.csr(op)
out = {"dofs": n, "nnz": int(rp[-1]), "preset": preset, "host_assembly_s": t_asm}
t0 = time.perf_counter(); amg = ma.AmgFromCsr(op, ma.AmgConfig.preset(preset)); out["from_csr_wall_s"] = time.perf_counter() - t0
d = amg.diagnostics(); out.update(d)
i = np.arange(n); xs = np.sin(0.1 * i) + 1j * np.cos(0.2 * i)
b = op.matvec(xs)
dev = torch.device("cuda", 0)
r = torch.from_numpy(b).to(dev); z = torch.empty_like(r)
L = ma.lib(); import ctypes as C
st = torch.cuda.current_stream().cuda_stream
ma.check(L.ma_precond_apply_dev(amg.h, C.c_void_p(r.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(st))); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ma.check(L.ma_precond_apply_dev(amg.h, C.c_void_p(r.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(st)))
torch.cuda.synchronize(); out["cycle_ms"] = (time.perf_counter() - t0) / 20 * 1e3
t0 = time.perf_counter(); x, info = ma.gmres_preconditioned(lin, amg, b, restart=30, max_iterations=300, tol=1e-8); t1 = time.perf_counter() - t0
out["gmres_amg"] = {"iterations": info.iterations, "converged": bool(info.converged), "seconds": t1, "error_vs_known_solution": float(np.abs(x - xs).max() / np.abs(xs).max())}
t0 = time.perf_counter(); x0, i0 = ma.gmres(lin, b, restart=30, max_iterations=300, tol=1e-8); t2 = time.perf_counter() - t0
out["gmres_plain"] = {"iterations": i0.iterations, "converged": bool(i0.converged), "seconds": t2}
print(json.dumps(out))
