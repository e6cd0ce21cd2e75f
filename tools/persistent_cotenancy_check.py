"""The persistent kernels (value-as-flag sweeps, one-launch Gram-Schmidt steps) beside a chip-filling background load on another stream:
GMRES + ILU(0) on the F1M family while the trailing-update kernel of the LU runs back to back. Same iteration count and solution as
alone, no abandoned wait. usage: python tools/persistent_cotenancy_check.py [cells_per_side]"""
import sys, os, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import math_audio_amd as ma
from math_audio_amd import fem
nside = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nodes, rp, ci, K, M = fem.helmholtz_box(nside, nside, nside)
n = len(rp) - 1
op = ma.CsrOperator(rp, ci, K=K, M=M); op.set_wavenumber(1.832 + 0.01j)
lin = ma.LinearOperator.csr(op); ilu = ma.IluPreconditioner(op)
i = np.arange(n); b = op.matvec(np.sin(0.1 * i) + 1j * np.cos(0.2 * i))
dev = torch.device("cuda", 0); L = ma.lib()
Mz = 4096
bA = torch.randn(Mz, 256, dtype=torch.complex128, device=dev); bB = torch.randn(256, Mz, dtype=torch.complex128, device=dev) * 1e-3; bC = torch.zeros(Mz, Mz, dtype=torch.complex128, device=dev)
side = torch.cuda.Stream()
out = {"dofs": n}
for name, load in (("alone", False), ("beside_trailing_updates", True), ("alone_again", False)):
    if load:
        ma.check(L.ma_diag_zgemm_dev(Mz, Mz, 256, C.c_void_p(bA.data_ptr()), C.c_void_p(bB.data_ptr()), C.c_void_p(bC.data_ptr()), 4000, C.c_void_p(side.cuda_stream)))
    t0 = time.perf_counter()
    x, info = ma.gmres_preconditioned(lin, ilu, b, restart=50, max_iterations=20, tol=1e-9)
    dt = time.perf_counter() - t0
    still = not side.query() if load else False
    torch.cuda.synchronize()
    ma.check(L.ma_csr_status(op.h))
    out[name] = {"seconds": dt, "iterations": info.iterations, "converged": bool(info.converged), "background_outlasted_the_solve": bool(still), "x_checksum": [float(x.real.sum()), float(x.imag.sum())]}
    print(name, out[name], file=sys.stderr, flush=True)
print(json.dumps(out))
