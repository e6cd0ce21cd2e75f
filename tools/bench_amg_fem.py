"""The FEM solve of config #4's family with the AMG V-cycle on the device: box 5 x 4 x 2.5 m, n^3 cells (n a multiple of 8), P1 Kuhn
tets, shifted operator K + (0.3^2 M ...) below the first mode so that the cycle contracts; V-cycle apply time, GMRES + AMG.
usage: python tools/bench_amg_fem.py [cells_per_edge=96] [levels=4]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, scipy.sparse as sp, torch
import math_audio_amd as ma
from math_audio_amd import fem
from amg_hierarchy import box_hierarchy, csr_triplet
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 96
nlev = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t0 = time.perf_counter()
nodes, rp, ci, K, M = fem.helmholtz_box(nc, nc, nc)
n = len(rp) - 1
k = complex(0.3, 0.01)
A = (sp.csr_matrix((K - (k * k) * M, ci, rp), shape=(n, n)) + 0.05 * sp.identity(n)).tocsr()
levels = box_hierarchy(A, nc, nc, nc, nlev)
t_host = time.perf_counter() - t0
dl = []
for lv in levels:
    a = csr_triplet(lv["A"]); d = {"A": ma.CsrOperator(a[0], a[1], values=a[2])}
    if "P" in lv:
        pt = csr_triplet(lv["P"]); rt = csr_triplet(lv["R"])
        d["P"] = ma.CsrOperator.rect(lv["P"].shape[0], lv["P"].shape[1], *pt); d["R"] = ma.CsrOperator.rect(lv["R"].shape[0], lv["R"].shape[1], *rt)
    dl.append(d)
out = {"dofs": [int(l["A"].shape[0]) for l in levels], "nnz": [int(l["A"].nnz) for l in levels], "host_hierarchy_s": t_host}
dev = torch.device("cuda", 0)
r = torch.ones(n, dtype=torch.complex128, device=dev); z = torch.empty_like(r)
st = torch.cuda.current_stream().cuda_stream
L = ma.lib()
import ctypes as C
for name, kw in (("jacobi_2_2", dict(smoother="jacobi", jacobi_weight=0.8, num_pre_smooth=2, num_post_smooth=2)),
                 ("l1_1_1", dict(smoother="l1", num_pre_smooth=1, num_post_smooth=1)),
                 ("sgs_1_1", dict(smoother="sgs", num_pre_smooth=1, num_post_smooth=1))):
    Mp = ma.AmgPreconditioner(dl, **kw)
    ma.check(L.ma_precond_apply_dev(Mp.h, C.c_void_p(r.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(st))); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ma.check(L.ma_precond_apply_dev(Mp.h, C.c_void_p(r.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(st)))
    torch.cuda.synchronize()
    out["vcycle_ms_" + name] = (time.perf_counter() - t0) / 5 * 1e3
    if name == "jacobi_2_2":
        i = np.arange(n); xt = np.sin(0.1 * i) + 1j * np.cos(0.2 * i); b = A @ xt
        op = ma.LinearOperator.csr(dl[0]["A"])
        t0 = time.perf_counter(); xg, info = ma.gmres_preconditioned(op, Mp, b, restart=30, max_iterations=10, tol=1e-8); tg = time.perf_counter() - t0
        out["gmres_amg"] = {"seconds": tg, "iterations": info.iterations, "converged": info.converged, "true_residual": float(np.linalg.norm(A @ xg - b) / np.linalg.norm(b))}
        t0 = time.perf_counter(); xp, ip = ma.gmres(op, b, restart=30, max_iterations=10, tol=1e-8); tp = time.perf_counter() - t0
        out["gmres_plain"] = {"seconds": tp, "iterations": ip.iterations, "converged": ip.converged}
        op.close()
    Mp.close()
print(json.dumps(out))
