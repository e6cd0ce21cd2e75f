"""Diagnostic: per-phase time of the LU panel kernel (needs a -DMA_PANEL_STAMPS build of the library)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import math_audio_amd as ma
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
A0 = (torch.randn(n, n, dtype=torch.float64, generator=g) + 1j * torch.randn(n, n, dtype=torch.float64, generator=g)).to(dev)
b0 = torch.ones(n, dtype=torch.complex128, device=dev)
lu = ma.LuPlan(n)
L = ma.lib()
L.ma_lu_plan_panel_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros(8, dtype=np.uint64)
load = sys.argv[2] if len(sys.argv) > 2 else "none"          # "update": the trailing-update kernel runs beside the factorisation on a second stream
side = torch.cuda.Stream()
M = 8192
bA = torch.zeros(M, 256, dtype=torch.complex128, device=dev); bB = torch.zeros(256, M, dtype=torch.complex128, device=dev); bC = torch.zeros(M, M, dtype=torch.complex128, device=dev)
for it in range(2):
    A = A0.clone(); b = b0.clone()
    L.ma_lu_plan_panel_stamps(lu.h, out.ctypes.data_as(C.c_void_p), 1)
    if load == "update":
        ma.check(L.ma_diag_zgemm_dev(M, M, 256, C.c_void_p(bA.data_ptr()), C.c_void_p(bB.data_ptr()), C.c_void_p(bC.data_ptr()), 200, C.c_void_p(side.cuda_stream)))
    lu.set_timing(True)
    lu.factor_solve_dev(A.data_ptr(), b.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
    assert lu.status(torch.cuda.current_stream().cuda_stream) == 0
    t = lu.last_timing()
    L.ma_lu_plan_panel_stamps(lu.h, out.ctypes.data_as(C.c_void_p), 0)
names = ["wait(poll)+barrier", "reduce candidates", "fetch rows+barrier", "swap+multipliers+col c+1", "scan+priority rows", "publish+drain+arrive", "bulk update+barrier", "-"]
tot = out.sum() / 100.0  # us
torch.cuda.synchronize()
print("panel phase totals for workgroup 0 (us), n=%d, columns=%d, background load: %s" % (n, n, load))
for nm, v in zip(names, out):
    print("  %-28s %10.1f us  %6.2f us/col" % (nm, v / 100.0, v / 100.0 / n))
print("  total %.1f us = %.2f us/col; event-timed panel phase %.1f ms" % (tot, tot / n, t[0]))
print("res", float(torch.linalg.norm(A0 @ b - b0) / torch.linalg.norm(b0)))
