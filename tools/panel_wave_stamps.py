"""Diagnostic: per-phase time of the batched (wavefront-per-system) panel kernel inside a lock-step batch
(needs a -DMA_PANEL_STAMPS build of the library and MA_LU_BATCH_PANEL=1). usage: panel_wave_stamps.py [n] [nmat]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import math_audio_amd as ma
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nmat = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
A0 = [(torch.randn(n, n, dtype=torch.float64, generator=g) + 1j * torch.randn(n, n, dtype=torch.float64, generator=g)).to(dev) for _ in range(nmat)]
b0 = torch.ones(n, dtype=torch.complex128, device=dev)
lu = ma.LuPlan(n)
L = ma.lib()
L.ma_lu_plan_panel_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros(8, dtype=np.uint64)
st = torch.cuda.current_stream().cuda_stream
for it in range(2):
    A = [a.clone() for a in A0]; b = [b0.clone() for _ in range(nmat)]
    L.ma_lu_plan_panel_stamps(lu.h, out.ctypes.data_as(C.c_void_p), 1)
    lu.set_timing(True)
    lu.factor_solve_batch_dev([a.data_ptr() for a in A], [x.data_ptr() for x in b], 1, st)
    assert lu.status(st) == 0
    t = lu.last_timing()
    L.ma_lu_plan_panel_stamps(lu.h, out.ctypes.data_as(C.c_void_p), 0)
names = ["poll (wait for the round)", "fetch rows + LDS + sync", "interchange + multipliers + pick", "publish (stores, drain, granule)", "leader gather", "bulk rank-1 update", "-", "-"]
tot = out[:6].sum() / 100.0
print("system 0, workgroup 0: phase totals (us), n=%d, batch=%d, MA_LU_BATCH_LDS=%s" % (n, nmat, os.environ.get("MA_LU_BATCH_LDS")))
for nm, v in zip(names[:6], out[:6]):
    print("  %-36s %10.1f us  %6.2f us/col" % (nm, v / 100.0, v / 100.0 / n))
print("  total %.1f us = %.2f us/col; event-timed: panel %.1f ms (all systems), total %.1f ms" % (tot, tot / n, t[0], t[6]))
