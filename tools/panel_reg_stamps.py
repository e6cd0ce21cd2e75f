"""Diagnostic: per-phase time of lu_panel_reg_kernel's workgroup 0 (needs the -DMA_PANEL_STAMPS build: make -C math_audio_amd/csrc stamps;
MA_LIB_PATH=math_audio_amd/lib/libmathaudio_hip_stamps.so MA_LU_REG_PANEL=1 MA_LU_LOOKAHEAD=0 python tools/panel_reg_stamps.py [n])"""
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
st = torch.cuda.current_stream().cuda_stream
for it in range(2):
    A = A0.clone(); b = b0.clone()
    L.ma_lu_plan_panel_stamps(lu.h, out.ctypes.data_as(C.c_void_p), 1)
    lu.set_timing(True)
    lu.factor_solve_dev(A.data_ptr(), b.data_ptr(), 1, st)
    assert lu.status(st) == 0
    t = lu.last_timing()
    L.ma_lu_plan_panel_stamps(lu.h, out.ctypes.data_as(C.c_void_p), 0)
v = [float(x) for x in out]
cols = max(v[6], 1.0); snd = max(v[1], 1.0)
print("register panel kernel, all workgroups (lane 0 of the acting wavefront), n=%d: %d workgroup-columns" % (n, int(cols)))
print("  sender: B1 -> row stores issued     %6.3f us" % (v[7] / snd / 100.0))
print("  sender: B1 -> granule stored        %6.3f us" % (v[0] / snd / 100.0))
print("  wave 0: B1 -> sweep starts          %6.3f us" % (v[2] / cols / 100.0))
print("  wave 0: sweep until all tagged      %6.3f us  (%.2f sweeps per column)" % (v[3] / cols / 100.0, v[4] / cols))
print("  wave 0: B2 -> B1 of the next column %6.3f us" % (v[5] / cols / 100.0))
print("  event-timed panel phase %.1f ms = %.2f us per column" % (t[0], t[0] * 1e3 / n))
print("res", float(torch.linalg.norm(A0 @ b - b0) / torch.linalg.norm(b0)))
