"""Diagnostic: what slows the panel kernels when trailing updates run beside them? One 10k system is factored (its own
look-ahead updates included) while a second stream carries a background load: nothing, the matrix-core probe (MFMA issue only,
no memory traffic), a device-to-device copy loop (memory traffic only), or the trailing-update kernel itself.
Prints the summed panel-kernel time (phase 0) and the whole factorisation per case.
usage: python tools/panel_cotenancy.py [n]"""
import sys, os, json, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import math_audio_amd as ma

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu"); g.manual_seed(1)
A0 = (torch.randn(n, n, dtype=torch.float64, generator=g) + 1j * torch.randn(n, n, dtype=torch.float64, generator=g)).to(dev)
b0 = torch.ones(n, dtype=torch.complex128, device=dev)
lu = ma.LuPlan(n); lu.set_timing(True)
side = torch.cuda.Stream()
L = ma.lib()
M = 4096
bA = torch.randn(M, 256, dtype=torch.complex128, device=dev); bB = torch.randn(256, M, dtype=torch.complex128, device=dev) * 1e-3
bC = torch.zeros(M, M, dtype=torch.complex128, device=dev)
burn = torch.zeros(256 * 512, dtype=torch.float64, device=dev)
src = torch.empty(1 << 28, dtype=torch.float64, device=dev); dst = torch.empty_like(src)       # 2 GiB each


def background(kind):
    s = side.cuda_stream
    if kind == "mfma_only":
        ma.check(L.ma_diag_mfma_burn(C.c_void_p(burn.data_ptr()), 512, 20000, 40, C.c_void_p(s)))
    elif kind == "copy_only":
        with torch.cuda.stream(side):
            for _ in range(60):
                dst.copy_(src)
    elif kind == "update_kernel":
        ma.check(L.ma_diag_zgemm_dev(M, M, 256, C.c_void_p(bA.data_ptr()), C.c_void_p(bB.data_ptr()), C.c_void_p(bC.data_ptr()), 400, C.c_void_p(s)))
    elif kind == "update_kernel_small":        # K = 64, the look-ahead lanes' shape
        ma.check(L.ma_diag_zgemm_dev(M, M, 64, C.c_void_p(bA.data_ptr()), C.c_void_p(bB.data_ptr()), C.c_void_p(bC.data_ptr()), 1200, C.c_void_p(s)))


out = {}
for kind in ("none", "none", "mfma_only", "copy_only", "update_kernel", "update_kernel_small", "none"):
    A = A0.clone(); b = b0.clone()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        t0.record()
    background(kind)
    with torch.cuda.stream(side):
        t2.record()
    st = torch.cuda.current_stream().cuda_stream
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    lu.factor_solve_dev(A.data_ptr(), b.data_ptr(), 1, stream=st)
    e1.record()
    e1.synchronize()
    bg_still_running = not t2.query()
    torch.cuda.synchronize()
    tm = lu.last_timing()
    rec = {"factor_solve_ms": e0.elapsed_time(e1), "panel_ms": float(tm[0]), "phases_ms": [float(v) for v in tm], "background_ms": t0.elapsed_time(t2),
           "background_outlasted_the_solve": bool(bg_still_running), "status": lu.status()}
    out.setdefault(kind, []).append(rec)
    print(kind, json.dumps(rec), flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "panel_cotenancy.json"), "w"), indent=1)
