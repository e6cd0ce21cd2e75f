"""Diagnostic (round 4): the big trailing updates of one S10 factorisation (n = 10 000, 64-column panels, kb = 6: K = 384) launched ALONE,
one after the other, on (a) the plan's CU-masked update stream (192 CUs) and (b) an ordinary stream (256 CUs): what the update kernel
needs for a step's worth of updates without any co-tenant, to set beside the 40.3 ms the sweep's events report for them."""
import ctypes as C
import sys
import torch
import math_audio_amd as ma

dev = torch.device("cuda", 0)
lib = ma.lib()
lib.ma_diag_zgemm_dev.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
n, nb, kb = 10000, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 6
K = nb * kb
Q = (n + nb - 1) // nb
G = (Q + kb - 1) // kb
lu = ma.LuPlan(n)
masked = lu.main_stream()
plain = torch.cuda.Stream(device=dev)
A = torch.randn(n * K * 2, dtype=torch.float64, device=dev)
B = torch.randn(K * n * 2, dtype=torch.float64, device=dev)
Cm = torch.zeros(n * n * 2, dtype=torch.float64, device=dev)
shapes = []
for g in range(G):
    e = min(n, (g + 1) * K); enext = min(n, (g + 2) * K)
    if n - e > 0 and n - enext > 0:
        shapes.append((n - e, n - enext))
flops = sum(8.0 * M * N * K for M, N in shapes)
for name, sp in (("masked 192 CUs", masked), ("whole chip", plain.cuda_stream)):
    if not sp:
        print(name, ": the plan has no masked stream"); continue
    ext = torch.cuda.ExternalStream(sp, device=dev)
    for rep in range(2):
        tot = 0.0; rows = []
        for M, N in shapes:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            ma.check(lib.ma_diag_zgemm_dev(M, N, K, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), 1, C.c_void_p(sp)))   # warm
            e0.record(ext)
            ma.check(lib.ma_diag_zgemm_dev(M, N, K, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), 1, C.c_void_p(sp)))
            e1.record(ext); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1); tot += ms; rows.append((M, N, ms, 8.0 * M * N * K / ms / 1e9))
        print("%s, pass %d: %d launches, %.2f ms in all = %.1f TFLOP/s algorithmic over %.3e flop" % (name, rep, len(shapes), tot, flops / tot / 1e9, flops))
    for M, N, ms, tf in rows[::3]:
        print("    M %5d N %5d: %7.3f ms  %5.1f TFLOP/s" % (M, N, ms, tf))
lu.close()
