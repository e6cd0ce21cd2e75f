"""Diagnostic: the trailing-update kernel (zgemm3m_sub_kernel via ma_diag_zgemm_dev) alone on an idle chip, over K and size:
where does it leave the matrix-core peak -- in the main loop (flat in K) or in the per-tile prologue / C epilogue (rises with K)?"""
import ctypes as C
import sys
import torch
import math_audio_amd as ma

dev = torch.device("cuda", 0)
lib = ma.lib()
PEAK = 78.6
for (M, N) in ((9984, 9984), (4992, 4992)):
    for K in (64, 128, 256, 384, 512, 1024, 2048):
        A = torch.randn(M * K, dtype=torch.complex128, device=dev)
        B = torch.randn(K * N, dtype=torch.complex128, device=dev)
        Cm = torch.zeros(M * N, dtype=torch.complex128, device=dev)
        rep = max(3, int(3e12 / (8.0 * M * N * K)))
        ma.check(lib.ma_diag_zgemm_dev(M, N, K, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), 2, C.c_void_p(0)))
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ma.check(lib.ma_diag_zgemm_dev(M, N, K, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), rep, C.c_void_p(0)))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / rep
        tf = 8.0 * M * N * K / (ms * 1e-3) / 1e12
        print("M=N=%5d K=%5d  %8.3f ms  %6.1f TFLOP/s complex-equivalent  %5.1f real on the matrix cores = %.2f of peak" % (M, K, ms, tf, 0.75 * tf, 0.75 * tf / PEAK))
        sys.stdout.flush()
        del A, B, Cm
