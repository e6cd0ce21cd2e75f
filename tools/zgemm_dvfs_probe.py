"""Diagnostic: is the trailing-update kernel held back by its own cycles or by the clock the chip holds under load? The same
launch on random, constant and zero operands (MI355X_MICROARCH.md, 'DVFS give-back': the power of an MFMA loop depends on the data)."""
import ctypes as C
import sys
import torch
import math_audio_amd as ma

dev = torch.device("cuda", 0)
lib = ma.lib()
M = N = 9984
for K in (256, 1024):
    for kind in ("random", "ones", "zeros"):
        mk = {"random": lambda n: torch.randn(n, dtype=torch.complex128, device=dev),
              "ones": lambda n: torch.ones(n, dtype=torch.complex128, device=dev),
              "zeros": lambda n: torch.zeros(n, dtype=torch.complex128, device=dev)}[kind]
        A = mk(M * K); B = mk(K * N); Cm = torch.zeros(M * N, dtype=torch.complex128, device=dev)
        rep = max(3, int(6e12 / (8.0 * M * N * K)))
        ma.check(lib.ma_diag_zgemm_dev(M, N, K, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), 2, C.c_void_p(0)))
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        ma.check(lib.ma_diag_zgemm_dev(M, N, K, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), rep, C.c_void_p(0)))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / rep
        tf = 8.0 * M * N * K / (ms * 1e-3) / 1e12
        print("K=%5d %-7s %8.3f ms  %6.1f TFLOP/s complex-equivalent  %5.1f real on the matrix cores" % (K, kind, ms, tf, 0.75 * tf))
        sys.stdout.flush()
        del A, B, Cm
