"""One configuration of the update kernel for a counter pass: python tools/zgemm_one.py M N K reps"""
import ctypes as C
import sys
import torch
import math_audio_amd as ma

M, N, K, reps = (int(v) for v in sys.argv[1:5])
dev = torch.device("cuda", 0)
lib = ma.lib()
A = torch.randn(M * K, dtype=torch.complex128, device=dev)
B = torch.randn(K * N, dtype=torch.complex128, device=dev)
Cm = torch.zeros(M * N, dtype=torch.complex128, device=dev)
ma.check(lib.ma_diag_zgemm_dev(M, N, K, C.c_void_p(A.data_ptr()), C.c_void_p(B.data_ptr()), C.c_void_p(Cm.data_ptr()), reps, C.c_void_p(0)))
torch.cuda.synchronize()
