"""HBM-write-bound assembly: the room-acoustics collocation matrix (16 B per pair, one kernel evaluation per pair)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import math_audio_amd as ma
from math_audio_amd import mesh as mm
import ctypes as C
m = mm.generate_sphere_mesh(0.1, 51, 100); n = m.n_elem
dev = torch.device("cuda", 0)
c = torch.tensor(m.center, device=dev); nr = torch.tensor(m.normal, device=dev); ar = torch.tensor(m.area, device=dev)
A = torch.empty(n * n, dtype=torch.complex128, device=dev)
L = ma.lib(); st = torch.cuda.current_stream().cuda_stream
def run():
    ma.check(L.ma_room_build_matrix_dev(n, C.c_void_p(c.data_ptr()), C.c_void_p(nr.data_ptr()), C.c_void_p(ar.data_ptr()), 18.3, C.c_void_p(A.data_ptr()), C.c_void_p(st)))
run(); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(json.dumps({"kernel": "room_matrix_kernel", "panels": n, "ms": ms, "pairs_per_s": n * n / ms * 1e3, "GB/s_written": 16.0 * n * n / ms / 1e6, "frac_of_8TBs": 16.0 * n * n / ms / 1e6 / 8000.0}))
