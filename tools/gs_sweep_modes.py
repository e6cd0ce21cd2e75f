"""Symmetric Gauss-Seidel sweep and ILU(0) apply on the F1M box under the three schedules: a launch per level (MA_CSR_GS_FLAGS=0), the persistent
launch with a device-wide barrier per level (MA_CSR_GS_PERSISTENT=1), the persistent launch with one flag per row (the default).
usage: python tools/gs_sweep_modes.py [cells_per_side]"""
import sys, os, time, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import math_audio_amd as ma
from math_audio_amd import fem
nside = int(sys.argv[1]) if len(sys.argv) > 1 else 99
nodes, rp, ci, K, M = fem.helmholtz_box(nside, nside, nside)
n = len(rp) - 1
op = ma.CsrOperator(rp, ci, K=K, M=M); op.set_wavenumber(1.832 + 0.01j)
dev = torch.device("cuda", 0)
i = np.arange(n)
x0 = torch.from_numpy(np.sin(0.1 * i) + 1j * np.cos(0.2 * i)).to(dev); b = torch.from_numpy(np.sin(0.2 * i) + 1j * np.cos(0.1 * i)).to(dev)
L = ma.lib(); st = torch.cuda.current_stream().cuda_stream
fw, bw = op.gauss_seidel_levels()
out = {"dofs": n, "levels_forward": fw, "levels_backward": bw}
ref = None
ilu = ma.IluPreconditioner(op)
r = b.clone(); z = torch.empty_like(r)
for name, env in (("launch_per_level", {"MA_CSR_GS_FLAGS": "0"}), ("barrier_per_level", {"MA_CSR_GS_PERSISTENT": "1"}), ("flag_per_row", {})):
    for k_ in ("MA_CSR_GS_PERSISTENT", "MA_CSR_GS_FLAGS"):
        os.environ.pop(k_, None)
    os.environ.update(env)
    x = x0.clone()
    ma.check(L.ma_csr_sym_gauss_seidel_dev(op.h, C.c_void_p(x.data_ptr()), C.c_void_p(b.data_ptr()), 1, C.c_void_p(st))); torch.cuda.synchronize()
    res = x.cpu().numpy()
    if ref is None:
        ref = res
    same = bool((res == ref).all())
    t0 = time.perf_counter()
    for _ in range(10):
        ma.check(L.ma_csr_sym_gauss_seidel_dev(op.h, C.c_void_p(x.data_ptr()), C.c_void_p(b.data_ptr()), 1, C.c_void_p(st)))
    torch.cuda.synchronize(); t_gs = (time.perf_counter() - t0) / 10
    ma.check(L.ma_precond_apply_dev(ilu.h, C.c_void_p(r.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(st))); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        ma.check(L.ma_precond_apply_dev(ilu.h, C.c_void_p(r.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(st)))
    torch.cuda.synchronize(); t_ilu = (time.perf_counter() - t0) / 10
    ma.check(L.ma_csr_status(op.h))
    out[name] = {"sym_gauss_seidel_sweep_ms": t_gs * 1e3, "ilu0_apply_ms": t_ilu * 1e3, "bit_identical_to_the_launches": same}
    print(name, out[name], file=sys.stderr, flush=True)
print(json.dumps(out))
