"""Diagnostic: factor + solve of one large random complex system through LuPlan only (works with older builds of the library via
MA_LIB_PATH). usage: python tools/lu_big_random.py [n]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import math_audio_amd as ma
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50172
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
A = torch.empty(n * n, dtype=torch.complex128, device=dev)
Av = torch.view_as_real(A)
step = 1 << 28
for i in range(0, Av.shape[0], step):
    Av[i:i + step].copy_(torch.randn(Av[i:i + step].shape, dtype=torch.float64, device=dev, generator=g))
x = torch.ones(n, dtype=torch.complex128, device=dev)
st = torch.cuda.current_stream().cuda_stream
lu = ma.LuPlan(n)
lu.set_timing(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
lu.factor_solve_dev(A.data_ptr(), x.data_ptr(), 1, st)
rc = lu.status(st); dt = time.perf_counter() - t0
print(json.dumps({"n": n, "status": rc, "s": dt, "TF": ((8.0 / 3.0) * n ** 3) / dt / 1e12, "phases_ms": [float(v) for v in lu.last_timing()], "lib": os.environ.get("MA_LIB_PATH", "HEAD")}), flush=True)
