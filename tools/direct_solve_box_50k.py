"""Config #5's system solved DIRECTLY on one MI355X: 50 172 panels = 40 GB, (8/3) n^3 = 3.4e14 flop. The matrix is assembled on
the device, a copy is kept to check the residual (80 GB of the 288 GB). usage: python tools/direct_solve_box_50k.py [scale]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import math_audio_amd as ma
from math_audio_amd import mesh as mm
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
m = mm.generate_box_mesh(0.30, 0.40, 0.60, max(2, int(46 * scale)), max(2, int(61 * scale)), max(2, int(91 * scale)))
n = m.n_elem
k = mm.wave_number(1000.0); beta = mm.burton_miller_beta_scaled(k, 4.0)
dev = torch.device("cuda", 0)
plan = ma.BemPlan(m)
A = torch.empty(n * n, dtype=torch.complex128, device=dev); x = torch.empty(n, dtype=torch.complex128, device=dev)
st = torch.cuda.current_stream().cuda_stream
plan.assemble_dev(k, beta, A.data_ptr(), x.data_ptr(), stream=st)
plan.incident_rhs_dev(k, beta, x.data_ptr(), kind=1, vec=(0.15, 0.20, 1.0), amp=1.0, accumulate=True, stream=st)
A0 = A.clone(); b = x.clone()
lu = ma.LuPlan(n)
torch.cuda.synchronize(); t0 = time.perf_counter()
lu.factor_solve_dev(A.data_ptr(), x.data_ptr(), 1, st)
rc = lu.status(st); dt = time.perf_counter() - t0
op = ma.LinearOperator.dense_dev(n, A0.data_ptr(), keep=A0)
y = torch.empty_like(x); op.apply_dev(x.data_ptr(), y.data_ptr(), st); torch.cuda.synchronize()
res = float((y - b).norm() / b.norm())
flops = (8.0 / 3.0) * n ** 3 + 8.0 * n * n
print(json.dumps({"panels": n, "status": rc, "factor_solve_s": dt, "TFLOPs": flops / dt / 1e12, "relative_residual": res}))
