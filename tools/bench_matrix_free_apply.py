"""Diagnostic: time of y = A x and y = A^T x of the matrix-free TBEM operator on the 50 172-panel box (config #5)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import math_audio_amd as ma
from math_audio_amd import mesh as mm
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
m = mm.generate_box_mesh(0.30, 0.40, 0.60, max(2, int(46 * scale)), max(2, int(61 * scale)), max(2, int(91 * scale)))
n = m.n_elem
k = mm.wave_number(1000.0); beta = mm.burton_miller_beta_scaled(k, 4.0)
plan = ma.BemPlan(m); op = ma.LinearOperator.tbem(plan, k, beta)
dev = torch.device("cuda", 0)
x = torch.ones(n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
out = {"panels": n}
for name, fn in (("apply", op.apply_dev), ("apply_transpose", op.apply_transpose_dev)):
    fn(x.data_ptr(), y.data_ptr(), st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn(x.data_ptr(), y.data_ptr(), st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    out[name] = {"ms": dt * 1e3, "pairs_per_s": n * n / dt}
print(json.dumps(out))
