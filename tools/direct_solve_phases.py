import sys, os, time, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import math_audio_amd as ma
from math_audio_amd import mesh as mm
m = mm.generate_box_mesh(0.30, 0.40, 0.60, 46, 61, 91)
n = m.n_elem
k = mm.wave_number(1000.0); beta = mm.burton_miller_beta_scaled(k, 4.0)
dev = torch.device("cuda", 0)
plan = ma.BemPlan(m)
A = torch.empty(n * n, dtype=torch.complex128, device=dev); x = torch.empty(n, dtype=torch.complex128, device=dev)
st = torch.cuda.current_stream().cuda_stream
lu = ma.LuPlan(n)
for timing in (0, 1):
    plan.assemble_dev(k, beta, A.data_ptr(), x.data_ptr(), stream=st)
    plan.incident_rhs_dev(k, beta, x.data_ptr(), kind=1, vec=(0.15, 0.20, 1.0), amp=1.0, accumulate=True, stream=st)
    lu.set_timing(bool(timing))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lu.factor_solve_dev(A.data_ptr(), x.data_ptr(), 1, st)
    rc = lu.status(st); dt = time.perf_counter() - t0
    out = {"timing": timing, "s": dt, "TF": ((8.0 / 3.0) * n ** 3) / dt / 1e12}
    if timing:
        out["phases_ms"] = [float(v) for v in lu.last_timing()]; out["upd"] = lu.last_update_stats()
    print(json.dumps(out), flush=True)
