"""Diagnostic (round 4): how far below the 1e-12 of tests/test_sweep_headline_gpu.py the S10 sweep (three systems per pass, staged
pipeline) sits from the single-system path: relative L2 distance per frequency (measured: 4e-16 .. 4e-15). usage: PYTHONPATH=. python tools/r04_sweep_margin.py"""
import numpy as np, torch, sys
sys.path.insert(0,'tests')
import math_audio_amd as ma
from math_audio_amd import mesh as mm
mesh = mm.generate_sphere_mesh(0.1, 51, 100); n = mesh.n_elem
fl = mm.log_space(100.0, 8000.0, 64); idx=[0,7,14,15,23,32,47,56,63]; freqs=[fl[i] for i in idx]
plan = ma.BemPlan(mesh); sw = ma.BemSweep(plan, len(freqs), slots=3)
X, st = sw.run(freqs); sw.close()
dev = torch.device("cuda",0); s0 = torch.cuda.current_stream().cuda_stream
lu = ma.LuPlan(n); A = torch.empty(n*n, dtype=torch.complex128, device=dev); x = torch.empty(n, dtype=torch.complex128, device=dev)
for fi,f in enumerate(freqs):
    k = mm.wave_number(f); beta = mm.burton_miller_beta_scaled(k, 4.0)
    plan.assemble_dev(k, beta, A.data_ptr(), x.data_ptr(), stream=s0); plan.incident_rhs_dev(k, beta, x.data_ptr(), accumulate=True, stream=s0)
    lu.factor_solve_dev(A.data_ptr(), x.data_ptr(), 1, stream=s0); assert lu.status(s0)==0
    x1 = x.cpu().numpy(); print(idx[fi], "rel err %.2e" % (np.linalg.norm(X[fi]-x1)/np.linalg.norm(x1)))
