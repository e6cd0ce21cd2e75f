"""Round 5: where does partial pivoting find its pivots on the S10 systems? (which half-panels the speculative panel must reject, and by how much)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.is_available()
import math_audio_amd as ma
from math_audio_amd import mesh as mm

mesh = mm.generate_sphere_mesh(0.1, 51, 100)
n = mesh.n_elem
fl = mm.log_space(100.0, 8000.0, 64)
os.environ["MA_LU_SPECULATE"] = "0"
for fi in (0, 20, 40, 63):
    k = mm.wave_number(fl[fi]); beta = mm.burton_miller_beta_scaled(k, 4.0)
    A, r0 = ma.assemble_tbem(mesh, k, beta)
    b = r0 + ma.incident_rhs(mesh.center, mesh.normal, k, beta)
    x, piv = ma.zgesv(A, b, return_pivots=True)
    # replay the interchanges: which ORIGINAL row served as pivot of column c, and where it sat when chosen
    cols = np.arange(n)
    moved = np.nonzero(piv != cols)[0]
    dist = piv[moved] - moved
    print("f[%d] = %.1f Hz: %d of %d columns interchange; distance: median %d, max %d; beyond 31 rows: %d" % (fi, fl[fi], len(moved), n, np.median(dist) if len(moved) else 0, dist.max() if len(moved) else 0, (dist > 31).sum()))
    hp = {}
    for c in moved:
        h = c // 32
        far = piv[c] >= (h + 1) * 32
        if far:
            hp.setdefault(h, []).append(int(piv[c]))
    print("   half-panels with a pivot below their top block: %d; rows involved per such half-panel: %s" % (len(hp), sorted(len(set(v)) for v in hp.values())))
    print("   first such half-panels: %s" % [(h, sorted(set(v))[:6]) for h, v in sorted(hp.items())[:6]])
