"""Config #5's box through the reference's single-level FMM operator (ma_op_create_slfmm): build and apply times on one MI355X.
usage: python tools/bench_slfmm_box.py [scale] [cell_m]   (scale 1.0 = 46 x 61 x 91 cells = 50 172 panels)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import math_audio_amd as ma
from math_audio_amd import mesh as mm
from fmm_clusters import grid_clusters
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
cell = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
m = mm.generate_box_mesh(0.30, 0.40, 0.60, max(2, int(46 * scale)), max(2, int(61 * scale)), max(2, int(91 * scale)))
n = m.n_elem
k = mm.wave_number(1000.0)
t0 = time.perf_counter(); cl = grid_clusters(m.center, cell); t_cl = time.perf_counter() - t0
sizes = np.diff(cl.elem_ptr)
plan = ma.BemPlan(m)
torch.cuda.synchronize(); t0 = time.perf_counter()
op = ma.LinearOperator.slfmm(plan, cl, k, 8, 16, 6)
torch.cuda.synchronize(); t_build = time.perf_counter() - t0
dev = torch.device("cuda", 0)
x = torch.ones(n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
op.apply_dev(x.data_ptr(), y.data_ptr(), st); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    op.apply_dev(x.data_ptr(), y.data_ptr(), st)
torch.cuda.synchronize(); t_apply = (time.perf_counter() - t0) / 10
near_entries = int((sizes[:, None] * 0).sum()) if False else None
nb = 0
for c in range(cl.n):
    nb += int(sizes[c]) ** 2 + sum(int(sizes[c]) * int(sizes[j]) for j in cl.near_idx[cl.near_ptr[c]:cl.near_ptr[c + 1]] if j > c)
print(json.dumps({"panels": n, "clusters": cl.n, "elements_per_cluster_mean": float(sizes.mean()), "elements_per_cluster_max": int(sizes.max()),
                  "near_entries": nb, "near_GB": nb * 16 / 1e9, "far_pairs": int(cl.far_ptr[-1]), "sphere_points": 128,
                  "cluster_build_host_s": t_cl, "operator_build_s": t_build, "apply_ms": t_apply * 1e3,
                  "apply_near_GBs": nb * 16 / t_apply / 1e9, "finite": bool(torch.isfinite(torch.view_as_real(y)).all())}))
