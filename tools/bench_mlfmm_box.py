"""Config #5's box through the reference's multi-level FMM operator (ma_cluster_tree_build + ma_op_create_mlfmm): tree, build and
apply times on one MI355X.
usage: python tools/bench_mlfmm_box.py [scale] [target_elements_per_leaf] [frequency_hz]   (scale 1.0 = 50 172 panels)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import math_audio_amd as ma
from math_audio_amd import mesh as mm
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
target = int(sys.argv[2]) if len(sys.argv) > 2 else 64
freq = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0
m = mm.generate_box_mesh(0.30, 0.40, 0.60, max(2, int(46 * scale)), max(2, int(61 * scale)), max(2, int(91 * scale)))
n = m.n_elem
k = mm.wave_number(freq)
t0 = time.perf_counter(); tree = ma.ClusterTree(m, target, k); t_tree = time.perf_counter() - t0
levels = []
for l in range(tree.num_levels()):
    lv = tree.level(l)
    levels.append({"clusters": lv["n_clusters"], "theta": lv["theta_points"], "far_pairs": int(lv["far_ptr"][-1]), "elements_listed": int(lv["elem_ptr"][-1])})
leaf = tree.level(tree.num_levels() - 1)
sizes = np.diff(leaf["elem_ptr"])
nb = 0
for c in range(leaf["n_clusters"]):
    nb += int(sizes[c]) ** 2 + sum(int(sizes[c]) * int(sizes[j]) for j in leaf["near_idx"][leaf["near_ptr"][c]:leaf["near_ptr"][c + 1]] if j > c)
out = {"panels": n, "frequency_hz": freq, "target_elements_per_leaf": target, "tree_host_s": t_tree, "levels": levels, "near_entries": nb, "near_GB": nb * 16 / 1e9}
plan = ma.BemPlan(m)
try:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    op = ma.LinearOperator.mlfmm(plan, tree, k)
    torch.cuda.synchronize(); out["operator_build_s"] = time.perf_counter() - t0
    dev = torch.device("cuda", 0)
    x = torch.ones(n, dtype=torch.complex128, device=dev); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    op.apply_dev(x.data_ptr(), y.data_ptr(), st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        op.apply_dev(x.data_ptr(), y.data_ptr(), st)
    torch.cuda.synchronize(); t_apply = (time.perf_counter() - t0) / 10
    out.update({"apply_ms": t_apply * 1e3, "apply_near_GBs": nb * 16 / t_apply / 1e9, "finite": bool(torch.isfinite(torch.view_as_real(y)).all())})
except ma.MaError as e:
    out["refused"] = str(e)
print(json.dumps(out))
