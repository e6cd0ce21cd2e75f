"""BASELINE.json configs[4] with the rows of the matrix-free operator sharded over the GPUs of one node:
closed box 0.30 x 0.40 x 0.60 m, 46 x 61 x 91 cells -> 50 172 Tri3, f = 1 kHz, monopole at (0.15, 0.20, 1.0),
GMRES(50), tol 1e-6. One all-gather of y per apply (math_audio_amd/sharded.py).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/bench_gmres_box_sharded.py [scale]
(also runs as a single process: world size 1)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl", device_id=dev)
import math_audio_amd as ma
from math_audio_amd import mesh as mm, sharded
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nx, ny, nz = max(2, int(46 * scale)), max(2, int(61 * scale)), max(2, int(91 * scale))
m = mm.generate_box_mesh(0.30, 0.40, 0.60, nx, ny, nz)
n = m.n_elem
k = mm.wave_number(1000.0); beta = mm.burton_miller_beta_scaled(k, 4.0)
plan = ma.BemPlan(m, device=local)
t0 = time.perf_counter(); op = sharded.tbem_sharded_operator(plan, k, beta, dist=dist if world > 1 else None, device=dev); t_op = time.perf_counter() - t0
b = torch.tensor(ma.incident_rhs(m.center, m.normal, k, beta, kind=1, vec=(0.15, 0.20, 1.0), amp=1.0), device=dev)
x = torch.ones(n, dtype=torch.complex128, device=dev)
op.apply(x); torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    op.apply(x)
torch.cuda.synchronize(); t_apply = (time.perf_counter() - t0) / reps
t0 = time.perf_counter(); xs, info = sharded.gmres(op, b, restart=50, max_iterations=20, tol=1e-6); torch.cuda.synchronize(); t_gm = time.perf_counter() - t0
if rank == 0:
    print(json.dumps({"panels": n, "n_gpus": world, "rows_per_gpu": op.per, "operator_setup_s": t_op, "apply_s": t_apply, "pairs_per_s": n * n / t_apply,
                      "gmres_s": t_gm, **info}))
if world > 1:
    dist.destroy_process_group()
