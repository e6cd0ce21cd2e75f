"""Diagnostic: the sweep as S INDEPENDENT single-stream chains (no look-ahead lanes, no shared update stream): stream i assembles
and factors frequencies i, i + S, ... one after the other with a plan of its own; S streams = S hardware queues.
usage: python tools/seq_streams_sweep.py [S streams] [K frequencies] [stagger 0/1]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MA_LU_LOOKAHEAD"] = os.environ.get("MA_LU_LOOKAHEAD", "0")
import torch
import math_audio_amd as ma
from math_audio_amd import mesh as mm

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 24
stagger = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda", 0)
mesh = mm.generate_sphere_mesh(0.1, 51, 100)
n = mesh.n_elem
freqs = mm.log_space(100.0, 8000.0, 64)
plan = ma.BemPlan(mesh)
lus = [ma.LuPlan(n) for _ in range(S)]
As = [torch.empty(n * n, dtype=torch.complex128, device=dev) for _ in range(S)]
xs = [torch.empty(n, dtype=torch.complex128, device=dev) for _ in range(S)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]


def run(first, count):
    for i in range(count):
        s = i % S
        st = streams[s].cuda_stream
        if stagger and i < S and s > 0:
            with torch.cuda.stream(streams[s]):
                torch.cuda._sleep(int(2.1e9 * 0.150 * s / S))        # start a fraction of a factorisation later (clock ~2.1 GHz)
        k = mm.wave_number(freqs[(first + i) % 64], 343.0); beta = mm.burton_miller_beta_scaled(k, 4.0)
        plan.assemble_dev(k, beta, As[s].data_ptr(), xs[s].data_ptr(), stream=st)
        plan.incident_rhs_dev(k, beta, xs[s].data_ptr(), kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, accumulate=True, stream=st)
        lus[s].factor_solve_dev(As[s].data_ptr(), xs[s].data_ptr(), 1, st)


run(0, S)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(S, K)
torch.cuda.synchronize()
t1 = time.perf_counter()
for s in range(S):
    assert lus[s].status(streams[s].cuda_stream) == 0
    assert bool(torch.isfinite(torch.view_as_real(xs[s])).all())
print(json.dumps({"streams": S, "frequencies": K, "stagger": stagger, "ms_per_frequency": (t1 - t0) * 1e3 / K, "lookahead": os.environ["MA_LU_LOOKAHEAD"]}), flush=True)
