"""Diagnostic: the staged LU schedule (slots at staggered block indices) against the lock-step batch, same frequencies.
usage: python tools/pipeline_probe.py [K systems] [slots]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import math_audio_amd as ma
from math_audio_amd import mesh as mm

K = int(sys.argv[1]) if len(sys.argv) > 1 else 24
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
mesh = mm.generate_sphere_mesh(0.1, 51, 100)
n = mesh.n_elem
freqs = mm.log_space(100.0, 8000.0, 64)
plan = ma.BemPlan(mesh); lu = ma.LuPlan(n)
As = [torch.empty(n * n, dtype=torch.complex128, device=dev) for _ in range(S)]
xs = [torch.empty(n, dtype=torch.complex128, device=dev) for _ in range(S)]
keep = [torch.empty(n, dtype=torch.complex128, device=dev) for _ in range(K)]
st = torch.cuda.current_stream().cuda_stream


def assemble(i, s, stream=None):
    stream = st if stream is None else stream
    k = mm.wave_number(freqs[i % 64], 343.0); beta = mm.burton_miller_beta_scaled(k, 4.0)
    plan.assemble_dev(k, beta, As[s].data_ptr(), xs[s].data_ptr(), stream=stream)
    plan.incident_rhs_dev(k, beta, xs[s].data_ptr(), kind=0, vec=(0.0, 0.0, 1.0), amp=1.0, accumulate=True, stream=stream)


def batch_mode():
    i = 0
    while i < K:
        c = min(S, K - i)
        for j in range(c):
            assemble(i + j, j)
        lu.factor_solve_batch_dev([a.data_ptr() for a in As[:c]], [v.data_ptr() for v in xs[:c]], 1, stream=st)
        for j in range(c):
            keep[i + j].copy_(xs[j])
        i += c


ASM_ON_LANE = False


def staged_mode():
    G = lu.num_blocks()
    lanes = [lu.slot_stream(s) for s in range(S)]
    ext = [torch.cuda.ExternalStream(p) for p in lanes]
    copied = [None] * S
    off = [s * ((G + S - 1) // S) for s in range(S)]
    lu.stage_reset(st)
    r = 0
    while True:
        sl, bl, active = [], [], False
        for s in range(S):
            lr = r - off[s]
            if lr < 0:
                active = True
                continue
            sysno, g = divmod(lr, G)
            idx = s + S * sysno
            if idx >= K:
                continue
            active = True
            if g == 0:
                if ASM_ON_LANE:
                    if copied[s] is not None:
                        ext[s].wait_event(copied[s])       # the previous solution of this slot has been copied out
                    else:
                        ev0 = torch.cuda.Event(); ev0.record(); ext[s].wait_event(ev0)
                    assemble(idx, s, lanes[s])
                    lu.stage_begin(s, As[s].data_ptr(), xs[s].data_ptr(), 1, lanes[s])
                else:
                    assemble(idx, s)
                    lu.stage_begin(s, As[s].data_ptr(), xs[s].data_ptr(), 1, st)
            sl.append(s); bl.append(g)
        if not active:
            break
        if sl:
            lu.stage_round(sl, bl, st)
        for s, g in zip(sl, bl):
            if g == G - 1:
                lu.stage_finish(s, st)
                idx = s + S * ((r - off[s]) // G)
                keep[idx].copy_(xs[s])
                copied[s] = torch.cuda.Event(); copied[s].record()
        r += 1


def staged_lane_asm():
    global ASM_ON_LANE
    ASM_ON_LANE = True
    try:
        staged_mode()
    finally:
        ASM_ON_LANE = False


for name, fn in (("batch", batch_mode), ("staged", staged_mode), ("staged+asm-on-lane", staged_lane_asm), ("staged", staged_mode), ("staged+asm-on-lane", staged_lane_asm)):
    fn(); torch.cuda.synchronize()          # warm-up of the mode
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert lu.status(st) == 0
    res = [k.clone() for k in keep]
    print("%-19s K=%d slots=%d: %.2f ms per system" % (name, K, S, dt / K * 1e3))
    if name == "batch":
        ref = res
    else:
        err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(res, ref))
        print("        staged vs batch solutions: max rel diff %.2e" % err)
