"""GPU parity tests of the path bench.py measures: the frequency loop of room_simulator_bem.rs:328-360 as the library's sweep
handle runs it at BASELINE.json's size (S10, 10 000 panels) with the DEFAULT plan -- 64-column panels as two register
half-panels, the big updates on the CU-masked stream, three slots a third of a factorisation apart, the next three systems
assembled ahead in twelve pieces into swapped spares -- against the single-system path (one assembly, one factor_solve_dev)
and the CPU restatement; and the small-mesh end of the same entry point (a plan of ONE block, more slots than blocks)."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import mesh as mm
from helpers import to_ma_mesh, RADIUS, rel_l2

pytestmark = pytest.mark.gpu
C_SOUND = 343.0


def _single_system(plan, lu, k, beta, A, x, st):
    """One system on the single-system path: ma_bem_plan_assemble_dev (tbem_far_kernel<1, .>) + incident RHS + factor_solve_dev."""
    plan.assemble_dev(k, beta, A.data_ptr(), x.data_ptr(), stream=st)
    plan.incident_rhs_dev(k, beta, x.data_ptr(), accumulate=True, stream=st)
    lu.factor_solve_dev(A.data_ptr(), x.data_ptr(), 1, stream=st)
    assert lu.status(st) == ma.MA_OK
    return x.cpu().numpy()


@pytest.mark.parametrize("slots", [3, 4])
def test_default_sweep_of_a_one_block_plan(gpu, slots):
    """an 80-panel icosphere is ONE block of the LU plan, so three or four slots begin in the same round and
    slot 0 takes frequencies 0, 3, 6 while slots 1 and 2 are still at 1 and 2 -- the spare-system bookkeeping of the assembly-ahead
    assumed frequencies are consumed in order and failed with 'no spare system for frequency 6'. 16 frequencies across the
    ka = 0.5 sign switch (tbem.rs:108-123), default slots, against the per-frequency one-shot path."""
    om = O.icosphere(RADIUS, 1)
    mesh = to_ma_mesh(om)
    assert om.n_elem == 80
    plan = ma.BemPlan(mesh)
    freqs = list(np.geomspace(120.0, 4000.0, 16))
    X, st = ma.solve_sweep(plan, freqs, speed_of_sound=C_SOUND, beta_scale=4.0, slots=slots)
    assert np.all(st == ma.MA_OK)
    sw = ma.BemSweep(plan, len(freqs), slots=slots)
    info = sw.info()
    assert info["blocks"] == 1 and info["systems_ahead"] == 1      # nothing is assembled ahead when the slots do not take the frequencies in order
    X2, st2 = sw.run(freqs, speed_of_sound=C_SOUND, beta_scale=4.0)
    X3, st3 = sw.run(freqs[:5], speed_of_sound=C_SOUND, beta_scale=4.0)       # a handle is reusable, also for fewer frequencies
    sw.close()
    assert np.array_equal(X2, X) and np.array_equal(X3, X[:5]) and np.all(st2 == 0) and np.all(st3 == 0)
    for fi, f in enumerate(freqs):
        k = O.wave_number(f, C_SOUND); beta = complex(0.0, 4.0 / k)
        A, r0 = ma.assemble_tbem(mesh, k, beta)
        x1 = ma.zgesv(A, r0 + ma.incident_rhs(om.center, om.normal, k, beta))
        assert rel_l2(X[fi], x1) <= 1e-10, fi
    plan.close()


def test_sweep_handle_with_spares_on_a_small_plan(gpu):
    """640 panels = 10 panels of 64 columns: with MA_LU_KB the plan has several blocks, the slots take the frequencies in order and
    the assembly-ahead with its two sets of spares runs (the same bookkeeping as at S10, at a size the one-shot path checks in
    seconds); two runs of one handle give the same bits."""
    import os
    om = O.icosphere(RADIUS, 3)
    mesh = to_ma_mesh(om)
    old = os.environ.get("MA_LU_KB")
    os.environ["MA_LU_KB"] = "1"
    try:
        plan = ma.BemPlan(mesh)
        freqs = list(np.geomspace(150.0, 3000.0, 11))
        sw = ma.BemSweep(plan, len(freqs), slots=3)
        info = sw.info()
        assert info["staged"] and info["systems_ahead"] == 3 and (info["slots"] - 1) * info["spacing"] < info["blocks"], info
        X, st = sw.run(freqs, speed_of_sound=C_SOUND, beta_scale=4.0)
        Xb, stb = sw.run(freqs, speed_of_sound=C_SOUND, beta_scale=4.0, to_host=False)
        assert Xb is None and np.all(st == 0) and np.all(stb == 0)
        import torch
        ptr, cnt = sw.solutions_dev()
        assert cnt == len(freqs)
        Xd = torch.empty(cnt * om.n_elem, dtype=torch.complex128, device="cuda")
        ma.memcpy_dtod(Xd.data_ptr(), ptr, Xd.numel() * 16)
        assert np.array_equal(Xd.cpu().numpy().reshape(cnt, -1), X)
        sw.close()
    finally:
        if old is None:
            del os.environ["MA_LU_KB"]
        else:
            os.environ["MA_LU_KB"] = old
    for fi, f in enumerate(freqs):
        k = O.wave_number(f, C_SOUND); beta = complex(0.0, 4.0 / k)
        A, r0 = ma.assemble_tbem(mesh, k, beta)
        x1 = ma.zgesv(A, r0 + ma.incident_rhs(om.center, om.normal, k, beta))
        assert rel_l2(X[fi], x1) <= 1e-10, fi
    plan.close()


def test_sweep_mode_switches_on_a_small_plan(gpu):
    """MA_SWEEP_PIVOTING / MA_SWEEP_SPECULATE (read when a handle is created): the sweep's plan in partial pivoting, with verified
    speculation, and with none, on the 640-panel staged plan -- each against the default handle (tournament, optimistic) to 1e-12:
    an accepted half-panel carries LAPACK's pivots in every mode, a tournament panel other rows of a stable factorisation."""
    import os
    om = O.icosphere(RADIUS, 3)
    mesh = to_ma_mesh(om)
    freqs = list(np.geomspace(150.0, 3000.0, 7))
    saved = {v: os.environ.get(v) for v in ("MA_LU_KB", "MA_SWEEP_PIVOTING", "MA_SWEEP_SPECULATE")}
    os.environ["MA_LU_KB"] = "1"
    try:
        plan = ma.BemPlan(mesh)
        sw = ma.BemSweep(plan, len(freqs), slots=3)
        assert sw.lu_plan().pivoting() == "tournament" and sw.lu_plan().speculation() == "optimistic"
        X, st = sw.run(freqs, speed_of_sound=C_SOUND, beta_scale=4.0)
        sw.close()
        assert np.all(st == ma.MA_OK)
        for piv, spec, want in (("partial", "verified", ("partial", "verified")), ("tournament", "off", ("tournament", "off")),
                                ("partial", "off", ("partial", "off")), ("tournament", "verified", ("tournament", "verified"))):
            os.environ["MA_SWEEP_PIVOTING"] = piv; os.environ["MA_SWEEP_SPECULATE"] = spec
            sw = ma.BemSweep(plan, len(freqs), slots=3)
            got = (sw.lu_plan().pivoting(), sw.lu_plan().speculation())
            Xm, stm = sw.run(freqs, speed_of_sound=C_SOUND, beta_scale=4.0)
            sw.close()
            assert got == want, (got, want)
            assert np.all(stm == ma.MA_OK)
            for fi in range(len(freqs)):
                assert rel_l2(Xm[fi], X[fi]) <= 1e-12, (piv, spec, fi, rel_l2(Xm[fi], X[fi]))
        plan.close()
    finally:
        for v, val in saved.items():
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val


def test_s10_sweep_as_benchmarked_equals_the_single_system_path(gpu):
    """VERDICT r3 item 2. S10 through ma_bem_sweep_run with the default plan, nine frequencies of the 64-point list spanning the
    ka = 0.5 sign switch (indices 14 / 15: ka = 0.485 / 0.520) and both ends: every solution against the single-system path
    (<= 1e-12 relative L2: same LU arithmetic, the multi-frequency far kernel sums in a different order), the solve's residual
    ||A x - b|| / ||b|| <= 1e-10 with A and b re-assembled on the single-system path, and sampled rows of the assembly AS THE SWEEP
    ISSUES IT (three systems per pass, twelve pieces: tbem_far_kernel<3, true>) against the CPU restatement to 1e-9 of the row scale."""
    import torch
    mesh = mm.generate_sphere_mesh(RADIUS, 51, 100)
    n = mesh.n_elem
    assert n == 10000
    fl = mm.log_space(100.0, 8000.0, 64)
    idx = [0, 7, 14, 15, 23, 32, 47, 56, 63]
    freqs = [fl[i] for i in idx]
    ks = [mm.wave_number(f) for f in freqs]
    assert ks[2] * RADIUS < 0.5 < ks[3] * RADIUS
    plan = ma.BemPlan(mesh)
    sw = ma.BemSweep(plan, len(freqs), slots=3)
    info = sw.info()
    assert info["staged"] and info["slots"] == 3 and info["systems_ahead"] == 3 and info["blocks"] >= 20, info   # the benchmarked configuration, not a fallback
    slu = sw.lu_plan()
    assert slu.main_stream(), "the default 10 000-row plan runs its big updates on the CU-masked stream"
    # round 5: the sweep's plan factors its half-panels speculatively (lu_spec.hip), without a fallback behind them (a frequency that met a
    # rejected panel would be solved again), and its own panel kernel is the tournament
    assert slu.pivoting() == "tournament" and slu.speculation() == "optimistic"
    X, st = sw.run(freqs, speed_of_sound=C_SOUND, beta_scale=4.0)
    assert np.all(st == ma.MA_OK) and np.all(np.isfinite(X.view(np.float64)))
    acc, wid, rej = slu.speculation_stats()
    print("S10, 9 frequencies: half-panels accepted at once %d, by a widened attempt %d, rejected %d" % (acc, wid, rej))
    assert acc + wid + rej == 9 * 313 and rej == 0 and wid > 0      # the UV sphere's poles need the widened attempt; nothing is left to the tournament
    sw.close()
    dev = torch.device("cuda", 0)
    s0 = torch.cuda.current_stream().cuda_stream
    # the single-system path in the sweep's own pivoting mode (accepted half-panels carry LAPACK's pivots in either mode: the same bits)
    lu = ma.LuPlan(n, pivoting="tournament")
    A = torch.empty(n * n, dtype=torch.complex128, device=dev); x = torch.empty(n, dtype=torch.complex128, device=dev)
    worst = 0.0
    for fi, k in enumerate(ks):
        beta = mm.burton_miller_beta_scaled(k, 4.0)
        x1 = _single_system(plan, lu, k, beta, A, x, s0)
        err = rel_l2(X[fi], x1)
        worst = max(worst, err)
        assert err <= 1e-12, (idx[fi], err)
        # residual with a fresh copy of the system (the factorisation destroyed A)
        plan.assemble_dev(k, beta, A.data_ptr(), x.data_ptr(), stream=s0)
        plan.incident_rhs_dev(k, beta, x.data_ptr(), accumulate=True, stream=s0)
        xs = torch.from_numpy(X[fi]).to(dev)
        res = float(torch.linalg.norm(A.view(n, n) @ xs - x) / torch.linalg.norm(x))
        assert res <= 1e-10, (idx[fi], res)
    lu.close()
    # and against LAPACK-style partial pivoting without any speculation (the round-4 path), both ends of the list and the sign switch:
    # two backward-stable solves of a well-conditioned system; with the growth of the factors printed (max |L|, max |U| / max |A|)
    import os
    os.environ["MA_LU_SPECULATE"] = "0"
    try:
        lup = ma.LuPlan(n)
    finally:
        del os.environ["MA_LU_SPECULATE"]
    assert lup.pivoting() == "partial" and lup.speculation() == "off"
    for fi in (0, 3, 8):
        k = ks[fi]; beta = mm.burton_miller_beta_scaled(k, 4.0)
        plan.assemble_dev(k, beta, A.data_ptr(), x.data_ptr(), stream=s0)
        amax = float(A.abs().max())
        xp = _single_system(plan, lup, k, beta, A, x, s0)
        F = A.view(n, n)
        lmax = float(torch.tril(F, -1).abs().max()); umax = float(torch.triu(F).abs().max())
        print("S10 f[%d]: partial pivoting max |L| = %.3f, growth max |U| / max |A| = %.3f; sweep vs partial rel L2 = %.2e" % (idx[fi], lmax, umax / amax, rel_l2(X[fi], xp)))
        assert lmax <= 2.0 ** 0.5 + 1e-9 and umax / amax <= 16.0     # izamax compares |re| + |im|: a multiplier's modulus can reach sqrt(2)
        assert rel_l2(X[fi], xp) <= 1e-11, (idx[fi], rel_l2(X[fi], xp))
    lup.close()
    del A
    # the assembly as the sweep issues it: three systems per pass, in twelve pieces
    om = O.uv_sphere(RADIUS, 51, 100)
    om.nodes[:] = mesh.nodes; om.center[:] = mesh.center; om.normal[:] = mesh.normal; om.area[:] = mesh.area
    trio = [2, 3, 8]                                             # ka = 0.485 (sign +1), 0.520 (sign -1), 14.7
    kk = [ks[t] for t in trio]; bb = [mm.burton_miller_beta_scaled(k, 4.0) for k in kk]
    As = [torch.zeros(n * n, dtype=torch.complex128, device=dev) for _ in trio]
    rs = [torch.zeros(n, dtype=torch.complex128, device=dev) for _ in trio]
    for part in range(12):
        plan.assemble_multi_part_dev(kk, bb, [a.data_ptr() for a in As], [r.data_ptr() for r in rs], part, 12, stream=s0)
    torch.cuda.synchronize()
    for q in range(3):
        Aq = As[q].view(n, n)
        for r0 in (0, 137, 5000, 9999):
            S, _ = O.build_tbem_rows(om, kk[q], bb[q], r0, r0 + 1)
            row = Aq[r0].cpu().numpy()
            assert np.abs(row - S[0]).max() <= 1e-9 * np.abs(S[0]).max(), (trio[q], r0)
    plan.close()
