"""Kernels whose workgroups wait for one another (LU panels, the flag-driven Gauss-Seidel / triangular sweeps, the one-launch
Gram-Schmidt step) share ONE admission window per device (lu_kernels.hip "Residency", SpinLaunch). The reference's solvers are
Send + Sync and its FEM driver solves frequencies from worker threads (traits.rs:316, room_simulator_fem.rs:1143-1158): two host
threads -- one inside a GMRES + ILU(0) solve, one inside an LU frequency sweep -- must get the results they get alone, bit for bit,
and no wait may be abandoned."""
import threading

import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import fem
from helpers import to_ma_mesh, RADIUS

pytestmark = pytest.mark.gpu


def test_two_host_threads_krylov_with_ilu_and_lu_sweep(gpu):
    nodes, rp, ci, K, M = fem.helmholtz_box(14, 12, 10)
    n = len(rp) - 1
    c = ma.CsrOperator(rp, ci, K=K, M=M); c.set_wavenumber(1.8 + 0.05j)
    vals = O.helmholtz_values(K, M, 1.8 + 0.05j)
    cs = ma.CsrOperator(rp, ci, values=vals)              # stored values: what the ILU factorises
    op = ma.LinearOperator.csr(cs)
    ilu = ma.IluPreconditioner(cs)
    i = np.arange(n)
    b = np.sin(0.1 * i) + 1j * np.cos(0.2 * i)
    mesh = to_ma_mesh(O.icosphere(RADIUS, 3))
    plan = ma.BemPlan(mesh)
    freqs = [200.0, 400.0, 545.9, 800.0, 1200.0, 1600.0]

    def krylov():
        return ma.gmres_preconditioned(op, ilu, b, restart=30, max_iterations=40, tol=1e-9)

    def sweep():
        return ma.solve_sweep(plan, freqs, slots=3)
    x_ref, info_ref = krylov()
    X_ref, st_ref = sweep()
    assert info_ref.converged == 1 and np.all(st_ref == 0)
    out = {}

    def t1():
        try:
            out["k"] = [krylov() for _ in range(3)]
        except Exception as e:
            out["k"] = e

    def t2():
        try:
            out["s"] = [sweep() for _ in range(2)]
        except Exception as e:
            out["s"] = e
    th = [threading.Thread(target=t1), threading.Thread(target=t2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in th)
    assert not isinstance(out["k"], Exception), out["k"]
    assert not isinstance(out["s"], Exception), out["s"]
    for x, info in out["k"]:
        assert info.iterations == info_ref.iterations and info.converged == 1
        assert np.array_equal(x, x_ref)
    for X, st in out["s"]:
        assert np.all(st == 0) and np.array_equal(X, X_ref)
    assert ma.lib().ma_csr_status(cs.h) == ma.MA_OK
    ilu.close(); op.close(); cs.close(); c.close(); plan.close()


def test_ilu_back_substitution_with_a_broken_pivot(gpu):
    """ilu.rs:154-170: x_i = y_i - sum_j u_ij x_j always; the division by u_ii only where |u_ii| > 1e-30. A zero pivot row must not keep
    whatever the output vector held before."""
    n = 6
    A = np.zeros((n, n), dtype=complex)
    for i in range(n):
        A[i, i] = 2.0 + 0.1j * i
        if i + 1 < n:
            A[i, i + 1] = 0.5; A[i + 1, i] = -0.25
    A[3, 3] = 0.0; A[3, 2] = 0.0                           # u_33 = 0 after the factorisation (no fill-in on this pattern)
    # keep the explicit zero diagonal in the pattern
    rows, cols = np.nonzero(np.abs(A) + np.eye(n)); order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]); vals = A[rows, cols]
    cs = ma.CsrOperator(rp, cols, values=vals)
    ilu = ma.IluPreconditioner(cs)
    ref = O.ilu_module().IluPreconditioner(rp, cols, vals)
    assert abs(ref.u_diag[3]) < 1e-30                       # the case: a pivot the back substitution must not divide by
    r = np.arange(1, n + 1) + 0.5j
    z1 = ilu.apply(r)
    z2 = ilu.apply(r)
    assert np.array_equal(z1, z2) and np.all(np.isfinite(z1.view(float)))
    assert np.abs(z1 - ref.apply(r)).max() <= 1e-12 * max(1.0, np.abs(z1).max())
    ilu.close(); cs.close()
