"""GPU parity of the operator boundary (dense, CSR, matrix-free TBEM) and of device GMRES(m) against the
CPU oracle. gmres.rs:632-705 known answers; the matrix-free operator must equal A x of the dense TBEM
matrix."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import fem
from helpers import to_ma_mesh, k_from_ka, RADIUS, rel_l2

pytestmark = pytest.mark.gpu


def test_dense_and_csr_operators_apply(gpu):
    rng = np.random.default_rng(2)
    n = 333
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)); x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    op = ma.LinearOperator.dense(A)
    assert np.abs(op.apply(x) - A @ x).max() <= 1e-13 * np.abs(A @ x).max()
    op.close()
    nodes, rp, ci, K, M = fem.helmholtz_box(7, 6, 5)
    c = ma.CsrOperator(rp, ci, K=K, M=M); c.set_wavenumber(1.8 + 0.01j)
    op = ma.LinearOperator.csr(c)
    xx = np.sin(0.1 * np.arange(op.n)) + 1j * np.cos(0.2 * np.arange(op.n))
    ref = O.csr_matvec(rp, ci, O.helmholtz_values(K, M, 1.8 + 0.01j), xx)
    assert np.abs(op.apply(xx) - ref).max() <= 1e-13 * np.abs(ref).max()
    op.close(); c.close()


@pytest.mark.parametrize("sub,ka", [(1, 0.2), (2, 1.0), (2, 3.0)])
def test_matrix_free_tbem_equals_dense_matvec(gpu, sub, ka):
    om = O.icosphere(RADIUS, sub)
    k = k_from_ka(ka); beta, _ = O.beta_adaptive(k, RADIUS)
    mesh = to_ma_mesh(om)
    A, _ = ma.assemble_tbem(mesh, k, beta)
    A_ref, _ = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    plan = ma.BemPlan(mesh)
    op = ma.LinearOperator.tbem(plan, k, beta)
    i = np.arange(om.n_elem)
    x = np.sin(0.1 * i) + 1j * np.cos(0.2 * i)                     # tests/test_fmm_validation.rs:121
    y = op.apply(x)
    assert np.abs(y - A @ x).max() <= 1e-12 * np.abs(A @ x).max()   # same entries, different summation order
    assert np.abs(y - A_ref @ x).max() <= 1e-9 * np.abs(A_ref @ x).max()
    # row-block form (one block per GPU when sharded): the blocks tile the product
    n = om.n_elem; h = n // 3
    parts = np.zeros(n, dtype=complex)
    for r0, r1 in ((0, h), (h, 2 * h), (2 * h, n)):
        blk = ma.LinearOperator.tbem(plan, k, beta, rows=(r0, r1))
        parts[r0:r1] = blk.apply(x)[r0:r1]
        blk.close()
    assert np.array_equal(parts, y)
    # apply_transpose / apply_hermitian (traits.rs:326-358) of the streamed operator: same entries, loop nest turned around
    yt = op.apply_transpose(x)
    assert np.abs(yt - A.T @ x).max() <= 1e-12 * np.abs(A.T @ x).max()
    yh = op.apply_hermitian(x)
    assert np.abs(yh - A.conj().T @ x).max() <= 1e-12 * np.abs(A.conj().T @ x).max()
    # a row block contributes its rows' part of A^T x: the blocks' results add up
    acc = np.zeros(n, dtype=complex)
    for r0, r1 in ((0, h), (h, 2 * h), (2 * h, n)):
        blk = ma.LinearOperator.tbem(plan, k, beta, rows=(r0, r1))
        part = blk.apply_transpose(x)
        assert np.abs(part - A[r0:r1].T @ x[r0:r1]).max() <= 1e-12 * np.abs(A.T @ x).max()
        acc += part
        blk.close()
    assert np.abs(acc - yt).max() <= 1e-13 * np.abs(yt).max()
    op.close(); plan.close()


def test_gmres_known_answers(gpu):                                  # gmres.rs:632-705
    A = np.array([[4.0, 1.0], [1.0, 3.0]], dtype=complex); b = np.array([1.0, 2.0], dtype=complex)
    op = ma.LinearOperator.dense(A)
    x, info = ma.gmres(op, b, restart=10, max_iterations=10, tol=1e-10)
    assert info.converged == 1 and np.abs(A @ x - b).max() < 1e-8
    x, info = ma.gmres(op, np.zeros(2, dtype=complex))
    assert info.converged == 1 and info.iterations == 0 and np.all(x == 0)
    x, info = ma.gmres(op, b, x0=np.linalg.solve(A, b), tol=1e-8)
    assert info.converged == 1 and info.iterations == 0            # the guess already satisfies the tolerance
    op.close()
    I = ma.LinearOperator.dense(np.eye(5, dtype=complex))
    x, info = ma.gmres(I, np.arange(1, 6).astype(complex), tol=1e-12)
    assert info.converged == 1 and np.abs(x - np.arange(1, 6)).max() < 1e-10
    I.close()


def test_gmres_matches_oracle_on_bem_system(gpu):
    """qa_suite-style system (icosphere 2, ka = 1): same iteration count and solution as the CPU restatement,
    with the dense, and with the matrix-free operator; GMRES and LU agree."""
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(1.0); beta, _ = O.beta_adaptive(k, RADIUS)
    A, r0 = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    b = r0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
    x_ref, info_ref = O.gmres(b, dense=A, restart=50, max_iterations=20, tol=1e-6)
    assert info_ref.converged == 1
    op = ma.LinearOperator.dense(A)
    x, info = ma.gmres(op, b, restart=50, max_iterations=20, tol=1e-6)
    assert info.converged == 1 and info.iterations == info_ref.iterations and info.restarts == info_ref.restarts
    assert rel_l2(x, x_ref) <= 1e-9
    op.close()
    plan = ma.BemPlan(to_ma_mesh(om))
    mf = ma.LinearOperator.tbem(plan, k, beta)
    x2, info2 = ma.gmres(mf, b, restart=50, max_iterations=20, tol=1e-6)
    assert info2.converged == 1 and info2.iterations == info_ref.iterations
    assert rel_l2(x2, x_ref) <= 1e-7
    x_lu, _, rc = O.zgesv(A, b)
    assert rc == 0 and rel_l2(x2, x_lu) <= 1e-4                     # GMRES tolerance 1e-6 on the residual
    mf.close(); plan.close()


def test_gmres_restarts_and_nonconvergence_flag(gpu):
    rng = np.random.default_rng(9)
    n = 120
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) + 3.0 * np.eye(n); b = rng.standard_normal(n) + 0j
    op = ma.LinearOperator.dense(A)
    x_ref, info_ref = O.gmres(b, dense=A, restart=10, max_iterations=3, tol=1e-12)
    x, info = ma.gmres(op, b, restart=10, max_iterations=3, tol=1e-12)
    assert info.converged == info_ref.converged == 0                # not an error, a flag (gmres.rs:270-276)
    assert info.iterations == info_ref.iterations == 30 and info.restarts == info_ref.restarts == 3
    assert rel_l2(x, x_ref) <= 1e-8 and abs(info.residual - info_ref.residual) <= 1e-8 * info_ref.residual
    op.close()


def test_preconditioned_gmres_matches_oracle_on_fem_system(gpu):
    """P1 Helmholtz box system with a damped wavenumber: GMRES through the Preconditioner boundary (Jacobi and
    l1-Jacobi sweeps from zero) vs the CPU restatement of gmres_preconditioned (gmres.rs:282-428)."""
    nodes, rp, ci, K, M = fem.helmholtz_box(9, 8, 7)
    n = len(rp) - 1
    k = 0.3 + 1.5j
    c = ma.CsrOperator(rp, ci, K=K, M=M); c.set_wavenumber(k)
    vals = O.helmholtz_values(K, M, k)
    b = np.cos(0.3 * np.arange(n)) + 0.5j
    op = ma.LinearOperator.csr(c)
    its = {}
    for kind, pk in (("jacobi", 1), ("l1", 2)):
        pre = ma.Preconditioner(c, kind=kind, omega=0.8, sweeps=2)
        x_ref, info_ref = O.gmres_preconditioned(b, (rp, ci, vals), pkind=pk, omega=0.8, sweeps=2, restart=30, max_iterations=20, tol=1e-8)
        x, info = ma.gmres_preconditioned(op, pre, b, restart=30, max_iterations=20, tol=1e-8)
        assert info_ref.converged == 1 and info.converged == 1
        assert info.iterations == info_ref.iterations and info.restarts == info_ref.restarts
        assert rel_l2(x, x_ref) <= 1e-9
        assert np.linalg.norm(O.csr_matvec(rp, ci, vals, x) - b) / np.linalg.norm(b) < 1e-6
        its[kind] = info.iterations
        pre.close()
    _, info_plain = ma.gmres(op, b, restart=30, max_iterations=20, tol=1e-8)
    assert its["jacobi"] < info_plain.iterations          # the preconditioner pays on this system
    op.close(); c.close()


def test_diagonal_preconditioner_known_answers(gpu):               # preconditioners/diagonal.rs:106-145
    d = ma.CsrOperator([0, 1, 2, 3], [0, 1, 2], values=[2, 4, 1])
    pre = ma.Preconditioner(d, kind="jacobi", omega=1.0, sweeps=1)
    assert np.allclose(pre.apply([2, 8, 3]), [1, 2, 3], atol=1e-10)
    pre.close(); d.close()
    a = ma.CsrOperator([0, 2, 4], [0, 1, 0, 1], values=[4, 1, 1, 2])
    pre = ma.Preconditioner(a, kind="jacobi", omega=1.0, sweeps=1)
    assert np.allclose(pre.apply([4, 4]), [1, 2], atol=1e-10)
    pre.close(); a.close()


def test_apply_transpose_and_hermitian(gpu):
    """LinearOperator::apply_transpose / apply_hermitian (traits.rs:326-358) for the dense and the CSR operator;
    hermitian == conj(A^T conj(x)) exactly as the trait's default implementation states it."""
    rng = np.random.default_rng(31)
    n = 777
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    op = ma.LinearOperator.dense(A)
    scale = np.abs(A.T @ x).max()
    assert np.abs(op.apply_transpose(x) - A.T @ x).max() <= 1e-12 * scale
    assert np.abs(op.apply_hermitian(x) - A.conj().T @ x).max() <= 1e-12 * scale
    op.close()
    # unsymmetric sparse operator with empty rows and columns
    dens = 0.02
    S = np.where(rng.random((n, n)) < dens, A, 0.0); S[5, :] = 0.0; S[:, 9] = 0.0
    rp = np.concatenate(([0], np.cumsum((S != 0).sum(axis=1)))); ci = np.nonzero(S)[1]; vals = S[S != 0]
    c = ma.CsrOperator(rp, ci, values=vals)
    ops = ma.LinearOperator.csr(c)
    assert np.abs(ops.apply(x) - S @ x).max() <= 1e-12 * np.abs(S @ x).max()
    assert np.abs(ops.apply_transpose(x) - S.T @ x).max() <= 1e-12 * np.abs(S.T @ x).max()
    assert np.abs(ops.apply_hermitian(x) - S.conj().T @ x).max() <= 1e-12 * np.abs(S.T @ x).max()
    assert np.array_equal(ops.apply_hermitian(x), np.conj(ops.apply_transpose(np.conj(x))))
    ops.close(); c.close()
    # K/M mode keeps its wavenumber through the transpose
    nodes, rp2, ci2, K, M = fem.helmholtz_box(5, 4, 3)
    h = ma.CsrOperator(rp2, ci2, K=K, M=M); h.set_wavenumber(1.3 + 0.2j)
    oph = ma.LinearOperator.csr(h)
    x2 = _xvec(len(rp2) - 1)
    ref = O.csr_matvec(rp2, ci2, O.helmholtz_values(K, M, 1.3 + 0.2j), x2)     # K, M symmetric: A^T = A
    assert np.abs(oph.apply_transpose(x2) - ref).max() <= 1e-12 * np.abs(ref).max()
    oph.close(); h.close()


def _xvec(n):
    i = np.arange(n)
    return np.sin(0.1 * i) + 1j * np.cos(0.2 * i)


def test_row_sharded_operator_and_gmres_on_one_gpu(gpu):
    """math_audio_amd.sharded at world size 1 (the all-gather degenerates; the 2-rank path is covered with gloo on the CPU):
    the row-block operator equals the dense product and the replicated GMRES equals the single-GPU ma_gmres."""
    import torch
    from math_audio_amd import sharded
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(1.0); beta, _ = O.beta_adaptive(k, RADIUS)
    mesh = to_ma_mesh(om)
    plan = ma.BemPlan(mesh)
    A, _ = ma.assemble_tbem(mesh, k, beta)
    dev = torch.device("cuda", 0)
    so = sharded.tbem_sharded_operator(plan, k, beta, dist=None, device=dev)
    x = np.cos(0.37 * np.arange(plan.num_dofs)) + 1j * np.sin(0.11 * np.arange(plan.num_dofs))
    y = so.apply(torch.tensor(x, device=dev)).cpu().numpy()
    assert np.abs(y - A @ x).max() <= 1e-10 * np.abs(A @ x).max()
    b = ma.incident_rhs(om.center, om.normal, k, beta)
    xs, info = sharded.gmres(so, torch.tensor(b, device=dev), restart=30, max_iterations=10, tol=1e-8)
    op = ma.LinearOperator.tbem(plan, k, beta)
    xr, info_r = ma.gmres(op, b, restart=30, max_iterations=10, tol=1e-8)
    assert info["converged"] and info_r.converged == 1
    assert info["iterations"] == info_r.iterations and info["restarts"] == info_r.restarts
    assert rel_l2(xs.cpu().numpy(), xr) <= 1e-9
    assert np.linalg.norm(A @ xs.cpu().numpy() - b) / np.linalg.norm(b) < 1e-7
    op.close(); plan.close()


def test_diagonal_preconditioner_of_an_operator(gpu):
    """DiagonalPreconditioner::from_diagonal (math-bem/src/core/solver/fmm_interface.rs:177-212) for the dense and the
    matrix-free operator: z = r / a_ii with the operator's true diagonal; left-preconditioned GMRES (tolerance relative to
    |M^-1 b|, gmres.rs:299-300) converges to the solution of plain GMRES."""
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(1.0); beta, _ = O.beta_adaptive(k, RADIUS)
    mesh = to_ma_mesh(om)
    A, _ = ma.assemble_tbem(mesh, k, beta)
    plan = ma.BemPlan(mesh)
    r = _xvec(om.n_elem)
    for op in (ma.LinearOperator.dense(A), ma.LinearOperator.tbem(plan, k, beta)):
        Mp = ma.Preconditioner(op, kind="diagonal")
        z = Mp.apply(r)
        assert np.abs(z - r / np.diag(A)).max() <= 1e-12 * np.abs(z).max()
        b = ma.incident_rhs(om.center, om.normal, k, beta)
        x0, i0 = ma.gmres(op, b, restart=30, max_iterations=200, tol=1e-8)
        x1, i1 = ma.gmres_preconditioned(op, Mp, b, restart=30, max_iterations=200, tol=1e-8)
        assert i0.converged == 1 and i1.converged == 1
        assert np.linalg.norm(x1 - x0) <= 1e-6 * np.linalg.norm(x0)
        Mp.close(); op.close()
    D = ma.LinearOperator.dense(np.array([[0.0, 1.0], [1.0, 2.0]], dtype=complex))     # zero diagonal entry: left alone (:199-204)
    Mz = ma.Preconditioner(D, kind="diagonal")
    assert np.allclose(Mz.apply(np.array([3.0, 4.0], dtype=complex)), [3.0, 2.0])
    Mz.close(); D.close(); plan.close()


def test_stored_operator_equals_matrix_free(gpu):
    """The MI355X-first form of the iterative path: the system assembled into HBM once (ma_bem_plan_assemble_dev) and wrapped as a
    dense operator (ma_op_create_dense_dev, borrowed device pointer) must act like the matrix-free operator, and GMRES with
    the diagonal preconditioner reaches the same solution on both."""
    import torch
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(1.0); beta, _ = O.beta_adaptive(k, RADIUS)
    mesh = to_ma_mesh(om)
    plan = ma.BemPlan(mesh)
    n = plan.num_dofs
    dev = torch.device("cuda", 0)
    A = torch.empty(n * n, dtype=torch.complex128, device=dev); r0 = torch.empty(n, dtype=torch.complex128, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    plan.assemble_dev(k, beta, A.data_ptr(), r0.data_ptr(), stream=st)
    torch.cuda.synchronize()
    stored = ma.LinearOperator.dense_dev(n, A.data_ptr(), keep=A)
    free = ma.LinearOperator.tbem(plan, k, beta)
    x = _xvec(n)
    ys, yf = stored.apply(x), free.apply(x)
    assert np.abs(ys - yf).max() <= 1e-12 * np.abs(ys).max()
    assert np.abs(stored.apply_transpose(x) - free.apply_transpose(x)).max() <= 1e-12 * np.abs(ys).max()
    b = ma.incident_rhs(om.center, om.normal, k, beta)
    sols = []
    for op in (stored, free):
        Mp = ma.Preconditioner(op, kind="diagonal")
        xs, info = ma.gmres_preconditioned(op, Mp, b, restart=30, max_iterations=200, tol=1e-10)
        assert info.converged == 1
        sols.append(xs); Mp.close()
    assert np.linalg.norm(sols[0] - sols[1]) <= 1e-8 * np.linalg.norm(sols[0])
    stored.close(); free.close(); plan.close()


def test_row_sharded_operator_inside_the_library(gpu):
    """ma_op_create_tbem_multi: the matrix-free operator row-sharded over devices with the y all-gather done by peer copies
    inside the library, driven by the library's own device GMRES. A one-GPU box runs the shards on
    the same device -- which only the DIAGNOSTIC build of the library accepts (MA_TEST_ALLOW_DUPLICATE_DEVICES; a process of its own):
    block bounds, the x broadcast, the slice gather, the summed transposed apply, the diagonal preconditioner and GMRES are all the
    multi-GPU code path. The shipped library refuses a device listed twice."""
    from test_lu_gpu import _run_with_diagnostic_library
    code = r'''
import numpy as np
import oracle_lib as O
import math_audio_amd as ma
from helpers import to_ma_mesh, k_from_ka, RADIUS, rel_l2
om = O.icosphere(RADIUS, 2)
k = k_from_ka(1.0); beta, _ = O.beta_adaptive(k, RADIUS)
mesh = to_ma_mesh(om)
A, _ = ma.assemble_tbem(mesh, k, beta)
n = om.n_elem
i = np.arange(n)
x = np.sin(0.1 * i) + 1j * np.cos(0.2 * i)
for devs in ([0], [0, 0], [0, 0, 0]):
    op = ma.LinearOperator.tbem_multi(mesh, k, beta, devs)
    rows, dv = op.shards()
    assert list(rows) == [n * g // len(devs) for g in range(len(devs))] and list(dv) == devs
    y = op.apply(x)
    assert np.abs(y - A @ x).max() <= 1e-12 * np.abs(A @ x).max()
    assert np.abs(op.apply_transpose(x) - A.T @ x).max() <= 1e-12 * np.abs(A.T @ x).max()
    assert np.abs(op.apply_hermitian(x) - A.conj().T @ x).max() <= 1e-12 * np.abs(A.T @ x).max()
    b = ma.incident_rhs(om.center, om.normal, k, beta)
    single = ma.LinearOperator.tbem(ma.BemPlan(mesh), k, beta)
    xr, ir = ma.gmres(single, b, restart=30, max_iterations=10, tol=1e-8)
    xs, info = ma.gmres(op, b, restart=30, max_iterations=10, tol=1e-8)
    assert info.converged == 1 and info.iterations == ir.iterations and info.restarts == ir.restarts
    assert rel_l2(xs, xr) <= 1e-10
    Mp = ma.Preconditioner(op, kind="diagonal")
    z = Mp.apply(x)
    assert np.abs(z - x / np.diag(A)).max() <= 1e-12 * np.abs(z).max()
    xp, ip = ma.gmres_preconditioned(op, Mp, b, restart=30, max_iterations=10, tol=1e-8)
    assert ip.converged == 1 and np.linalg.norm(A @ xp - b) <= 1e-6 * np.linalg.norm(b)
    Mp.close(); op.close(); single.close()
print("ok")
'''
    r = _run_with_diagnostic_library(code, {"MA_TEST_ALLOW_DUPLICATE_DEVICES": 1})
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(1.0); beta, _ = O.beta_adaptive(k, RADIUS)
    with pytest.raises(ma.MaError) as e:
        ma.LinearOperator.tbem_multi(to_ma_mesh(om), k, beta, [0, 0])
    assert e.value.status == ma.MA_ERR_INVALID


def test_pipelined_gmres_on_the_device(gpu):
    """gmres_pipelined (gmres_pipelined.rs:18-250): the reference's own 2 x 2 test (:258-285), a dense BEM system, and a CSR
    operator with the Jacobi and the identity preconditioner, against the restatement: same iteration counts, same solution."""
    A2 = np.array([[4.0, 1.0], [1.0, 3.0]], dtype=complex); b2 = np.array([1.0, 2.0], dtype=complex)
    op = ma.LinearOperator.dense(A2)
    x, info = ma.gmres_pipelined(op, b2, restart=10, max_iterations=100, tol=1e-10)
    assert info.converged == 1 and np.linalg.norm(A2 @ x - b2) < 1e-8
    op.close()
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(1.0); beta, _ = O.beta_adaptive(k, RADIUS)
    mesh = to_ma_mesh(om)
    A, _ = ma.assemble_tbem(mesh, k, beta)
    b = ma.incident_rhs(om.center, om.normal, k, beta)
    xr, ir = O.gmres_pipelined(b, dense=A, restart=30, max_iterations=10, tol=1e-8)
    for op in (ma.LinearOperator.dense(A), ma.LinearOperator.tbem(ma.BemPlan(mesh), k, beta)):
        x, info = ma.gmres_pipelined(op, b, restart=30, max_iterations=10, tol=1e-8)
        assert info.converged == ir.converged == 1 and info.iterations == ir.iterations and info.restarts == ir.restarts
        assert rel_l2(x, xr) <= 1e-8
        Mp = ma.Preconditioner(op, kind="diagonal")
        xd, idg = ma.gmres_pipelined(op, b, precond=Mp, restart=30, max_iterations=10, tol=1e-8)
        assert idg.converged == 1 and np.linalg.norm(A @ xd - b) <= 1e-6 * np.linalg.norm(b)
        Mp.close(); op.close()
    nodes, rp, ci, K, M = fem.helmholtz_box(6, 5, 4)
    vals = O.helmholtz_values(K, M, 0.4 + 0.05j) + 0.0
    n = len(rp) - 1
    vals = vals.copy(); diag_idx = np.array([np.nonzero(ci[rp[i]:rp[i + 1]] == i)[0][0] + rp[i] for i in range(n)]); vals[diag_idx] += 0.3
    h = ma.CsrOperator(rp, ci, values=vals); opc = ma.LinearOperator.csr(h)
    bb = _xvec(n)
    for pk, pkind in ((None, 0), ("jacobi", 1)):
        Mj = None if pk is None else ma.Preconditioner(h, kind="jacobi", omega=0.8, sweeps=2)
        xo, io = O.gmres_pipelined(bb, csr=(rp, ci, vals), pkind=pkind, omega=0.8, sweeps=2, restart=30, max_iterations=20, tol=1e-9)
        xg, ig = ma.gmres_pipelined(opc, bb, precond=Mj, restart=30, max_iterations=20, tol=1e-9)
        assert ig.converged == io.converged == 1 and abs(ig.iterations - io.iterations) <= 1
        assert rel_l2(xg, xo) <= 1e-7
        if Mj is not None:
            Mj.close()
    opc.close(); h.close()


def test_gmres_gram_schmidt_step_forms_agree_at_every_size_class(gpu, monkeypatch):
    """A Krylov step's modified Gram-Schmidt runs as one launch with 1, 2, 4, 8, 16 or 32 elements of w per thread, and as separate
    kernels above 2^21 unknowns: on a shifted 1-D Laplacian of each size class the iterates of the two forms agree to rounding and the
    iteration counts are equal (12 iterations: the basis is then 13 vectors deep)."""
    import scipy.sparse as sp
    for n in (3000, 100000, 200000, 500000, 1000000, 1500000, 2200000):
        main = np.full(n, 2.5 + 0.1j); off = np.full(n - 1, -1.0 + 0j)
        A = sp.diags([off, main, off], [-1, 0, 1], format="csr")
        op = ma.CsrOperator(A.indptr.astype(np.int64), A.indices.astype(np.int64), values=A.data.astype(np.complex128))
        lin = ma.LinearOperator.csr(op)
        i = np.arange(n); b = (np.sin(0.01 * i) + 1j * np.cos(0.02 * i)).astype(np.complex128)
        res = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("MA_GMRES_FUSED_MGS", mode)
            res[mode] = ma.gmres(lin, b, restart=12, max_iterations=1, tol=1e-14)
        (x1, i1), (x0, i0) = res["1"], res["0"]
        assert i1.iterations == i0.iterations == 12
        assert np.abs(x1 - x0).max() <= 1e-12 * np.abs(x0).max(), n
        assert np.linalg.norm(A @ x1 - b) < 0.1 * np.linalg.norm(b)
        lin.close(); op.close()
