"""GPU parity of the other Krylov solvers (math-solvers/src/iterative/bicgstab.rs, cgs.rs, cg.rs) behind the operator boundary: the
reference's own tests through the C-ABI, and the same iterations as the numpy restatement on a sparse FEM operator, a dense
Hermitian positive definite one and the matrix-free TBEM operator (what BemSolver hands to bicgstab)."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from helpers import to_ma_mesh, k_from_ka, RADIUS

pytestmark = pytest.mark.gpu


def _csr_of(A):
    import scipy.sparse as sp
    M = sp.csr_matrix(A)
    return ma.CsrOperator(M.indptr.astype(np.int64), M.indices.astype(np.int64), values=M.data.astype(np.complex128))


def test_reference_unit_tests_through_the_c_abi(gpu):
    A = np.array([[4.0, 1.0], [1.0, 3.0]], dtype=complex); b = np.array([1.0, 2.0], dtype=complex)
    op = ma.LinearOperator.dense(A)
    for fn in (ma.bicgstab, ma.cgs, ma.cg):                              # bicgstab.rs:190-219, cgs.rs:151-180, cg.rs:146-168
        x, info = fn(op, b, 100, 1e-10)
        assert info.converged and np.linalg.norm(A @ x - b) < 1e-8
        x0, i0 = fn(op, np.zeros(2, dtype=complex), 100, 1e-10)
        assert i0.converged and i0.iterations == 0 and np.all(x0 == 0)
    ident = ma.LinearOperator.dense(np.eye(5, dtype=complex))            # cg.rs:170-189
    bi = np.arange(1, 6, dtype=complex)
    x, info = ma.cg(ident, bi, 10, 1e-12)
    assert info.converged and info.iterations <= 2 and np.linalg.norm(x - bi) < 1e-10


@pytest.mark.parametrize("which", ["bicgstab", "cgs", "cg"])
def test_same_iterations_as_the_restatement(gpu, which):
    K = O.krylov_module()
    rng = np.random.default_rng(11)
    n = 400
    Bm = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    G = Bm @ Bm.conj().T / n + 2.0 * np.eye(n)                          # Hermitian positive definite (cg is only defined there)
    rhs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    xr, itr, resr, convr = getattr(K, which)(lambda v: G @ v, rhs, 300, 1e-9)
    x, info = getattr(ma, which)(ma.LinearOperator.dense(G), rhs, 300, 1e-9)
    assert convr and info.converged and abs(info.iterations - itr) <= 1
    assert np.linalg.norm(x - xr) <= 1e-7 * np.linalg.norm(xr) and np.linalg.norm(G @ x - rhs) <= 1e-8 * np.linalg.norm(rhs)
    # a sparse shifted Laplacian (complex shift: non-Hermitian, bicgstab / cgs only)
    import scipy.sparse as sp
    m = 900
    L = sp.diags([-1.0, 2.0 + 0.3j, -1.0], [-1, 0, 1], shape=(m, m)).tocsr().astype(np.complex128)
    bb = np.sin(0.05 * np.arange(m)) + 0.2j
    if which != "cg":
        csr = ma.CsrOperator(L.indptr.astype(np.int64), L.indices.astype(np.int64), values=L.data)
        xr, itr, resr, convr = getattr(K, which)(lambda v: L @ v, bb, 2000, 1e-8)
        x, info = getattr(ma, which)(ma.LinearOperator.csr(csr), bb, 2000, 1e-8)
        assert convr and info.converged and abs(info.iterations - itr) <= max(2, itr // 20)
        assert np.linalg.norm(L @ x - bb) <= 1e-7 * np.linalg.norm(bb)


def test_bicgstab_on_the_matrix_free_tbem_operator(gpu):
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(1.0)
    beta, _ = O.beta_adaptive(k, RADIUS)
    A, rhs0 = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    rhs = rhs0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
    K = O.krylov_module()
    xr, itr, resr, convr = K.bicgstab(lambda v: A @ v, rhs, 500, 1e-8)
    plan = ma.BemPlan(to_ma_mesh(om))
    op = ma.LinearOperator.tbem(plan, k, beta)
    x, info = ma.bicgstab(op, rhs, 500, 1e-8)
    assert convr and info.converged and abs(info.iterations - itr) <= 2
    assert np.linalg.norm(x - xr) <= 1e-6 * np.linalg.norm(xr)
    assert np.linalg.norm(A @ x - rhs) <= 1e-7 * np.linalg.norm(rhs)
