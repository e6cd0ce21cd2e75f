"""The FEM solver dispatcher (math-fem/src/solver/mod.rs) through `math_audio_amd.fem_solver`: the reference's own tests
(:1513-1737) restated on the matrices this package can assemble (P1 tetrahedra on a box instead of P1 triangles on the unit square:
FEM assembly is outside the package), every SolverType against a dense solve of the same system, and the error behaviour."""
import numpy as np
import pytest
import scipy.sparse as sp
import math_audio_amd as ma
from math_audio_amd import fem_solver as fs

pytestmark = pytest.mark.gpu


def _problem(n=4, k=1.0, source=None):
    return fs.HelmholtzProblem.box(n, n, n, k, source=source, lx=1.0, ly=1.0, lz=1.0)


def _dense_solution(p):
    nd = p.num_dofs()
    A = sp.csr_matrix((p.stiffness - (p.k * p.k) * p.mass, p.col_indices, p.row_ptrs), shape=(nd, nd)).toarray()
    return np.linalg.solve(A, p.rhs), A


def test_solve_helmholtz_direct(gpu):                       # :1514-1530
    p = _problem()
    s = fs.solve(p, fs.SolverConfig(solver_type=fs.SolverType.Direct))
    assert s.converged and len(s.values) == p.num_dofs() and s.iterations == 0
    x, A = _dense_solution(p)
    assert np.abs(s.values - x).max() <= 1e-10 * np.abs(x).max() and s.residual <= 1e-12


ITERATIVE = [fs.SolverType.Gmres, fs.SolverType.GmresIlu, fs.SolverType.GmresJacobi, fs.SolverType.GmresIluColoring, fs.SolverType.GmresIluFixedPoint, fs.SolverType.GmresSchwarz, fs.SolverType.GmresAmg,
             fs.SolverType.GmresPipelined, fs.SolverType.GmresPipelinedIlu, fs.SolverType.GmresPipelinedAmg, fs.SolverType.GmresShiftedLaplacian]


@pytest.mark.parametrize("solver_type", ITERATIVE, ids=[t.name for t in ITERATIVE])
def test_solve_helmholtz_iterative(gpu, solver_type):      # :1533-1578, :1646-1717 (max_iterations 100, restart 20, tolerance 1e-8)
    p = _problem()
    cfg = fs.SolverConfig(solver_type=solver_type, gmres=fs.GmresConfig(100, 20, 1e-8), wavenumber=1.0)
    s = fs.solve(p, cfg)
    assert s.converged and len(s.values) == p.num_dofs()
    x, A = _dense_solution(p)
    # left preconditioning: the tolerance is on M^-1 (b - A x), the true residual may sit above it
    assert np.linalg.norm(A @ s.values - p.rhs) <= 1e-5 * np.linalg.norm(p.rhs)
    assert np.abs(s.values - x).max() <= 1e-5 * np.abs(x).max()


def test_ilu_preconditioner_improves_convergence(gpu):      # :1601-1643
    p = _problem(8, 2.0, source=lambda x, y, z: np.sin(np.pi * x) * np.sin(np.pi * y))
    g = fs.GmresConfig(500, 30, 1e-8)
    plain = fs.solve(p, fs.SolverConfig(solver_type=fs.SolverType.Gmres, gmres=g))
    ilu = fs.solve(p, fs.SolverConfig(solver_type=fs.SolverType.GmresIlu, gmres=g))
    assert ilu.iterations <= plain.iterations + 10
    assert ilu.converged and plain.converged


def test_shifted_laplacian_config_constructors(gpu):        # :1720-1736
    d = fs.ShiftedLaplacianConfig()
    assert d.alpha == 1.0 and d.beta == 1.0
    c = fs.ShiftedLaplacianConfig.for_wavenumber(2.0)
    assert c.alpha == 2.0 and c.beta == 1.0
    a = fs.ShiftedLaplacianConfig.aggressive(2.0)
    assert a.alpha == 4.0 and a.beta == 2.0
    c = fs.ShiftedLaplacianConfig.conservative(2.0)
    assert c.alpha == 1.0 and c.beta == 0.5
    cfg = fs.SolverConfig()
    assert cfg.solver_type == fs.SolverType.GmresIlu and cfg.gmres.max_iterations == 1000 and cfg.gmres.restart == 50 and cfg.gmres.tolerance == 1e-10
    assert cfg.schwarz_subdomains == 8 and cfg.schwarz_overlap == 2 and cfg.shifted_laplacian is None and cfg.wavenumber is None


def test_shifted_laplacian_matrix_and_mg_variant(gpu):
    """build_shifted_laplacian (:1161-1208): P = K + (alpha + i beta) M on the shared pattern; GmresShiftedLaplacianMg (:1293-1350) as
    written: mg_cycles restarts of GMRES on P itself, converged = true, iterations = mg_cycles, residual = sqrt(sum |b - A x|)."""
    p = _problem(4, 1.0)
    rp, ci, v = fs.build_shifted_laplacian(p, 0.5, 0.5)
    nd = p.num_dofs()
    P = sp.csr_matrix((v, ci, rp), shape=(nd, nd)).toarray()
    K = sp.csr_matrix((p.stiffness, p.col_indices, p.row_ptrs), shape=(nd, nd)).toarray(); M = sp.csr_matrix((p.mass, p.col_indices, p.row_ptrs), shape=(nd, nd)).toarray()
    assert np.abs(P - (K + (0.5 + 0.5j) * M)).max() <= 1e-15
    s = fs.solve(p, fs.SolverConfig(solver_type=fs.SolverType.GmresShiftedLaplacianMg, gmres=fs.GmresConfig(100, 20, 1e-8), wavenumber=1.0))
    assert s.converged and s.iterations == 2 and len(s.values) == nd and np.isfinite(s.values).all()
    x, A = _dense_solution(p)
    assert abs(s.residual - np.sqrt(np.abs(p.rhs - A @ s.values).sum())) <= 1e-9 * max(1.0, s.residual)
    # what it solves is P y = b (the loop never looks at A): y is P^-1 b to the GMRES tolerance
    y = np.linalg.solve(P, p.rhs)
    assert np.abs(s.values - y).max() <= 1e-5 * np.abs(y).max()


def test_solve_csr_and_its_errors(gpu):                     # :1438-1503
    p = _problem(5, 1.5)
    nd = p.num_dofs()
    vals = (p.stiffness - (p.k * p.k) * p.mass).astype(np.complex128)
    x, A = _dense_solution(p)
    g = fs.GmresConfig(300, 30, 1e-9)
    for t in (fs.SolverType.Direct, fs.SolverType.Gmres, fs.SolverType.GmresIlu, fs.SolverType.GmresJacobi, fs.SolverType.GmresAmg, fs.SolverType.GmresPipelinedAmg):
        s = fs.solve_csr(p.row_ptrs, p.col_indices, vals, p.rhs, fs.SolverConfig(solver_type=t, gmres=g))
        assert s.converged and np.abs(s.values - x).max() <= 1e-6 * np.abs(x).max(), t.name
    # a warm start from the solution converges at once
    s = fs.solve_csr_with_guess(p.row_ptrs, p.col_indices, vals, p.rhs, x, fs.SolverConfig(solver_type=fs.SolverType.GmresIlu, gmres=g))
    assert s.converged and s.iterations <= 1
    with pytest.raises(fs.SolverError) as e:
        fs.solve_csr(p.row_ptrs, p.col_indices, vals, p.rhs[:-1], fs.SolverConfig())
    assert e.value.kind == "DimensionMismatch" and e.value.expected == nd and e.value.actual == nd - 1
    with pytest.raises(fs.SolverError) as e:
        fs.solve_csr_with_guess(p.row_ptrs, p.col_indices, vals, p.rhs, x[:-2], fs.SolverConfig())
    assert e.value.kind == "DimensionMismatch" and e.value.actual == nd - 2
    for t in (fs.SolverType.GmresShiftedLaplacian, fs.SolverType.GmresShiftedLaplacianMg):
        with pytest.raises(fs.SolverError) as e:
            fs.solve_csr(p.row_ptrs, p.col_indices, vals, p.rhs, fs.SolverConfig(solver_type=t))
        assert e.value.kind == "InvalidConfiguration"
    with pytest.raises(fs.SolverError) as e:                 # two iterations cannot reach 1e-14 on this system
        fs.solve_csr(p.row_ptrs, p.col_indices, vals, p.rhs, fs.SolverConfig(solver_type=fs.SolverType.Gmres, gmres=fs.GmresConfig(2, 2, 1e-14)))
    assert e.value.kind == "ConvergenceFailure" and e.value.iterations >= 1 and e.value.residual > 0.0
    # a singular matrix through Direct
    sing = np.zeros_like(vals)
    with pytest.raises(fs.SolverError) as e:
        fs.solve_csr(p.row_ptrs, p.col_indices, sing, p.rhs, fs.SolverConfig(solver_type=fs.SolverType.Direct))
    assert e.value.kind == "SingularMatrix"
