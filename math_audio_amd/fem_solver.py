"""Host-side mirror of the reference's FEM solver dispatcher (math-fem/src/solver/mod.rs): SolverType / SolverConfig /
ShiftedLaplacianConfig / Solution / SolverError and `solve`, `solve_csr`, `solve_csr_with_guess`.

The orchestration is the reference's, branch by branch (`solve`, :223-276; `solve_csr_with_guess`, :1456-1503); every numeric step runs
on the GPU through the C-ABI: the CSR operator with its fused K - k^2 M values, GMRES and pipelined GMRES, the ILU(0) / Jacobi / AMG
preconditioners (their setup on the host inside the library, as in the reference), the dense LU of `Direct`. There is no CPU fallback:
without the library or a GPU every call raises. All thirteen solver types run; `GmresIluColoring` is ILU(0) with level-scheduled solves in the
reference (ilu_parallel.rs:52-148: "same as sequential" factorisation) -- which is what the device ILU(0) apply is."""
import enum
import numpy as np
import math_audio_amd as ma
from . import fem


class SolverType(enum.Enum):                    # mod.rs:72-107
    Direct = 0
    Gmres = 1
    GmresIlu = 2
    GmresJacobi = 3
    GmresIluColoring = 4
    GmresIluFixedPoint = 5
    GmresSchwarz = 6
    GmresAmg = 7
    GmresPipelined = 8
    GmresPipelinedIlu = 9
    GmresPipelinedAmg = 10
    GmresShiftedLaplacian = 11
    GmresShiftedLaplacianMg = 12


class GmresConfig:                              # the values of SolverConfig::default (:52-58)
    def __init__(self, max_iterations=1000, restart=50, tolerance=1e-10, print_interval=0):
        self.max_iterations, self.restart, self.tolerance, self.print_interval = int(max_iterations), int(restart), float(tolerance), int(print_interval)


class ShiftedLaplacianConfig:                   # :110-185
    def __init__(self, alpha=1.0, beta=1.0, mg_cycles=2, amg_levels=0, omega=0.8, presmooth=2, postsmooth=2):
        self.alpha, self.beta, self.mg_cycles, self.amg_levels = float(alpha), float(beta), int(mg_cycles), int(amg_levels)
        self.omega, self.presmooth, self.postsmooth = float(omega), int(presmooth), int(postsmooth)

    @staticmethod
    def for_wavenumber(k):
        return ShiftedLaplacianConfig(0.5 * k * k, 0.5 * k, 2, 0, 0.8, 2, 2)

    @staticmethod
    def aggressive(k):
        return ShiftedLaplacianConfig(k * k, k, 3, 0, 0.7, 3, 3)

    @staticmethod
    def conservative(k):
        return ShiftedLaplacianConfig(0.25 * k * k, 0.25 * k, 1, 0, 0.9, 1, 1)


class SolverConfig:                             # :33-66
    def __init__(self, solver_type=SolverType.GmresIlu, gmres=None, verbosity=0, schwarz_subdomains=8, schwarz_overlap=2, shifted_laplacian=None,
                 wavenumber=None):
        self.solver_type, self.gmres, self.verbosity = solver_type, gmres if gmres is not None else GmresConfig(), int(verbosity)
        self.schwarz_subdomains, self.schwarz_overlap = int(schwarz_subdomains), int(schwarz_overlap)
        self.shifted_laplacian, self.wavenumber = shifted_laplacian, wavenumber


class Solution:                                 # :188-199
    def __init__(self, values, iterations, residual, converged):
        self.values, self.iterations, self.residual, self.converged = values, int(iterations), float(residual), bool(converged)


class SolverError(RuntimeError):                # :202-213
    def __init__(self, kind, text, **fields):
        super().__init__(text)
        self.kind = kind
        self.__dict__.update(fields)

    @staticmethod
    def convergence_failure(iterations, residual):
        return SolverError("ConvergenceFailure", "Solver failed to converge after %d iterations (residual: %g)" % (iterations, residual),
                           iterations=iterations, residual=residual)

    @staticmethod
    def dimension_mismatch(expected, actual):
        return SolverError("DimensionMismatch", "Matrix dimension mismatch: expected %d, got %d" % (expected, actual), expected=expected, actual=actual)


class HelmholtzProblem:
    """What `solve` reads of assembly::HelmholtzProblem: stiffness K and mass M on one CSR pattern, the wavenumber of the system matrix
    A = K - k^2 M, the right-hand side. (The reference keeps K and M as triplets and `matrix.to_csr()` merges them; FEM assembly itself
    is outside this package, SURVEY 2c -- `box` builds the F1M family of BASELINE.json configs[3] for tests and benches.)"""

    def __init__(self, row_ptrs, col_indices, stiffness, mass, k, rhs):
        self.row_ptrs = np.ascontiguousarray(row_ptrs, dtype=np.int64); self.col_indices = np.ascontiguousarray(col_indices, dtype=np.int64)
        self.stiffness = np.ascontiguousarray(stiffness, dtype=np.float64); self.mass = np.ascontiguousarray(mass, dtype=np.float64)
        self.k = complex(k); self.rhs = np.ascontiguousarray(rhs, dtype=np.complex128)

    def num_dofs(self):
        return len(self.row_ptrs) - 1

    @staticmethod
    def box(nx, ny, nz, k, source=None, lx=5.0, ly=4.0, lz=2.5):
        """P1 tetrahedra on a box; rhs = M f with f = source(x, y, z) at the nodes (1 if None): the consistent load of a nodal field."""
        nodes, rp, ci, K, M = fem.helmholtz_box(nx, ny, nz, lx, ly, lz)
        n = len(rp) - 1
        f = np.ones(n, dtype=np.complex128) if source is None else np.array([source(*p) for p in nodes], dtype=np.complex128)
        rhs = np.zeros(n, dtype=np.complex128)
        rows = np.repeat(np.arange(n), np.diff(rp))
        np.add.at(rhs, rows, M * f[ci])
        return HelmholtzProblem(rp, ci, K, M, k, rhs)


def _fail_unless_converged(x, info):
    if not info.converged:
        raise SolverError.convergence_failure(info.iterations, info.residual)
    return Solution(x, info.iterations, info.residual, info.converged)


def _amg(op, l1):
    """AmgConfig::for_parallel(), with L1Jacobi in `solve`'s branches (:681-684, :848-851) and as it stands in the with_guess ones (:1060)."""
    return ma.AmgFromCsr(op, ma.AmgConfig.preset("for_parallel", smoother=1) if l1 else ma.AmgConfig.preset("for_parallel"))


def _dense(op_arrays, n):
    rp, ci, v = op_arrays
    A = np.zeros((n, n), dtype=np.complex128)
    rows = np.repeat(np.arange(n), np.diff(rp))
    np.add.at(A, (rows, ci), v)
    return A


def _dispatch(op, lin, arrays, rhs, x0, config, from_solve):
    """The branches shared by `solve` (:245-263) and `solve_csr_with_guess` (:1479-1502)."""
    t, g = config.solver_type, config.gmres
    n = len(rhs)
    if t == SolverType.Direct:                  # solve_direct, :279-307: dense LU, residual = mean |A x - b|
        try:
            x = ma.lu_solve(_dense(arrays, n), rhs)
        except ma.MaError:
            raise SolverError("SingularMatrix", "Direct solver failed: singular matrix")
        return Solution(x, 0, float(np.abs(op.matvec(x) - rhs).sum() / n), True)
    if t == SolverType.Gmres:                   # :310-351, :898-921
        return _fail_unless_converged(*ma.gmres(lin, rhs, x0=x0, restart=g.restart, max_iterations=g.max_iterations, tol=g.tolerance))
    pipelined = t in (SolverType.GmresPipelined, SolverType.GmresPipelinedIlu, SolverType.GmresPipelinedAmg)
    if t in (SolverType.GmresIlu, SolverType.GmresIluColoring, SolverType.GmresPipelinedIlu):      # :359-416, :469-529, :775-831
        pre = ma.IluPreconditioner(op)
    elif t == SolverType.GmresIluFixedPoint:    # IluFixedPointPreconditioner::from_csr(csr, FP_ITERATIONS = 10), :532-592, :999-1022
        pre = ma.IluFixedPointPreconditioner(op, 10)
    elif t == SolverType.GmresSchwarz:          # AdditiveSchwarzPreconditioner::from_csr(csr, schwarz_subdomains, schwarz_overlap), :595-664
        pre = ma.AdditiveSchwarzPreconditioner(op, config.schwarz_subdomains, config.schwarz_overlap)
    elif t == SolverType.GmresJacobi:           # DiagonalPreconditioner::from_csr, :419-466
        pre = ma.Preconditioner(op, "jacobi", omega=1.0, sweeps=1)
    elif t in (SolverType.GmresAmg, SolverType.GmresPipelinedAmg):                                 # :667-727, :834-895, :1054-1077, :1130-1153
        pre = _amg(op, l1=from_solve)
    elif t == SolverType.GmresPipelined:        # IdentityPreconditioner, :730-772
        pre = None
    else:
        raise SolverError("InvalidConfiguration", "Invalid solver configuration: Shifted-Laplacian solver requires HelmholtzProblem, not CSR matrix. Use solve() instead.")
    try:
        if pipelined:
            x, info = ma.gmres_pipelined(lin, rhs, precond=pre, x0=x0, restart=g.restart, max_iterations=g.max_iterations, tol=g.tolerance)
        else:
            x, info = ma.gmres_preconditioned(lin, pre, rhs, x0=x0, restart=g.restart, max_iterations=g.max_iterations, tol=g.tolerance)
    finally:
        if pre is not None:
            pre.close()
    return _fail_unless_converged(x, info)


def build_shifted_laplacian(problem, alpha, beta):
    """build_shifted_laplacian (:1161-1208): P = K + (alpha + i beta) M entry by entry, entries of norm <= 1e-15 dropped; K and M share
    the pattern here. Returns (row_ptrs, col_indices, values)."""
    v = problem.stiffness.astype(np.complex128) + complex(alpha, beta) * problem.mass.astype(np.complex128)
    keep = np.sqrt(v.real * v.real + v.imag * v.imag) > 1e-15
    rows = np.repeat(np.arange(problem.num_dofs()), np.diff(problem.row_ptrs))[keep]
    rp = np.zeros(problem.num_dofs() + 1, dtype=np.int64)
    np.add.at(rp, rows + 1, 1)
    return np.cumsum(rp), problem.col_indices[keep], v[keep]


def solve(problem, config):
    """solve(problem, config) (:223-276)."""
    n = problem.num_dofs()
    if len(problem.rhs) != n:
        raise SolverError.dimension_mismatch(n, len(problem.rhs))
    op = ma.CsrOperator(problem.row_ptrs, problem.col_indices, K=problem.stiffness, M=problem.mass)
    op.set_wavenumber(problem.k)
    lin = ma.LinearOperator.csr(op)
    g = config.gmres
    try:
        t = config.solver_type
        if t not in (SolverType.GmresShiftedLaplacian, SolverType.GmresShiftedLaplacianMg):
            values = problem.stiffness - (problem.k * problem.k) * problem.mass
            return _dispatch(op, lin, (problem.row_ptrs, problem.col_indices, values), problem.rhs, None, config, True)
        k = config.wavenumber if config.wavenumber is not None else 1.0                             # :1227-1229
        sl = config.shifted_laplacian if config.shifted_laplacian is not None else ShiftedLaplacianConfig.for_wavenumber(k)
        prp, pci, pv = build_shifted_laplacian(problem, sl.alpha, sl.beta)
        pop = ma.CsrOperator(prp, pci, values=pv)
        try:
            if t == SolverType.GmresShiftedLaplacian:                                               # :1221-1290: AMG of P (for_parallel, L1Jacobi, theta 0.5) on A
                pre = ma.AmgFromCsr(pop, ma.AmgConfig.preset("for_parallel", smoother=1, strong_threshold=0.5))
                try:
                    return _fail_unless_converged(*ma.gmres_preconditioned(lin, pre, problem.rhs, restart=g.restart, max_iterations=g.max_iterations, tol=g.tolerance))
                finally:
                    pre.close()
            pre = ma.AmgFromCsr(pop, ma.AmgConfig.preset("for_parallel"))                            # :1293-1350, as written there
            plin = ma.LinearOperator.csr(pop)
            try:
                residual = problem.rhs.copy(); solution = np.zeros(n, dtype=np.complex128)
                for _ in range(sl.mg_cycles):
                    x, info = ma.gmres_preconditioned(plin, pre, residual, x0=solution, restart=g.restart, max_iterations=g.max_iterations, tol=g.tolerance)
                    solution = x
                    if info.converged:
                        break
                    residual = residual - pop.matvec(solution)
                rn = float(np.sqrt(np.abs(problem.rhs - op.matvec(solution)).sum()))               # the square root of the SUM of moduli, :1338-1340
                return Solution(solution, sl.mg_cycles, rn, True)
            finally:
                pre.close(); plin.close()
        finally:
            pop.close()
    finally:
        lin.close(); op.close()


def solve_csr_with_guess(row_ptrs, col_indices, values, rhs, x0, config):
    """solve_csr_with_guess(csr, rhs, x0, config) (:1456-1503)."""
    rhs = np.ascontiguousarray(rhs, dtype=np.complex128)
    n = len(row_ptrs) - 1
    if n != len(rhs):
        raise SolverError.dimension_mismatch(n, len(rhs))
    if x0 is not None and len(x0) != len(rhs):
        raise SolverError.dimension_mismatch(len(rhs), len(x0))
    op = ma.CsrOperator(row_ptrs, col_indices, values=values)
    lin = ma.LinearOperator.csr(op)
    try:
        return _dispatch(op, lin, (np.asarray(row_ptrs), np.asarray(col_indices), np.asarray(values, dtype=np.complex128)), rhs, x0, config, False)
    finally:
        lin.close(); op.close()


def solve_csr(row_ptrs, col_indices, values, rhs, config):
    """solve_csr(csr, rhs, config) (:1438-1444)."""
    return solve_csr_with_guess(row_ptrs, col_indices, values, rhs, None, config)
