"""math-bem/tests/test_fmm_validation.rs restated (the tests that do not need the ILU preconditioner): each function takes a backend
with  tbem_matrix(mesh, k)  (build_tbem_system: beta = physics.burton_miller_beta()),  slfmm_one_cluster(mesh, k) -> matvec,
mlfmm(mesh, target, k) -> matvec,  gmres(A, b, restart, max_iterations, tol) -> (x, iterations, restarts, converged),
cgs(A, b, max_iterations, tol) -> (x, iterations, converged)  and asserts what the reference asserts, with its numbers."""
import numpy as np

RADIUS, FREQ, C0 = 0.1, 500.0, 343.0                       # setup_test_problem (:27-50)


def wave_number():
    return 2.0 * np.pi * FREQ / C0


def check_slfmm_matvec_vs_tbem(backend, mesh):             # :103-147
    k = wave_number()
    n = mesh.n_elem
    A = backend.tbem_matrix(mesh, k)
    assert A.shape == (n, n) and np.abs(A).sum() > 0.0     # test_tbem_system_valid (:54-75)
    mv = backend.slfmm_one_cluster(mesh, k)
    i = np.arange(n)
    x = np.sin(0.1 * i) + 1j * np.cos(0.2 * i)
    y_t = A @ x; y_s = mv(x)
    rel = np.linalg.norm(y_t - y_s) / max(np.linalg.norm(y_t), 1e-15)
    assert rel < 0.5, rel                                  # "SLFMM matvec should approximate TBEM"
    # test_slfmm_operator_matvec (:309-353): non-zero output, linear to 1e-10
    x2 = np.sin(0.3 * i) + 0j
    y2 = mv(x2)
    assert np.linalg.norm(y2) > 0.0
    alpha = 2.0 - 1.0j
    assert np.linalg.norm(mv(alpha * x2) - alpha * y2) < 1e-10
    return rel


def check_mlfmm_matvec_nonzero(backend, mesh):             # :150-210
    k = wave_number()
    n = mesh.n_elem
    mv = backend.mlfmm(mesh, 5, k)
    i = np.arange(n)
    y = mv(np.sin(0.1 * i) + 1j * np.cos(0.2 * i))
    assert len(y) == n and np.linalg.norm(y) > 0.0


def _tri(n, d, lo, up):
    A = np.zeros((n, n), dtype=complex)
    for i in range(n):
        A[i, i] = d
        if i > 0:
            A[i, i - 1] = lo
        if i < n - 1:
            A[i, i + 1] = up
    return A


def check_solvers_with_operator(backend):
    # test_iterative_solver_with_operator (:251-303): CGS, n = 10
    A = _tri(10, 10.0, -1.0 + 0.1j, -1.0 - 0.1j); b = np.sin(0.3 * np.arange(10)) + 0j
    x, it, conv = backend.cgs(A, b, 200, 1e-10)
    assert conv and np.linalg.norm(b - A @ x) / np.linalg.norm(b) < 1e-6
    # test_gmres_with_operator (:538-586): n = 20, GMRES(15)
    A = _tri(20, 10.0, -1.0 + 0.1j, -1.0 - 0.1j); b = np.sin(0.3 * np.arange(20)) + 0j
    x, it, rs, conv = backend.gmres(A, b, 15, 50, 1e-10)
    assert conv and np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-8
    # test_gmres_restart_behavior (:644-700): GMRES(5) and GMRES(50) converge, the larger restart needs no more restarts
    A = _tri(50, 4.0, -1.0, -1.0); b = np.ones(50, dtype=complex)
    xs, its, rss, cs = backend.gmres(A, b, 5, 100, 1e-10)
    xl, itl, rsl, cl = backend.gmres(A, b, 50, 100, 1e-10)
    assert cs and cl and rsl <= rss
    # test_gmres_vs_cgs_convergence (:714-785): same solution to 1e-6
    A = _tri(20, 10.0, -1.0, -1.0); b = np.sin(0.25 * np.arange(20)) + 0j
    xg, itg, rg, cg_ = backend.gmres(A, b, 20, 100, 1e-10)
    xc, itc, cc = backend.cgs(A, b, 100, 1e-10)
    assert cg_ and cc and np.linalg.norm(xg - xc) / np.linalg.norm(xg) < 1e-6
    # test_gmres_robustness_vs_cgs (:788-867): GMRES(25) converges on the non-symmetric complex system to 1e-8
    A = _tri(25, 6.0 + 0.3j, -2.0 + 0.1j, -1.5 - 0.1j); b = np.sin(0.25 * np.arange(25)) + 0j
    xg, itg, rg, cg_ = backend.gmres(A, b, 25, 100, 1e-10)
    assert cg_ and np.linalg.norm(A @ xg - b) / np.linalg.norm(b) < 1e-8
