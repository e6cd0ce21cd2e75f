"""The reference's integration tests (math-bem/tests/test_accuracy_parity.rs, tests/test_bem_sphere_integration.rs)
run through the CPU restatement: every threshold those files hold must be met by the oracle, which pins the
oracle's UV-sphere generator + TBEM assembly + LU + compute_total_field chain on reference-held numbers
(tests/reference_cases.py lists the cases with their file:line). The device path runs the same cases in
tests/test_reference_integration_gpu.py."""
import numpy as np
import pytest
import oracle_lib as O
import reference_cases as RC


class OracleBackend:
    def solve(self, n_theta, n_phi, k, beta):
        om = O.uv_sphere(RC.RADIUS, n_theta, n_phi)
        A, r0 = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
        rhs = r0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
        x, _, rc = O.zgesv(A, rhs, nthreads=4)
        assert rc == 0

        def total_field(points):
            return O.incident_pressure(points, k) + O.compute_scattered_field(points, om, x, k)
        return om.center, x, total_field


def oracle_mie(k, radius, terms, r, thetas):
    return O.sphere_scattering_3d(k, radius, terms, [r], list(thetas))[0]


@pytest.mark.parametrize("case", RC.CASES, ids=[c["name"] for c in RC.CASES])
def test_reference_threshold_met_by_the_restatement(case):
    err = RC.run_case(case, OracleBackend(), oracle_mie)
    assert np.isfinite(err)
    assert err < case["limit"], (case["name"], err, case["limit"])


# ---- math-bem/tests/test_fmm_validation.rs through the restatements (tests/fmm_validation_cases.py)
class OracleFmmBackend:
    def tbem_matrix(self, mesh, k):
        A, _ = O.build_tbem_system_with_beta(mesh, k, complex(0.0, 1.0 / k), nthreads=4)       # build_tbem_system: physics.burton_miller_beta() = i / k
        return A

    def slfmm_one_cluster(self, mesh, k):
        from fmm_clusters import Clusters
        n = mesh.n_elem
        one = Clusters([[0.0, 0.0, 0.0]], [0, n], np.arange(n), [0, 0], [], [0, 0], [])
        S = O.Slfmm(mesh, one, k, 4, 8, 5)
        return lambda x: S.matvec(np.asarray(x, dtype=complex))

    def mlfmm(self, mesh, target, k):
        M = O.mlfmm_module()
        S = M.MlfmmSystem(mesh, M.build_cluster_tree(mesh.center, target, k), k, O)
        return S.matvec

    def gmres(self, A, b, restart, max_iterations, tol):
        x, info = O.gmres(b, dense=A, restart=restart, max_iterations=max_iterations, tol=tol)
        return x, info.iterations, info.restarts, bool(info.converged)

    def cgs(self, A, b, max_iterations, tol):
        x, it, res, conv = O.krylov_module().cgs(lambda v: A @ v, b, max_iterations, tol)
        return x, it, conv


def test_fmm_validation_thresholds_met_by_the_restatement():
    import fmm_validation_cases as F
    mesh = O.icosphere(F.RADIUS, 1)
    rel = F.check_slfmm_matvec_vs_tbem(OracleFmmBackend(), mesh)
    assert rel < 0.5
    F.check_mlfmm_matvec_nonzero(OracleFmmBackend(), mesh)
    F.check_solvers_with_operator(OracleFmmBackend())


# ---- analytic checks that do not rest on a restatement (tests/analytic_cases.py)
class OracleAnalyticBackend:
    def solve(self, mesh, k, beta):
        A, r0 = O.build_tbem_system_with_beta(mesh, k, beta, nthreads=8)
        x, _, rc = O.zgesv(A, r0 + O.compute_rhs_with_beta(mesh.center, mesh.normal, k, beta), nthreads=4)
        assert rc == 0
        return x

    def scattered(self, mesh, k, points, ps, vs):
        return O.compute_scattered_field(points, mesh, ps, k, surface_velocity=vs)


def cube_sphere(radius, m):
    """6 m^2 Quad4 panels: a cube's face grids projected onto the sphere."""
    idx = {}; nodes = []; conn = []

    def nid(p):
        key = tuple(np.round(p, 12))
        if key not in idx:
            idx[key] = len(nodes); nodes.append(p)
        return idx[key]
    g = np.linspace(-1.0, 1.0, m + 1)
    for axis in range(3):
        for sgn in (-1.0, 1.0):
            for i in range(m):
                for j in range(m):
                    q = []
                    for (u, v) in ((g[i], g[j]), (g[i + 1], g[j]), (g[i + 1], g[j + 1]), (g[i], g[j + 1])):
                        p = np.zeros(3); p[axis] = sgn; p[(axis + 1) % 3] = u; p[(axis + 2) % 3] = v
                        q.append(nid(radius * p / np.linalg.norm(p)))
                    conn.append(q if sgn > 0 else q[::-1])
    return O.Mesh(np.array(nodes), np.array(conn, dtype=np.int32))


def test_soft_sphere_against_the_analytic_series():
    """Pressure-type panels (tbem.rs:234-244, the unknown is dp/dn): the total field at r = 2a within 25 % of the sound-soft sphere's
    series at ka = 0.5, 1, 2 on the 1280-panel icosphere, the unknown within 10 % of the analytic dp/dn on average."""
    import analytic_cases as AC
    om = O.icosphere(AC.RADIUS, 3)
    om.bc_type[:] = 1
    for ka in (0.5, 1.0, 2.0):
        err, ratio = AC.soft_sphere_errors(OracleAnalyticBackend(), om, ka, O.incident_pressure)
        assert err < 0.25 and abs(ratio - 1.0) < 0.12, (ka, err, ratio)


def test_rigid_sphere_against_the_true_series_below_the_sign_switch():
    """ka < 0.5 (tbem.rs:108-123 keeps sign = +1 there): the surface solution against the rigid sphere's TRUE series (SciPy's Bessel
    functions, nothing restated): 0.3-0.7 % on the all-Quad4 sphere (384 panels), 0.9-1.5 % on the Tri3 icosphere (320 panels) at
    ka = 0.2, 0.3, 0.45. The series the reference's own tests use is farther from these solutions than the truth is at ka = 0.45
    (3.6 % against 0.65 %): its n = 0 term takes y_{-1}(x) = -sin(x) / x (solutions_3d.rs:167-173)."""
    import analytic_cases as AC
    quad = cube_sphere(AC.RADIUS, 8); tri = O.icosphere(AC.RADIUS, 2)
    assert np.all(quad.conn[:, 3] >= 0)
    for ka in (0.2, 0.3, 0.45):
        eq_ref, eq_true = AC.rigid_surface_error(OracleAnalyticBackend(), quad, ka)
        et_ref, et_true = AC.rigid_surface_error(OracleAnalyticBackend(), tri, ka)
        assert eq_true < 0.01 and et_true < 0.02, (ka, eq_true, et_true)
    assert eq_ref > 3.0 * eq_true


def test_quad4_and_tri3_agree_above_the_sign_switch():
    """ka >= 0.5: the reference flips the sign of the double-layer term (tbem.rs:108-123, :203) and its solutions leave the true series
    (50 % at ka = 1, more than 100 % at ka = 0.6) while staying 26-27 % from the series its own tests use (their threshold at ka = 1:
    30 %, test_accuracy_parity.rs:151-254). Restated, not repaired; what can be checked is that the Quad4 path and the Tri3 path
    give the same answer: within 0.02 of each other in relative L2 distance from either series."""
    import analytic_cases as AC
    quad = cube_sphere(AC.RADIUS, 8); tri = O.icosphere(AC.RADIUS, 2)
    for ka, lim in ((0.5, 0.08), (1.0, 0.30)):
        eq, eq_true = AC.rigid_surface_error(OracleAnalyticBackend(), quad, ka)
        et, et_true = AC.rigid_surface_error(OracleAnalyticBackend(), tri, ka)
        assert eq < lim and et < lim and abs(eq - et) < 0.02 and abs(eq_true - et_true) < 0.03, (ka, eq, et, eq_true, et_true)
