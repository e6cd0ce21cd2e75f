"""The CPU oracle pinned against the reference's own known answers for this path
and against independent libraries (NumPy/SciPy LAPACK, scipy.special). No GPU needed."""
import os
import numpy as np
import pytest
import scipy.linalg as sl
import scipy.special as sp
import oracle_lib as O
from helpers import k_from_ka, RADIUS, rel_l2

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "bem_golden.npz"))


# ---------------------------------------------------------------- quadrature (gauss.rs:415-442)
@pytest.mark.parametrize("n", [2, 4, 6, 8, 10, 12, 16, 20])
def test_gauss_weights_sum_to_two(n):
    x, w = O.gauss_legendre(n)
    assert len(x) == n and abs(w.sum() - 2.0) < 1e-10


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20])
def test_gauss_tables_are_gauss_legendre(n):
    x, w = O.gauss_legendre(n)
    xr, wr = np.polynomial.legendre.leggauss(n)
    assert np.abs(x - xr).max() < 2e-15 and np.abs(w - wr).max() < 2e-15


def test_gauss_fallback_to_next_table():          # gauss.rs:41-58
    assert len(O.gauss_legendre(9)[0]) == 12 and len(O.gauss_legendre(11)[0]) == 12
    assert len(O.gauss_legendre(13)[0]) == 16 and len(O.gauss_legendre(17)[0]) == 20 and len(O.gauss_legendre(33)[0]) == 20


def test_triangle_rules():                          # gauss.rs:424-432: weights sum to 0.5
    for order, npts in ((1, 1), (2, 4), (3, 7), (4, 13), (7, 13)):
        q = O.triangle_quadrature(order)
        assert q.shape == (npts, 3) and abs(q[:, 2].sum() - 0.5) < 1e-10
    q = O.triangle_quadrature(4)                    # degree-7 rule integrates x^3 y^2 over the unit triangle: 3!2!/7! = 1/420
    assert abs((q[:, 0] ** 3 * q[:, 1] ** 2 * q[:, 2]).sum() - 1.0 / 420.0) < 1e-12


def test_quad_rule():                               # gauss.rs:434-442
    q = O.quad_quadrature(2)
    assert q.shape == (4, 3) and abs(q[:, 2].sum() - 4.0) < 1e-10


# ---------------------------------------------------------------- panel integrals
TRI = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])


def test_planar_self_term_invariants():             # singular.rs:779-815
    k = O.wave_number(10.0, 343.0)
    x = TRI.mean(axis=0); n = np.array([0.0, 0.0, 1.0])
    r = O.singular_integration(x, n, TRI, k)
    assert r[0].real > 0.0                           # Re G > 0
    assert abs(r[1]) < 1e-10 and abs(r[2]) < 1e-10   # |H|, |H^T| vanish in the panel's own plane
    assert np.all(np.isfinite(r.view(np.float64)))


def test_far_field_g_is_small_and_matches_point_kernel():   # regular.rs:537-559 (|G| < 0.1) + closed form
    k = 2.0
    x = np.array([0.3, 0.2, 40.0]); n = np.array([0.0, 0.0, 1.0])
    r = O.regular_integration(x, n, TRI, 0.5, k)
    assert abs(r[0]) < 0.1
    d = np.linalg.norm(TRI.mean(axis=0) - x)
    g = 0.5 * np.exp(1j * k * d) / (4 * np.pi * d)   # area x G(centroid): exact to O((h/d)^2)
    assert abs(r[0] - g) / abs(g) < 5e-3


def test_green_kernel_unit_value():                  # math-wave helmholtz.rs:28-31: |G(r=1,k=2)| = 1/4pi
    tiny = TRI * 1e-4
    x = np.array([0.0, 0.0, 1.0])
    r = O.regular_integration(x, np.array([0.0, 0.0, 1.0]), tiny, 0.5e-8, 2.0)
    assert abs(abs(r[0]) / 0.5e-8 - 1.0 / (4 * np.pi)) < 1e-7


def _leaf_area(subs):
    area = 0.0
    for s in subs:
        t = np.array(list(s.tri)).reshape(3, 2)
        area += 0.5 * abs((t[1, 0] - t[0, 0]) * (t[2, 1] - t[0, 1]) - (t[2, 0] - t[0, 0]) * (t[1, 1] - t[0, 1]))
    return area


def test_subelements_unsplit_when_far_and_split_when_near():    # singular.rs:497-660
    far = O.generate_subelements([0.3, 0.3, 10.0], TRI, 0.5)
    assert len(far) == 1 and far[0].factor == 1.0 and far[0].gauss_order == 4
    near = O.generate_subelements([0.3, 0.3, 0.8], TRI, 0.5)       # two levels, every level fully kept
    assert len(near) == 16 and all(s.factor == 0.25 for s in near)
    assert abs(_leaf_area(near) - 0.5) < 1e-12                     # leaves tile the parent exactly


def test_subelements_level_overflow_quirk():
    """`ndie > 15 => break` (singular.rs:556-562) abandons the rest of a level: for a very near point the
    stored leaves do NOT cover the panel. The restatement (and the device kernel) must reproduce that."""
    very = O.generate_subelements([0.3, 0.3, 0.05], TRI, 0.5)
    assert len(very) == 109
    assert abs(_leaf_area(very) - 0.40625) < 1e-12
    capped = O.generate_subelements([0.33, 0.33, 1e-4], TRI, 0.5)
    assert len(capped) <= 110


def test_subdivided_integral_converges_to_fine_quadrature():
    """A near-singular integral from the adaptive path agrees with brute-force tensor quadrature."""
    k = 5.0
    x = np.array([0.3, 0.25, 0.8]); nx = np.array([0.0, 0.0, 1.0])     # two full levels of subdivision, no level overflow
    got = O.regular_integration(x, nx, TRI, 0.5, k)[0]
    xs, ws = np.polynomial.legendre.leggauss(200)
    u = 0.5 * (xs + 1); wu = 0.5 * ws
    U, V = np.meshgrid(u, u, indexing="ij"); W = np.outer(wu, wu)
    s = U; t = V * (1 - U); jac = (1 - U)              # Duffy map of the unit square onto the triangle
    y = np.stack([s, t, np.zeros_like(s)], axis=-1)
    r = np.linalg.norm(y - x, axis=-1)
    ref = (W * jac * np.exp(1j * k * r) / (4 * np.pi * r)).sum()
    assert abs(got - ref) / abs(ref) < 5e-4           # GAU_ACCU = 5e-4 is the reference's own accuracy target


# ---------------------------------------------------------------- meshes (generators.rs tests)
def test_icosphere_counts_and_radius():
    for sub, (nn, ne) in enumerate([(12, 20), (42, 80), (162, 320), (642, 1280)]):
        m = O.icosphere(RADIUS, sub)
        assert m.nodes.shape[0] == nn and m.n_elem == ne
        assert np.abs(np.linalg.norm(m.nodes, axis=1) - RADIUS).max() < 1e-12
        assert np.all((m.normal * m.center).sum(axis=1) > 0)         # outward
        assert abs(m.area.sum() - 4 * np.pi * RADIUS ** 2) / (4 * np.pi * RADIUS ** 2) < (0.35 if sub == 0 else 0.1)


def test_uv_sphere_counts():
    m = O.uv_sphere(RADIUS, 51, 100)
    assert m.n_elem == 10000 and m.nodes.shape[0] == 5002
    cs = GOLD["s10_checksums"]
    got = np.array([m.nodes.sum(), np.abs(m.nodes).sum(), m.area.sum(), m.center[:, 2].sum(), float(m.conn[:, :3].astype(np.int64).sum()), 10000.0])
    assert np.allclose(got, cs, rtol=1e-13, atol=1e-13)


# ---------------------------------------------------------------- Mie oracle (solutions_3d.rs)
@pytest.mark.parametrize("x", [0.2, 1.0, 3.0, 14.7])
def test_spherical_bessel_against_scipy(x):
    for n in range(0, 40):
        j = O.lib().mao_spherical_bessel_j(n, x); jr = sp.spherical_jn(n, x)
        assert abs(j - jr) <= 1e-12 * max(abs(jr), 1e-300) + 1e-300
        y = O.lib().mao_spherical_bessel_y(n, x); yr = sp.spherical_yn(n, x)
        assert abs(y - yr) <= 1e-10 * abs(yr)
    assert abs(O.lib().mao_legendre_p(7, 0.3) - sp.eval_legendre(7, 0.3)) < 1e-14


def _mie_scipy(k, a, r, theta, terms=50, reference_quirk=True):
    """Rigid-sphere series with scipy.special. reference_quirk=True reproduces solutions_3d.rs:167-173,
    which uses y_{-1}(x) = -sin(x)/x for the n = 0 derivative (the identity is +sin(x)/x), so the
    reference's a_0 is slightly off; the restatement follows the reference, not the textbook."""
    ka = k * a
    n = np.arange(terms)
    jp = sp.spherical_jn(n, ka, derivative=True)
    yp = sp.spherical_yn(n, ka, derivative=True)
    if reference_quirk:
        yp = yp.copy(); yp[0] = -np.sin(ka) / ka - (1.0 / ka) * sp.spherical_yn(0, ka)
    an = jp / (jp + 1j * yp)
    hn = sp.spherical_jn(n, k * r) + 1j * sp.spherical_yn(n, k * r)
    t = (2 * n + 1) * (1j ** n) * (sp.spherical_jn(n, k * r) - an * hn) * sp.eval_legendre(n, np.cos(theta))
    return t[np.isfinite(t)].sum()


@pytest.mark.parametrize("ka", [0.2, 1.0, 3.0])
def test_mie_series_against_scipy(ka):
    k = ka / RADIUS
    theta = np.linspace(0, np.pi, 9)
    got = O.sphere_scattering_3d(k, RADIUS, 50, [RADIUS, 2 * RADIUS], theta)
    for ir, r in enumerate([RADIUS, 2 * RADIUS]):
        for it, th in enumerate(theta):
            ref = _mie_scipy(k, RADIUS, r, th)
            assert abs(got[ir, it] - ref) <= 1e-12 * max(1.0, abs(ref))


def test_mie_reference_quirk_is_small():
    """How far the reference's series is from the textbook one (only the n = 0 term differs)."""
    for ka, bound in ((0.2, 2e-3), (1.0, 1.0), (3.0, 1.0)):     # ~1e-3 at ka=0.2, 0.84 at ka=1: the reference "analytical" series is off in a_0
        k = ka / RADIUS
        d = abs(_mie_scipy(k, RADIUS, RADIUS, 0.7) - _mie_scipy(k, RADIUS, RADIUS, 0.7, reference_quirk=False))
        assert 0.0 < d < bound


def test_textbook_series_has_zero_radial_velocity():
    """Sanity of the scipy comparison series itself (rigid wall: dp/dr = 0 on the surface)."""
    ka = 1.0; k = ka / RADIUS; h = 1e-6
    for th in (0.3, 1.2, 2.5):
        p1 = _mie_scipy(k, RADIUS, RADIUS * (1 + h), th, reference_quirk=False)
        p0 = _mie_scipy(k, RADIUS, RADIUS, th, reference_quirk=False)
        assert abs((p1 - p0) / (RADIUS * h)) < 1e-3 * k


# ---------------------------------------------------------------- QA-suite acceptance (bin/qa_suite.rs:175-179, 199-326)
@pytest.mark.parametrize("ka,sub,tol", [(0.2, 2, 0.05), (1.0, 3, 0.30), (3.0, 3, 0.30)])
def test_qa_suite_scattering_thresholds(ka, sub, tol):
    om = O.icosphere(RADIUS, sub)
    k = k_from_ka(ka)
    beta, _ = O.beta_adaptive(k, RADIUS)
    A, rhs0 = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    rhs = rhs0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
    x, _, rc = O.zgesv(A, rhs, nthreads=8)
    assert rc == 0
    r = np.linalg.norm(om.center, axis=1); theta = np.arccos(om.center[:, 2] / r)
    mie = np.array([O.sphere_scattering_3d(k, RADIUS, 50, [r[i]], [theta[i]])[0, 0] for i in range(om.n_elem)])
    assert rel_l2(x, mie) < tol
    assert np.all(np.abs(np.diag(A)) > 1e-15)           # tbem.rs:585-598


def test_threaded_assembly_is_bitwise_sequential():
    om = O.icosphere(RADIUS, 1)
    k = k_from_ka(1.0); beta, _ = O.beta_adaptive(k, RADIUS)
    A1, _ = O.build_tbem_system_with_beta(om, k, beta, nthreads=1)
    A8, _ = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    assert np.array_equal(A1, A8)


def test_golden_system_regression():
    om = O.icosphere(RADIUS, 1)
    assert np.array_equal(om.nodes, GOLD["ico1_nodes"]) and np.array_equal(om.conn, GOLD["ico1_conn"])
    for tag in ("ka1", "ka02"):
        k = float(GOLD["ico1_%s_k" % tag][0]); beta = complex(GOLD["ico1_%s_beta" % tag][0])
        A, rhs0 = O.build_tbem_system_with_beta(om, k, beta)
        rhs = rhs0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
        assert np.allclose(A, GOLD["ico1_%s_A" % tag], rtol=1e-13, atol=1e-15)
        assert np.allclose(rhs, GOLD["ico1_%s_rhs" % tag], rtol=1e-13, atol=1e-15)
        x, _, rc = O.zgesv(A, rhs)
        assert rc == 0 and rel_l2(x, GOLD["ico1_%s_x" % tag]) < 1e-12


def test_sign_switch_and_beta_tiers():                  # tbem.rs:108-123, types.rs:173-195
    assert O.beta_adaptive(2.0, 0.1)[1] == 1.0 and O.beta_adaptive(10.0, 0.1)[1] == 4.0
    assert O.beta_adaptive(15.0, 0.1)[1] == 8.0 and O.beta_adaptive(30.0, 0.1)[1] == 16.0
    om = O.icosphere(RADIUS, 1)
    i, j = 0, 40                                               # a far pair: coefficient = sign*K' + beta*E
    r1 = O.regular_integration(om.center[i], om.normal[i], om.coords(j), om.area[j], 4.9)
    r2 = O.regular_integration(om.center[i], om.normal[i], om.coords(j), om.area[j], 5.5)
    lo, _ = O.build_tbem_system_with_beta(om, 4.9, 0.2j)      # k * mean|c| < 0.5 -> +K'
    hi, _ = O.build_tbem_system_with_beta(om, 5.5, 0.2j)      # k * mean|c| = 0.518 >= 0.5 -> -K
    assert abs(lo[i, j] - (r1[1] + 0.2j * r1[3])) < 1e-15 * abs(lo[i, j]) + 1e-18
    assert abs(hi[i, j] - (-r2[1] + 0.2j * r2[3])) < 1e-15 * abs(hi[i, j]) + 1e-18


# ---------------------------------------------------------------- dense solve (lu.rs:163-240)
def test_lu_known_answers():
    x, _, rc = O.zgesv(np.array([[4.0, 1.0], [1.0, 3.0]]), np.array([1.0, 2.0]))
    assert rc == 0 and np.abs(np.array([[4.0, 1.0], [1.0, 3.0]]) @ x - [1.0, 2.0]).max() < 1e-10
    Ac = np.array([[4 + 1j, 1], [1, 3 - 1j]]); bc = np.array([1 + 1j, 2 - 1j])
    x, _, rc = O.zgesv(Ac, bc)
    assert rc == 0 and np.abs(Ac @ x - bc).max() < 1e-10
    x, _, rc = O.zgesv(np.eye(5), np.arange(1.0, 6.0))
    assert rc == 0 and np.abs(x - np.arange(1.0, 6.0)).max() < 1e-10
    _, _, rc = O.zgesv(np.array([[1.0, 2.0], [2.0, 4.0]]), np.array([1.0, 2.0]))
    assert rc == 1                                       # singular -> LuError::SingularMatrix
    A3 = np.array([[4.0, 1.0, 0.0], [1.0, 3.0, 1.0], [0.0, 1.0, 2.0]])
    for b in ([1.0, 2.0, 3.0], [4.0, 5.0, 6.0]):
        x, _, rc = O.zgesv(A3, np.array(b))
        assert rc == 0 and np.abs(A3 @ x - b).max() < 1e-10
        xf, rcf = O.lu_solve_fallback(A3, np.array(b, dtype=complex))
        assert rcf == 0 and np.abs(A3 @ xf - b).max() < 1e-10


def test_zgesv_is_lapack_on_the_transposed_view():
    """lu_solve hands the C-order buffer to LAPACK as column-major (= A^T) and solves with trans='T'."""
    rng = np.random.default_rng(3)
    n = 60
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)); b = rng.standard_normal(n) + 0j
    x, ipiv, rc = O.zgesv(A, b)
    lu, piv = sl.lu_factor(A.T, check_finite=False)
    assert rc == 0 and np.array_equal(ipiv[:n], piv)
    xr = sl.lu_solve((lu, piv), b, trans=1, check_finite=False)
    assert rel_l2(x, xr) < 1e-13 and rel_l2(x, np.linalg.solve(A, b)) < 1e-12


# ---------------------------------------------------------------- CSR / smoothers / GMRES
def _lap1d(n, shift=0.0):
    rows, cols, vals = [], [], []
    for i in range(n):
        for j, v in ((i - 1, -1.0), (i, 2.0 + shift), (i + 1, -1.0)):
            if 0 <= j < n:
                rows.append(i); cols.append(j); vals.append(v)
    import scipy.sparse as ss
    M = ss.csr_matrix((np.array(vals, dtype=complex), (rows, cols)), shape=(n, n))
    return M


def test_csr_matvec_known_answers():                     # csr.rs tests :659-736
    y = O.csr_matvec([0, 2, 4], [0, 1, 0, 1], [1, 2, 3, 4], [1, 2])
    assert np.allclose(y, [5, 11])
    y = O.csr_matvec([0, 2, 3, 5], [0, 2, 1, 0, 2], [1, 2, 3, 4, 5], [1, 1, 1])
    assert np.allclose(y, [3, 3, 9])
    M = _lap1d(300, 0.1 + 0.05j)
    x = np.sin(0.1 * np.arange(300)) + 1j * np.cos(0.2 * np.arange(300))
    assert np.allclose(O.csr_matvec(M.indptr, M.indices, M.data, x, nthreads=4), M @ x, rtol=1e-14)


def test_helmholtz_values_k0_is_stiffness():             # math-fem helmholtz.rs:354-390
    K = np.array([2.0, -1.0, 3.0]); Mv = np.array([0.5, 0.25, 1.0])
    assert np.allclose(O.helmholtz_values(K, Mv, 0.0), K)
    assert np.allclose(O.helmholtz_values(K, Mv, 2.0 + 0.1j), K - (2.0 + 0.1j) ** 2 * Mv)


def test_smoothers_reduce_the_residual():                # smoother.rs:192-237, amg.rs tests
    M = _lap1d(64, 0.05)
    b = np.ones(64, dtype=complex); x0 = np.zeros(64, dtype=complex)
    r0 = np.linalg.norm(b - M @ x0)
    for x in (O.amg_jacobi(M.indptr, M.indices, M.data, x0, b, 2.0 / 3.0, 5),
              O.amg_l1_jacobi(M.indptr, M.indices, M.data, x0, b, 5),
              O.amg_sym_gauss_seidel(M.indptr, M.indices, M.data, x0, b, 3)):
        assert np.linalg.norm(b - M @ x) < r0
    coo = M.tocoo()
    for kind in (0, 1, 2):
        x = O.fem_smooth(64, coo.row, coo.col, coo.data, x0, b, kind=kind, iterations=4)
        assert np.linalg.norm(O.fem_residual(64, coo.row, coo.col, coo.data, x, b)) < r0
    # one Jacobi sweep is x + omega D^-1 (b - A x)
    x1 = O.amg_jacobi(M.indptr, M.indices, M.data, x0 + 1.0, b, 0.8, 1)
    assert np.allclose(x1, (x0 + 1.0) + 0.8 * (b - M @ (x0 + 1.0)) / M.diagonal(), rtol=1e-14)


def test_gmres_solves_dense_and_csr():                   # gmres.rs:632-705
    rng = np.random.default_rng(1)
    n = 40
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) + 8 * np.eye(n)
    b = rng.standard_normal(n) + 0j
    x, info = O.gmres(b, dense=A, restart=30, max_iterations=50, tol=1e-10)
    assert info.converged == 1 and np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-8
    M = _lap1d(50, 0.5)
    bb = np.ones(50, dtype=complex)
    x, info = O.gmres(bb, csr=(M.indptr, M.indices, M.data), restart=50, max_iterations=10, tol=1e-10)
    assert info.converged == 1 and np.linalg.norm(M @ x - bb) / np.linalg.norm(bb) < 1e-8
    x, info = O.gmres(np.zeros(5, dtype=complex), dense=np.eye(5))
    assert info.converged == 1 and info.iterations == 0


def test_room_collocation_matrix_formula():              # room_acoustics/solver.rs:448-493
    om = O.icosphere(1.0, 1)
    k = 2.0
    A = O.room_build_matrix(om.center, om.normal, om.area, k, nthreads=2)
    i, j = 3, 40
    d = om.center[i] - om.center[j]; r = np.linalg.norm(d)
    ref = (1j * k * r - 1) * np.exp(1j * k * r) / (4 * np.pi * r * r) * (d @ om.normal[i]) / r * om.area[j]
    assert abs(A[i, j] - ref) <= 1e-14 * abs(ref)
    assert abs(A[5, 5] - (-1j * k / (2 * np.pi)) * om.area[5]) < 1e-16


def test_gmres_preconditioned_restatement():
    """gmres.rs:282-428: with the identity preconditioner the iteration is gmres itself; with Jacobi sweeps the
    tolerance is relative to ||M^-1 b|| and the true residual still falls."""
    rng = np.random.default_rng(21)
    n = 60
    dense = np.diag(4.0 + rng.random(n)) + 0.3 * rng.standard_normal((n, n)) + 0.1j * rng.standard_normal((n, n))
    rp = np.arange(0, n * n + 1, n); ci = np.tile(np.arange(n), n); vals = dense.reshape(-1)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    x0, i0 = O.gmres(b, csr=(rp, ci, vals), restart=20, max_iterations=10, tol=1e-10)
    x1, i1 = O.gmres_preconditioned(b, (rp, ci, vals), pkind=0, restart=20, max_iterations=10, tol=1e-10)
    assert i0.converged == i1.converged == 1 and i0.iterations == i1.iterations
    assert np.abs(x0 - x1).max() <= 1e-12
    for pk in (1, 2):
        x2, i2 = O.gmres_preconditioned(b, (rp, ci, vals), pkind=pk, omega=1.0, sweeps=1, restart=20, max_iterations=10, tol=1e-10)
        assert i2.converged == 1 and i2.iterations <= i0.iterations
        assert np.linalg.norm(dense @ x2 - b) / np.linalg.norm(b) < 1e-8


# ---------------------------------------------------------------- Quad4 restatement (regular.rs:211-260, singular.rs:257-357)
def _quad_brute(x, nx, coords, k, n=160):
    """G and dG/dn_y over a bilinear quad by an n x n mid-point... Gauss rule on [-1,1]^2 (independent of the restatement)."""
    g, w = np.polynomial.legendre.leggauss(n)
    S, T = np.meshgrid(g, g, indexing="ij"); W = np.outer(w, w)
    N = np.stack([0.25 * (S + 1) * (T + 1), 0.25 * (1 - S) * (T + 1), 0.25 * (1 - S) * (1 - T), 0.25 * (S + 1) * (1 - T)])
    dS = np.stack([0.25 * (T + 1), -0.25 * (T + 1), 0.25 * (T - 1), -0.25 * (T - 1)])
    dT = np.stack([0.25 * (S + 1), 0.25 * (1 - S), 0.25 * (S - 1), -0.25 * (S + 1)])
    P = np.einsum("aij,ad->ijd", N, coords); A = np.einsum("aij,ad->ijd", dS, coords); B = np.einsum("aij,ad->ijd", dT, coords)
    nrm = np.cross(A, B); J = np.linalg.norm(nrm, axis=2); ny = nrm / J[..., None]
    d = P - x; r = np.linalg.norm(d, axis=2)
    G = np.exp(1j * k * r) / (4 * np.pi * r)
    H = G * (-1.0 / r + 1j * k) * (np.einsum("ijd,ijd->ij", d, ny) / r)
    return (G * W * J).sum(), (H * W * J).sum()


def test_quad4_regular_integration_against_brute_force():
    """The restated Quad4 path (bilinear geometry, n x n rules by distance, quad-tree subdivision) against a 160 x 160
    Gauss rule on a warped quad: far point (4 x 4 rule) to 1e-6, near points to the GAU_ACCU level 5e-4 (singular.rs:507-512)."""
    coords = np.array([[1.0, 1.0, 0.1], [-1.0, 1.1, -0.05], [-0.9, -1.0, 0.0], [1.1, -0.9, 0.15]])
    area = 4.0
    k = 1.7
    nx = np.array([0.0, 0.0, 1.0])
    for x, tol in ((np.array([0.3, -0.2, 9.0]), 1e-6), (np.array([0.3, -0.2, 0.7]), 5e-4), (np.array([1.5, 0.4, 0.3]), 5e-4)):
        res = O.regular_integration(x, nx, coords, area, k)
        G, H = _quad_brute(x, nx, coords, k)
        assert abs(res[0] - G) <= tol * abs(G) and abs(res[1] - H) <= tol * max(abs(H), abs(G))
    # far away the quad and its two triangles integrate the same function
    x = np.array([0.5, 0.2, 12.0])
    flat = np.array([[1.0, 1.0, 0.0], [-1.0, 1.0, 0.0], [-1.0, -1.0, 0.0], [1.0, -1.0, 0.0]])
    q = O.regular_integration(x, nx, flat, 4.0, k)
    t = O.regular_integration(x, nx, flat[[0, 1, 2]], 2.0, k) + O.regular_integration(x, nx, flat[[0, 2, 3]], 2.0, k)
    assert np.all(np.abs(q[:4] - t[:4]) <= 1e-7 * np.abs(t[:4]).max())


def test_quad4_subelements_tile_the_square_and_self_term_is_planar():
    """Leaves of the quad-tree tile [-1,1]^2 (area 4) when no level overflows; the planar self term has Re G > 0 and
    vanishing H (the triangle's known answer, singular.rs:779-815, applied to a square)."""
    flat = np.array([[1.0, 1.0, 0.0], [-1.0, 1.0, 0.0], [-1.0, -1.0, 0.0], [1.0, -1.0, 0.0]])
    subs = O.generate_subelements(np.array([0.2, -0.1, 1.2]), flat, 4.0)
    assert len(subs) > 1
    assert abs(sum((2.0 * s.factor) ** 2 for s in subs) - 4.0) < 1e-12
    assert all(4 <= s.gauss_order <= 7 for s in subs)
    res = O.singular_integration(np.zeros(3), np.array([0.0, 0.0, 1.0]), flat, 1.0)
    assert res[0].real > 0 and abs(res[1]) < 1e-10 and abs(res[2]) < 1e-10
    # static limit of the planar self term: integral of 1/(4 pi r) over the square [-1,1]^2 from its centre = 2 ln(1+sqrt 2) * 4 / (4 pi) ...
    exact = (8.0 * np.log(1.0 + np.sqrt(2.0))) / (4.0 * np.pi)
    res0 = O.singular_integration(np.zeros(3), np.array([0.0, 0.0, 1.0]), flat, 1e-6)
    assert abs(res0[0].real - exact) <= 2e-3 * exact


def test_golden_mixed_quad_mesh_with_boundary_values():
    """Committed fixture: mixed Tri3 + Quad4 mesh with nodal velocity / pressure values (matrix and rhs)."""
    om = O.Mesh(GOLD["mixq_nodes"], GOLD["mixq_conn"])
    om.bc_type = GOLD["mixq_bc_type"].copy(); om.bc_len = GOLD["mixq_bc_len"].copy(); om.bc_values = GOLD["mixq_bc_values"].copy()
    A, rhs = O.build_tbem_system_with_beta(om, float(GOLD["mixq_k"][0]), complex(GOLD["mixq_beta"][0]), nthreads=4)
    assert np.allclose(A, GOLD["mixq_A"], rtol=1e-13, atol=1e-15) and np.allclose(rhs, GOLD["mixq_rhs"], rtol=1e-13, atol=1e-18)


def test_room_path_restatement_consistency():
    """room_acoustics/solver.rs: element data of a unit square quad and of its two triangles, adaptive assembly switched
    off equals build_bem_matrix_parallel, the incident derivative is -dG/dn of a monopole, the field pressure of a zero
    surface pressure is the incident field."""
    nodes = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0.2, 0.3, 1.0], [0.9, 0.3, 1.0], [0.9, 0.8, 1.2]], dtype=float)
    conn = np.array([[0, 1, 2, 3], [0, 1, 2, -1], [4, 5, 6, -1]], dtype=np.int32)
    c, nr, a, cl = O.room_element_data(nodes, conn)
    assert np.allclose(c[0], [0.5, 0.5, 0.0]) and np.allclose(nr[0], [0, 0, 1]) and abs(a[0] - 1.0) < 1e-15 and abs(a[1] - 0.5) < 1e-15
    assert abs(cl[0] - (1 + 1 + np.sqrt(2)) / 3) < 1e-15                    # the quad's length uses its first three nodes
    k = 3.0
    A0 = O.room_build_matrix_adaptive(nodes, conn, k, use_adaptive=False)
    assert np.abs(A0 - O.room_build_matrix(c, nr, a, k)).max() <= 1e-15
    A1 = O.room_build_matrix_adaptive(nodes, conn, k, use_adaptive=True)
    assert np.all(np.isfinite(A1.view(float))) and abs(A1[2, 2]) < 1e-10     # planar self term: vanishing double layer
    src = np.array([[0.5, 0.5, 2.0]])
    rhs = O.room_incident_derivative(c, nr, src, np.array([2.0]), k)
    d = c[0] - src[0]; r = np.linalg.norm(d)
    expect = -2.0 * (1j * k * r - 1) * np.exp(1j * k * r) / (4 * np.pi * r * r) * (d @ nr[0]) / r
    assert abs(rhs[0] - expect) < 1e-15
    pts = np.array([[0.1, 0.2, 0.7]])
    p = O.room_field_pressure(c, nr, a, np.zeros(3, dtype=complex), src, np.array([2.0]), pts, k)
    rr = np.linalg.norm(pts[0] - src[0])
    assert abs(p[0] - 2.0 * np.exp(1j * k * rr) / (4 * np.pi * rr)) < 1e-15


def test_amg_v_cycle_restatement_behaves_like_the_reference_tests():
    """amg.rs:1220+ (test_amg_*): the preconditioner reduces the residual of a Poisson-like system, apply() of a zero vector is
    zero, and a W-cycle is two V-cycles; F = V + V on the residual (amg.rs:1068-1103)."""
    import scipy.sparse as sp
    from amg_hierarchy import box_hierarchy, csr_triplet
    from math_audio_amd import fem
    nx, ny, nz = 8, 8, 4
    nodes, rp, ci, K, M = fem.helmholtz_box(nx, ny, nz)
    n = len(rp) - 1
    A = (sp.csr_matrix((K + 0.2 * M, ci, rp), shape=(n, n))).astype(np.complex128)
    levels = [{key: csr_triplet(l[key]) for key in l} for l in box_hierarchy(A, nx, ny, nz, 3)]
    b = A @ (np.sin(0.1 * np.arange(n)) + 1j * np.cos(0.2 * np.arange(n)))
    for sm in (0, 1, 2):
        H = O.AmgHierarchy(levels, smoother=sm)
        z = H.apply(b)
        assert np.linalg.norm(b - A @ z) < 0.6 * np.linalg.norm(b)          # one V-cycle from zero is a contraction
        assert np.abs(H.apply(np.zeros(n))).max() == 0.0
    V = O.AmgHierarchy(levels, cycle=0); F = O.AmgHierarchy(levels, cycle=2)
    z1 = V.apply(b)
    assert np.allclose(F.apply(b), z1 + V.apply(b - A @ z1), rtol=1e-13, atol=0)
    x, info = V.gmres(b, restart=30, max_iterations=5, tol=1e-8)
    assert info.converged == 1 and info.iterations < 15 and np.linalg.norm(A @ x - b) < 1e-6 * np.linalg.norm(b)


def test_pipelined_gmres_restatement():
    """gmres_pipelined.rs:258-285 (test_pgmres_simple_solver): [[4,1],[1,3]] x = [1,2], restart 10, tol 1e-10 -> converged with
    |Ax - b| < 1e-8; and on a larger well-conditioned system p-GMRES (classical Gram-Schmidt on Z = A V) reaches the solution
    of the standard GMRES (modified Gram-Schmidt) within the tolerance, with the same iteration count."""
    A = np.array([[4.0, 1.0], [1.0, 3.0]], dtype=complex); b = np.array([1.0, 2.0], dtype=complex)
    x, info = O.gmres_pipelined(b, dense=A, restart=10, max_iterations=100, tol=1e-10)
    assert info.converged == 1 and np.linalg.norm(A @ x - b) < 1e-8
    rng = np.random.default_rng(2)
    n = 150
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) + 40.0 * np.eye(n)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    xp, ip = O.gmres_pipelined(b, dense=A, restart=30, max_iterations=20, tol=1e-10)
    xs, is_ = O.gmres(b, dense=A, restart=30, max_iterations=20, tol=1e-10)
    assert ip.converged == 1 and is_.converged == 1 and abs(ip.iterations - is_.iterations) <= 1
    assert np.linalg.norm(xp - xs) <= 1e-8 * np.linalg.norm(xs)
    x0 = rng.standard_normal(n) + 0j
    xg, ig = O.gmres_pipelined(b, dense=A, x0=x0, restart=30, max_iterations=20, tol=1e-10)
    assert ig.converged == 1 and np.linalg.norm(A @ xg - b) <= 1e-8 * np.linalg.norm(b)
    assert O.gmres_pipelined(np.zeros(5, dtype=complex), dense=np.eye(5, dtype=complex))[1].iterations == 0


def test_slfmm_restatement_known_answers():
    """slfmm.rs tests (:790-877): system sizes (test_slfmm_system_creation / test_build_slfmm_system: 2 dofs, one T and S matrix),
    near block 2 x 2 with non-zero diagonal (test_near_field_block); plus what pins the ingredients: spherical_hankel_first_kind
    against SciPy, the sphere rule's weights summing to 1, <A x, z> = <x, A^T z> for matvec / matvec_transpose, and the single
    cluster of bem_solver.rs:375-381 reducing the operator to its near-field matrix."""
    import scipy.special as ss
    from fmm_clusters import Clusters, grid_clusters
    for x in (0.3, 2.0, 17.5):
        h = O.spherical_hankel_first_kind(6, x)
        ref = np.array([ss.spherical_jn(n, x) + 1j * ss.spherical_yn(n, x) for n in range(6)])
        assert np.abs(h - ref).max() <= 1e-10 * np.abs(ref).max()
    c, w = O.unit_sphere_quadrature(4, 8)
    assert len(w) == 32 and abs(w.sum() - 1.0) < 1e-14 and np.abs(np.linalg.norm(c, axis=1) - 1.0).max() < 1e-15
    nodes = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.5, 1.0, 0.0], [1.5, 1.0, 0.0]])          # slfmm.rs:801-841
    conn = np.array([[0, 1, 2, -1], [1, 3, 2, -1]], dtype=np.int32)
    om = O.Mesh(nodes, conn)
    one = Clusters([[0.5, 0.5, 0.0]], [0, 2], [0, 1], [0, 0], [], [0, 0], [])
    k = O.wave_number(100.0, 343.0)
    S = O.Slfmm(om, one, k, 4, 8, 5)
    N = S.near_matrix()
    assert N.shape == (2, 2) and abs(N[0, 0]) > 0.0 and abs(N[1, 1]) > 0.0
    sph = O.icosphere(RADIUS, 2)
    cl = grid_clusters(sph.center, 0.07)
    F = O.Slfmm(sph, cl, 10.0, 4, 8, 5)
    n = sph.n_elem
    x = np.sin(0.1 * np.arange(n)) + 1j * np.cos(0.2 * np.arange(n)); z = np.cos(0.3 * np.arange(n)) + 0.5j
    assert abs((F.matvec(x) * z).sum() - (x * F.matvec(z, transpose=True)).sum()) <= 1e-12 * abs((F.matvec(x) * z).sum())
    whole = grid_clusters(sph.center, 10.0)
    W = O.Slfmm(sph, whole, 10.0, 4, 8, 5)
    assert whole.n == 1 and np.abs(W.matvec(x) - W.near_matrix() @ x).max() <= 1e-12 * np.abs(W.matvec(x)).max()


def test_mlfmm_restatement_known_answers():
    """mlfmm.rs tests (:1267-1312): estimate_num_levels(10, 10, 1, 8) = 1, (100, 10, 1, 8) = 3, (1000, ...) >= 3; the tree of the
    two-triangle mesh has a root that holds both elements; the system has 2 dofs and >= 1 level; matvec returns 2 entries. Plus what
    pins the restatement: a two-level tree equals the single-level operator over its leaves minus the free term the multi-level
    near field does not add (the same T, D and S definitions, restated independently in C for slfmm.rs), and the quirks the
    docstring of oracle/oracle_mlfmm.py lists."""
    from fmm_clusters import Clusters
    M = O.mlfmm_module()
    assert M.estimate_num_levels(10, 10, 1, 8) == 1
    assert M.estimate_num_levels(100, 10, 1, 8) == 3
    assert M.estimate_num_levels(1000, 10, 1, 8) >= 3
    assert M.estimate_num_levels(0, 10, 2, 8) == 2
    nodes = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.5, 1.0, 0.0], [1.5, 1.0, 0.0]])          # mlfmm.rs:1231-1265
    conn = np.array([[0, 1, 2, -1], [1, 3, 2, -1]], dtype=np.int32)
    om = O.Mesh(nodes, conn)
    k = O.wave_number(100.0, 343.0)
    tree = M.build_cluster_tree(om.center, 10, k)
    assert len(tree) >= 1 and len(tree[0].clusters) >= 1 and len(tree[0].clusters[0].element_indices) == 2
    S = M.MlfmmSystem(om, tree, k, O)
    assert S.num_dofs == 2 and S.num_levels >= 1
    y = S.matvec(np.array([1.0, 1.0j]))
    assert y.shape == (2,) and np.all(np.isfinite(y))
    # expansion terms: clamp((kr + 6 max(ln kr, 1)) as usize, 4, 30)
    assert M._expansion_terms(0.0) == 6 and M._expansion_terms(0.5) == 6 and M._expansion_terms(3.1) == 9 and M._expansion_terms(100.0) == 30
    assert M._expansion_terms(float("nan")) == 4
    # a sphere, 3 levels at ka = 1: every level's theta is tabulated (6 or 7), lists are complementary, sons partition (or repeat) the father
    sph = O.icosphere(RADIUS, 3)                      # 1280 panels
    k1 = 1.0 / RADIUS
    T = M.build_cluster_tree(sph.center, 20, k1)
    assert len(T) >= 3
    for li, lv in enumerate(T):
        assert lv.theta_points == lv.expansion_terms and lv.phi_points == 2 * lv.expansion_terms
        for i, c in enumerate(lv.clusters):
            assert sorted(c.near_clusters + c.far_clusters) == [j for j in range(len(lv.clusters)) if j != i]
            if li + 1 < len(T):
                kids = set()
                for s_ in c.sons:
                    assert T[li + 1].clusters[s_].father == i
                    kids |= set(T[li + 1].clusters[s_].element_indices)
                assert kids == set(c.element_indices)
    # two levels (root + leaves) against the single-level operator over the same leaves: identical far field, near field without free
    # term. The sphere is moved off the dividing planes (an element centre ON a plane joins every octant that touches it -- the
    # centred icosphere has such elements: its leaves overlap, which the single-level restatement does not take)
    cnt = np.zeros(sph.n_elem, dtype=int)
    for c in T[1].clusters:
        cnt[c.element_indices] += 1
    assert cnt.max() == 2                               # the quirk is there
    off = O.Mesh(sph.nodes * np.array([1.0, 1.01, 0.99]) + np.array([0.0013, -0.0007, 0.0004]), sph.conn)
    T2 = M.build_cluster_tree(off.center, 200, k1)
    assert len(T2) == 2
    leaf = T2[1].clusters
    counts = np.zeros(off.n_elem, dtype=int)
    for c in leaf:
        counts[c.element_indices] += 1
    assert counts.max() == 1 and counts.min() == 1
    eptr = np.concatenate([[0], np.cumsum([len(c.element_indices) for c in leaf])])
    eidx = np.concatenate([c.element_indices for c in leaf])
    nptr = np.concatenate([[0], np.cumsum([len(c.near_clusters) for c in leaf])]); nidx = np.concatenate([c.near_clusters for c in leaf] + [[]]).astype(int)
    fptr = np.concatenate([[0], np.cumsum([len(c.far_clusters) for c in leaf])]); fidx = np.concatenate([c.far_clusters for c in leaf] + [[]]).astype(int)
    cl = Clusters(np.array([c.center for c in leaf]), eptr, eidx, nptr, nidx, fptr, fidx)
    lv = T2[1]
    assert lv.theta_points == 6
    x = np.sin(0.1 * np.arange(off.n_elem)) + 1j * np.cos(0.2 * np.arange(off.n_elem))
    from fmm_clusters import grid_clusters
    for kk, theta in ((k1, 6), (4.0 * k1, 8)):
        # a hand-made two-level tree: root + the grid clusters the single-level tests use (far pairs exist among them)
        g = grid_clusters(off.center, 0.07)
        root = M.Cluster([0.0, 0.0, 0.0]); root.element_indices = list(range(off.n_elem)); root.sons = list(range(g.n))
        l0 = M.ClusterLevel(); l0.clusters = [root]; l0.expansion_terms = l0.theta_points = 4; l0.phi_points = 8
        l1 = M.ClusterLevel(); l1.expansion_terms = 5; l1.theta_points = theta; l1.phi_points = 2 * theta
        for c in range(g.n):
            q = M.Cluster(g.center[c]); q.element_indices = [int(e) for e in g.elem_idx[g.elem_ptr[c]:g.elem_ptr[c + 1]]]
            q.near_clusters = [int(j) for j in g.near_idx[g.near_ptr[c]:g.near_ptr[c + 1]]]; q.far_clusters = [int(j) for j in g.far_idx[g.far_ptr[c]:g.far_ptr[c + 1]]]
            q.father = 0; q.level = 1
            l1.clusters.append(q)
        assert sum(len(q.far_clusters) for q in l1.clusters) > 0
        SL = O.Slfmm(off, g, kk, theta, 2 * theta, 5)
        ML = M.MlfmmSystem(off, [l0, l1], kk, O)
        ys, ym = SL.matvec(x), ML.matvec(x)
        assert np.abs(ym - (ys - 0.5 * x)).max() <= 1e-11 * np.abs(ys).max(), kk


def test_other_krylov_restatements_known_answers():
    """The reference's own tests on the numpy restatement: bicgstab.rs:190-219 and cgs.rs:151-180 (the 2 x 2 system to 1e-10, ||A x - b||
    < 1e-8), cg.rs:146-189 (the SPD 2 x 2 system; the identity with b = 1..5 in <= 2 iterations to 1e-10); plus b = 0 -> x = 0, converged,
    0 iterations (:57-64 of each), and SciPy's answers on a larger system."""
    K = O.krylov_module()
    A = np.array([[4.0, 1.0], [1.0, 3.0]], dtype=complex); b = np.array([1.0, 2.0], dtype=complex)
    for fn in (K.bicgstab, K.cgs, K.cg):
        x, it, res, conv = fn(lambda v: A @ v, b, 100, 1e-10)
        assert conv and np.linalg.norm(A @ x - b) < 1e-8 and it <= 3
        x0, it0, res0, conv0 = fn(lambda v: A @ v, np.zeros(2), 100, 1e-10)
        assert conv0 and it0 == 0 and res0 == 0.0 and np.all(x0 == 0)
    bi = np.arange(1, 6, dtype=complex)
    x, it, res, conv = K.cg(lambda v: v, bi, 10, 1e-12)
    assert conv and it <= 2 and np.linalg.norm(x - bi) < 1e-10
    rng = np.random.default_rng(7)
    n = 60
    Bm = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    G = Bm @ Bm.conj().T / n + 2.0 * np.eye(n)                          # Hermitian positive definite: all three apply
    rhs = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = np.linalg.solve(G, rhs)
    for fn in (K.bicgstab, K.cgs, K.cg):
        x, it, res, conv = fn(lambda v: G @ v, rhs, 500, 1e-10)
        assert conv and np.linalg.norm(x - ref) <= 1e-7 * np.linalg.norm(ref)


def test_sphere_series_known_answers_of_test_3d_sphere():
    """math-bem/tests/test_3d_sphere.rs on the restated series (the analytical reference of every BEM accuracy test): Rayleigh regime
    RCS / (pi a^2) < 0.01 at ka = 0.1 (:36-82), geometric limit |RCS / (pi a^2) - 2| < 0.3 at ka = 20 with ka + 20 terms (:124-166),
    the sweep ka = 0.1 .. 10: every value positive and finite, RCS(ka = 1.1) > RCS(ka = 0.1) (:169-220), and the series' error against
    150 terms falling from 5 to 10 to 20 terms at ka = 5 (:368-427)."""
    a = 1.0
    assert O.sphere_rcs_3d(0.1, a, 10) / (np.pi * a * a) < 0.01
    assert abs(O.sphere_rcs_3d(0.1, a, 10) / (np.pi * a * a) - (7.0 / 9.0) * 0.1 ** 4) < 0.03 * (7.0 / 9.0) * 0.1 ** 4   # the rigid sphere's Rayleigh law: a_0 = -i (ka)^3 / 3, a_1 = i (ka)^3 / 6 -> sigma = (7 pi / 9) a^2 (ka)^4
    ratio = O.sphere_rcs_3d(20.0, a, int(20.0 + 20.0)) / (np.pi * a * a)
    assert abs(ratio - 2.0) < 0.3
    rcs = [O.sphere_rcs_3d(0.1 * i / a, a, int(0.1 * i + 15.0)) for i in range(1, 101)]
    assert all(v > 0.0 and np.isfinite(v) for v in rcs) and rcs[10] > rcs[0]
    th = [0.0, np.pi / 4.0, np.pi / 2.0, 3.0 * np.pi / 4.0, np.pi]
    ref = O.sphere_scattering_3d(5.0, a, 150, [2.0], th)
    errs = []
    for terms in (5, 10, 20):
        sol = O.sphere_scattering_3d(5.0, a, terms, [2.0], th)
        errs.append(np.sqrt((np.abs(sol - ref) ** 2).sum() / (np.abs(ref) ** 2).sum()))
    assert errs[1] < errs[0] and errs[2] < errs[1]


def test_ilu_restatement_known_answers():
    """ilu.rs:177-274: the 3 x 3 tridiagonal system (A applied to M^-1 r within 0.5 of r; ILU(0) of a tridiagonal matrix is its exact LU:
    1e-14 here) and the 10 x 10 one under GMRES(10) (preconditioned iterations <= plain + 5); test_fmm_validation.rs:589-640
    (gmres_solve_with_ilu on the 30 x 30 band system: converged, ||A x - b|| / ||b|| < 1e-5)."""
    import scipy.sparse as sp
    ILU = O.ilu_module()
    A = sp.csr_matrix(np.array([[4.0, -1.0, 0.0], [-1.0, 4.0, -1.0], [0.0, -1.0, 4.0]], dtype=complex))
    P = ILU.IluPreconditioner(A.indptr, A.indices, A.data)
    r = np.array([1.0, 2.0, 3.0], dtype=complex)
    z = P.apply(r)
    assert np.abs(A @ z - r).max() < 1e-14
    n = 10
    T = sp.diags([-1.0, 4.0, -1.0], [-1, 0, 1], shape=(n, n)).tocsr().astype(complex)
    b = np.sin(np.arange(n)).astype(complex)
    x0, i0 = O.gmres(b, csr=(T.indptr, T.indices, T.data), restart=10, max_iterations=50, tol=1e-10)
    P = ILU.IluPreconditioner(T.indptr, T.indices, T.data)
    assert i0.converged and np.linalg.norm(T @ P.apply(b) - b) < 1e-13      # exact LU again: one preconditioned step solves it
    # ILU(0) with a dropped entry: the band system with the extra diagonal i + 3 (fill at (i + 1, i + 3) is dropped)
    n = 30
    M = np.zeros((n, n), dtype=complex)
    for i in range(n):
        M[i, i] = 5.0 + 0.5j
        if i > 0:
            M[i, i - 1] = -2.0 + 0.2j
        if i < n - 1:
            M[i, i + 1] = -2.0 - 0.2j
        if i + 3 < n:
            M[i, i + 3] = 0.5
    S = sp.csr_matrix(M)
    P = ILU.IluPreconditioner(S.indptr, S.indices, S.data)
    bb = np.sin(0.2 * np.arange(n)) + 0.5 + 0.1j
    z = P.apply(bb)
    assert 1e-6 < np.linalg.norm(M @ z - bb) / np.linalg.norm(bb) < 0.2      # incomplete: close to, not equal to, the inverse
    # gmres_solve_with_ilu feeds the DENSE matrix' pattern (from_dense(matrix, 1e-15)): the zero entries are dropped, the pattern is the band's
    from scipy.sparse.linalg import gmres as sgmres, LinearOperator as SLO
    xs, info = sgmres(S, bb, M=SLO((n, n), matvec=P.apply, dtype=complex), restart=20, maxiter=100, rtol=1e-10)
    assert info == 0 and np.linalg.norm(M @ xs - bb) / np.linalg.norm(bb) < 1e-5


# ---------------------------------------------------------------- AMG setup (AmgPreconditioner::from_csr, amg.rs:276-372)
def _amg_setup():
    return O.amg_setup_module()


def test_amg_setup_csr_algebra_against_scipy():
    """from_triplets / matmul / transpose of the restatement (csr.rs:135-205, 594-651; amg.rs:810-822) against SciPy on random
    complex matrices: same pattern where no product is below the 1e-15 cut, same values to rounding; duplicates accumulate."""
    import scipy.sparse as sp
    S = _amg_setup()
    rng = np.random.default_rng(5)
    A = sp.random(40, 30, density=0.15, random_state=1, format="csr").astype(np.complex128); A.data = rng.standard_normal(A.nnz) + 1j * rng.standard_normal(A.nnz)
    B = sp.random(30, 25, density=0.2, random_state=2, format="csr").astype(np.complex128); B.data = rng.standard_normal(B.nnz) + 1j * rng.standard_normal(B.nnz)
    a, b = S.from_scipy(A), S.from_scipy(B)
    c = S.to_scipy(S.matmul(a, b)); ref = (A @ B).tocsr(); ref.sort_indices()
    assert (c.indptr == ref.indptr).all() and (c.indices == ref.indices).all()
    assert np.abs(c.data - ref.data).max() <= 1e-14 * np.abs(ref.data).max()
    t = S.to_scipy(S.transpose(a)); reft = A.T.tocsr(); reft.sort_indices()
    assert (t.indptr == reft.indptr).all() and (t.indices == reft.indices).all() and (t.data == reft.data).all()
    m = S.from_triplets(3, 3, [(2, 1, 1 + 1j), (0, 0, 2 + 0j), (2, 1, 0.5 + 0j), (1, 2, 3j)])
    assert m.ptr == [0, 1, 2, 3] and m.col == [0, 2, 1] and m.val == [2 + 0j, 3j, 1.5 + 1j]
    assert S.get if hasattr(S, "get") else True
    assert m.get(2, 1) == 1.5 + 1j and m.get(0, 2) == 0j


def test_amg_setup_reference_unit_tests():
    """amg.rs:1158-1266 through the restatement (hierarchy) and the restated cycle (oracle_solvers.c): test_amg_creation,
    test_amg_apply, test_amg_pmis_coarsening, test_amg_different_smoothers, test_amg_reduces_residual, test_diagnostics."""
    S = _amg_setup()
    lv, gc, oc = S.from_csr(S.laplacian_1d(100), S.default_config())
    assert len(lv) >= 2 and gc >= 1.0 and oc >= 1.0
    assert len([l["A"].nr for l in lv]) == len(lv) and len([l["A"].nnz() for l in lv]) == len(lv)
    cfg = S.default_config(); cfg["coarsening"] = S.PMIS
    assert len(S.from_csr(S.laplacian_1d(100), cfg)[0]) >= 2
    m50 = S.laplacian_1d(50)
    r = np.arange(50, dtype=np.complex128)
    for smoother in (0, 1, 2):
        lv50, _, _ = S.from_csr(m50, S.default_config())
        z = O.AmgHierarchy(O.amg_levels_as_triplets(lv50), smoother=smoother).apply(r)
        assert z.shape == r.shape and np.abs(z - r).sum() > 1e-10
    n = 64
    m = S.laplacian_1d(n); A = S.to_scipy(m)
    H = O.AmgHierarchy(O.amg_levels_as_triplets(S.from_csr(m, S.default_config())[0]))
    b = np.sin(np.arange(n)).astype(np.complex128); x = np.zeros(n, dtype=np.complex128)
    r0 = np.linalg.norm(b - A @ x)
    for _ in range(10):
        x = x + H.apply(b - A @ x)
    assert np.linalg.norm(b - A @ x) < 0.1 * r0


def test_amg_setup_coarsening_properties():
    """What the two coarsenings promise: Ruge-Stuben leaves no F-point without the chance of a strong C-neighbour on the 1-D
    Laplacian (every second point is coarse); PMIS' C-points form an independent set of the strength graph unless promoted at the
    end; P has the identity on C-points and rows of F-points sum to 1 for a zero-row-sum operator (Direct weights -a_ij / a_ii)."""
    S = _amg_setup()
    m = S.laplacian_1d(101)
    strong = S.strength(m, 0.25)
    pt, c2f = S.coarsen_ruge_stuben(m, strong)
    assert all(pt[i] != pt[i + 1] or pt[i] == S.FINE for i in range(100)) and 40 <= len(c2f) <= 60
    pt2, c2f2 = S.coarsen_pmis(m, strong)
    assert not any(pt2[i] == S.COARSE and pt2[i + 1] == S.COARSE for i in range(100))
    P = S.build_interpolation(m, strong, pt, c2f, S.DIRECT, 0.0, 4)
    for ci, fi in enumerate(c2f):
        assert list(P.row(fi)) == [(ci, 1 + 0j)]
    for i in range(1, 100):
        if pt[i] == S.FINE and pt[i - 1] == S.COARSE and pt[i + 1] == S.COARSE:
            assert abs(sum(v for _, v in P.row(i)) - 1.0) < 1e-15
