"""The reference's test_3d_plane_wave (math-fem/tests/analytical_validation.rs:1237-1286) on the oracle: the only reference-held
number that pins P1-tet element matrices -> K - k^2 M -> Dirichlet elimination -> CSR -> GMRES as one chain."""
import numpy as np

import fem_plane_wave_case as pw
import oracle_lib as orc


def _oracle_solve(rp, col, val, rhs):
    x, info = orc.gmres(rhs, csr=(rp, col, val), restart=pw.RESTART, max_iterations=pw.MAX_ITERATIONS, tol=pw.TOLERANCE)
    return x, bool(info.converged)


def test_3d_plane_wave_on_the_oracle():
    err, case, x = pw.run(_oracle_solve)
    # the boundary rows are identities: the solution carries the imposed values exactly
    for node, value in case["dirichlet"].items():
        assert abs(x[node] - value) < 1e-9
    assert case["row_ptr"][-1] == len(case["val"]) and len(case["dirichlet"]) == 98 and len(case["rhs"]) == 125
    assert 1e-4 < err < 0.05


def test_element_matrices_against_closed_forms():
    fem = pw.oracle_fem()
    # P1 tetrahedron: K_e = V grad(phi_i).grad(phi_j), M_e = V / 20 (1 + delta_ij) -- what math_audio_amd/fem.py vectorises
    coords = [(0.1, 0.0, 0.2), (1.2, 0.1, 0.0), (0.3, 0.9, 0.1), (0.2, 0.3, 1.1)]
    p = np.array(coords)
    J = np.stack([p[1] - p[0], p[2] - p[0], p[3] - p[0]], axis=1)
    V = abs(np.linalg.det(J)) / 6.0
    g = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]) @ np.linalg.inv(J)
    assert np.allclose(fem.element_stiffness_tet_p1(coords), V * g @ g.T, rtol=1e-12, atol=1e-14)
    assert np.allclose(fem.element_mass_tet_p1(coords), V / 20.0 * (np.ones((4, 4)) + np.eye(4)), rtol=1e-12, atol=1e-15)


def test_generator_matches_the_host_generator():
    # the product's vectorised generator (math_audio_amd/fem.py) against the loop restatement: same nodes, same elements in order
    from math_audio_amd import fem as prod
    fem = pw.oracle_fem()
    n1, t1 = fem.box_mesh_tetrahedra(0.0, 1.0, 0.0, 2.0, 0.0, 0.5, 3, 4, 2)
    n2, t2 = prod.box_mesh_tetrahedra(0.0, 1.0, 0.0, 2.0, 0.0, 0.5, 3, 4, 2)
    assert np.array_equal(n1, n2) and np.array_equal(t1, t2)
