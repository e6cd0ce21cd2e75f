"""The reference's test_3d_plane_wave (math-fem/tests/analytical_validation.rs:1237-1286) through the device path: CSR operator +
ma_gmres(restart 50, max 500, tol 1e-10); the nodal L2 error must be < 0.05 and equal the oracle's to 1e-6."""
import numpy as np
import pytest

import fem_plane_wave_case as pw
import oracle_lib as orc
import math_audio_amd as ma
from math_audio_amd import fem

pytestmark = pytest.mark.gpu


def _device_solve(rp, col, val, rhs):
    c = ma.CsrOperator(rp, col, values=val)
    op = ma.LinearOperator.csr(c)
    x, info = ma.gmres(op, rhs, restart=pw.RESTART, max_iterations=pw.MAX_ITERATIONS, tol=pw.TOLERANCE)
    op.close(); c.close()
    return x, bool(info.converged)


def _oracle_solve(rp, col, val, rhs):
    x, info = orc.gmres(rhs, csr=(rp, col, val), restart=pw.RESTART, max_iterations=pw.MAX_ITERATIONS, tol=pw.TOLERANCE)
    return x, bool(info.converged)


def test_3d_plane_wave_on_the_device(gpu):
    err_dev, case, x_dev = pw.run(_device_solve)
    err_orc, _, x_orc = pw.run(_oracle_solve)
    assert abs(err_dev - err_orc) < 1e-6
    assert np.linalg.norm(x_dev - x_orc) / np.linalg.norm(x_orc) < 1e-8


def test_3d_plane_wave_from_the_host_generator(gpu):
    """The same case built by the product's own pieces: fem.box_mesh_tetrahedra + assemble_p1 (K, M on one pattern), K - k^2 M,
    fem.apply_dirichlet_csr, device GMRES -- the chain a13b -> a13 -> a12 -> a11 of SURVEY section 8."""
    fem_o = pw.oracle_fem()
    case = fem_o.plane_wave_3d_case()
    nodes, tets = fem.box_mesh_tetrahedra(0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 4, 4, 4)
    rp, ci, K, M = fem.assemble_p1(nodes, tets)
    k = 2.0
    bn = fem.boundary_nodes(tets)
    assert set(int(v) for v in bn) == set(case["dirichlet"].keys())
    g = np.array([case["analytical"](*nodes[i]) for i in bn])
    rp2, ci2, v2, b2 = fem.apply_dirichlet_csr(rp, ci, K - (k * k) * M, np.zeros(len(nodes), dtype=complex), bn, g)
    # the eliminated system equals the restatement's (same pattern up to dropped zeros, same values to rounding, same right-hand side)
    import scipy.sparse as sp
    n = len(nodes)
    A_prod = sp.csr_matrix((v2, ci2, rp2), shape=(n, n)); A_orc = sp.csr_matrix((case["val"], case["col"], case["row_ptr"]), shape=(n, n))
    assert abs(A_prod - A_orc).max() < 1e-13 and np.abs(b2 - case["rhs"]).max() < 1e-13
    x, ok = _device_solve(rp2, ci2, v2, b2)
    assert ok
    err = fem_o.l2_error(nodes, x, case["analytical"])
    err_orc, _, _ = pw.run(_oracle_solve)
    assert err < pw.THRESHOLD and abs(err - err_orc) < 1e-6
