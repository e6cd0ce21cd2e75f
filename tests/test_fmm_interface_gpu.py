"""The solver wrappers and mesh utilities of math-bem/src/core/solver/fmm_interface.rs (:356-600) through
`math_audio_amd.fmm_interface`: each wrapper against the call it stands for, with the reference's quirks (CGS "with ILU" runs
without it, the hierarchical preconditioner is the identity)."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import fmm_interface as F
from test_ilu_gpu import to_ma_mesh, RADIUS

pytestmark = pytest.mark.gpu


def _dense_system(n=96, seed=4):
    rng = np.random.default_rng(seed)
    A = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) * 0.05 + np.eye(n) * (2.0 + 0.3j)
    A[np.abs(A) < 0.03] = 0.0                      # some exact zeros for from_dense to drop
    x = np.sin(0.1 * np.arange(n)) + 1j * np.cos(0.2 * np.arange(n))
    return A, A @ x, x


def test_dense_wrappers(gpu):
    A, b, x = _dense_system()
    cfg = F.KrylovConfig(max_iterations=300, tolerance=1e-10, restart=30)
    op = ma.LinearOperator.dense(A)
    for solve in (F.solve_cgs, F.solve_bicgstab, F.solve_gmres):
        xs, info = solve(op, b, cfg)
        assert info.converged and np.abs(xs - x).max() <= 1e-7
    ref, iref = ma.cgs(op, b, 300, 1e-10)
    for xs, info in (F.solve_with_ilu(A, b, cfg), F.solve_tbem_with_ilu(A, b, cfg), F.solve_with_ilu_operator(op, A, b, cfg)):
        assert info.iterations == iref.iterations and np.abs(xs - ref).max() <= 1e-12      # plain CGS, whatever the name says (:389-439)
    rp, ci, v = F.csr_from_dense(A)
    assert len(v) == int((np.abs(A) > 1e-15).sum()) and rp[-1] == len(v)
    xg, ig = F.gmres_solve_with_ilu(A, b, cfg)
    xt, it = F.gmres_solve_tbem_with_ilu(A, b, cfg)
    x0, i0 = ma.gmres(op, b, restart=30, max_iterations=300, tol=1e-10)
    assert ig.converged and it.iterations == ig.iterations and np.abs(xg - x).max() <= 1e-7 and ig.iterations <= i0.iterations
    op.close()


def test_fmm_wrappers(gpu):
    from fmm_clusters import grid_clusters
    om = O.icosphere(RADIUS, 2)
    k = 1.0 / RADIUS
    cl = grid_clusters(om.center, 0.07)
    plan = ma.BemPlan(to_ma_mesh(om))
    op = ma.LinearOperator.slfmm(plan, cl, k, 4, 8, 5)
    b = np.ones(om.n_elem, dtype=complex)
    cfg = F.KrylovConfig(max_iterations=200, tolerance=1e-8, restart=30)
    x0, i0 = ma.gmres(op, b, restart=30, max_iterations=200, tol=1e-8)
    xh, ih = F.gmres_solve_fmm_hierarchical(op, b, cfg)                     # identity preconditioner (:326-355)
    xb, ib = F.gmres_solve_fmm_batched(op, b, cfg)
    assert ih.iterations == i0.iterations == ib.iterations and np.abs(xh - x0).max() <= 1e-13 and np.abs(xb - x0).max() <= 1e-13
    assert F.gmres_solve_with_hierarchical_precond is F.gmres_solve_fmm_hierarchical
    xi, ii = F.gmres_solve_fmm_batched_with_ilu(op, b, cfg)                  # ILU(0) of extract_near_field_matrix (:527-538)
    xo, io = F.gmres_solve_with_ilu_operator(op, op.slfmm_near_matrix(), b, cfg)
    assert ii.converged and ii.iterations == io.iterations and np.abs(xi - xo).max() <= 1e-12
    assert np.linalg.norm(op.apply(xi) - b) <= 1e-6 * np.linalg.norm(b) and ((not i0.converged) or ii.iterations <= i0.iterations)
    # SparseNearfieldIlu (:249-297) = division by the diagonal of the self blocks
    P = F.sparse_nearfield_ilu(op)
    d = np.diag(op.slfmm_near_matrix())
    r = np.sin(np.arange(om.n_elem)) + 0.5j
    assert np.abs(P.apply(r) - r / d).max() <= 1e-13 * np.abs(r / d).max()
    P.close(); op.close()


def test_mesh_utilities(gpu):
    """:544-603. 6 elements per wavelength at 343 Hz in air: wavelength 1 m, 6 elements per metre."""
    assert F.recommended_mesh_resolution(343.0, 343.0, 6) == 6.0
    assert F.mesh_resolution_for_frequency_range(20.0, 686.0, 343.0, 6) == 12.0
    assert F.estimate_element_count((5.0, 4.0, 2.5), 2.0) == int(np.ceil(2.0 * (20.0 + 12.5 + 10.0) / 0.25))
    c = F.AdaptiveMeshConfig.for_frequency_range(20.0, 343.0)
    assert c.base_resolution == 6.0 and c.source_refinement == 1.5 and c.source_refinement_radius == 0.5
    c = F.AdaptiveMeshConfig.from_resolution(3.0)
    assert c.base_resolution == 3.0 and c.source_refinement == 1.0 and c.source_refinement_radius == 0.0
