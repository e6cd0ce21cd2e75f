"""GPU parity of the single-level fast multipole operator (math-bem/src/core/assembly/slfmm.rs: build_slfmm_system, SlfmmSystem::
matvec / matvec_transpose / extract_near_field_matrix) against the CPU restatement, on the sphere of the QA suite and on the
box of BASELINE.json configs[4]; the checks math-bem/tests/test_fmm_validation.rs makes (operator vs assembled matrix on
x_i = sin(0.1 i) + i cos(0.2 i), :103-130) are made against the reference's own near-field matrix, which is what a single
cluster reduces the operator to."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import mesh as mm
from helpers import to_ma_mesh, k_from_ka, RADIUS, rel_l2
from fmm_clusters import grid_clusters

pytestmark = pytest.mark.gpu


def _xvec(n):
    i = np.arange(n)
    return np.sin(0.1 * i) + 1j * np.cos(0.2 * i)


# sphere rules of 32, 72, 288 (more than one pass of 256 points in the upward pass, three translation passes of 128) and 6 points
@pytest.mark.parametrize("sub,ka,cell,nt,nphi", [(2, 1.0, 0.07, 4, 8), (3, 3.0, 0.05, 6, 12), (2, 0.2, 10.0, 4, 8), (2, 2.0, 0.07, 12, 24), (2, 1.0, 0.07, 2, 3)])
def test_slfmm_operator_matches_the_restatement(gpu, sub, ka, cell, nt, nphi):
    om = O.icosphere(RADIUS, sub)
    k = k_from_ka(ka)
    cl = grid_clusters(om.center, cell)
    ref = O.Slfmm(om, cl, k, nt, nphi, 5)
    plan = ma.BemPlan(to_ma_mesh(om))
    op = ma.LinearOperator.slfmm(plan, cl, k, nt, nphi, 5)
    n = om.n_elem
    N_ref = ref.near_matrix(); N = op.slfmm_near_matrix()
    scale = np.abs(N_ref).max(axis=1, keepdims=True)
    assert (np.abs(N - N_ref) / scale).max() <= 1e-9                    # blocks integrated by the TBEM near / self kernels
    for x in (_xvec(n), np.ones(n, dtype=complex)):
        y = op.apply(x); yr = ref.matvec(x)
        assert np.abs(y - yr).max() <= 1e-10 * np.abs(yr).max()
        yt = op.apply_transpose(x); ytr = ref.matvec(x, transpose=True)
        assert np.abs(yt - ytr).max() <= 1e-10 * np.abs(ytr).max()
        assert np.abs(op.apply_hermitian(x) - np.conj(ref.matvec(np.conj(x), transpose=True))).max() <= 1e-10 * np.abs(ytr).max()
    if cl.n == 1:                                                       # bem_solver.rs:375-381: one cluster holding everything: A = [N]
        assert np.abs(op.apply(_xvec(n)) - N_ref @ _xvec(n)).max() <= 1e-10 * np.abs(N_ref @ _xvec(n)).max()
    # the far field is a genuine contribution when clusters are apart
    if cl.n > 1:
        assert np.abs(ref.matvec(_xvec(n)) - N_ref @ _xvec(n)).max() > 1e-6 * np.abs(N_ref @ _xvec(n)).max()
    # GMRES through the LinearOperator boundary (SlfmmSystem implements LinearOperator, slfmm.rs:378-395)
    b = ma.incident_rhs(om.center, om.normal, k, complex(0.0, 1.0 / k))
    xg, info = ma.gmres(op, b, restart=30, max_iterations=5, tol=1e-6)
    y = ref.matvec(xg)
    assert info.converged == 0 or np.linalg.norm(y - b) <= 1e-4 * np.linalg.norm(b)
    op.close(); plan.close()


def test_slfmm_sphere_rule_beyond_the_fast_passes(gpu):
    """a sphere rule of more than 1024 points (20 x 56 = 1120; 20 is the largest tabulated Gauss order) does not fit the LDS arrays of the recomputed-phase passes;
    the operator must then run the STORED-TABLE passes (mode 1) -- it used to fall silently to the libm form (mode 0), the slowest of the
    three -- and still be the restatement's operator. A rule of 288 points runs the recomputed form (mode 2)."""
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(2.0)
    cl = grid_clusters(om.center, 0.07)
    plan = ma.BemPlan(to_ma_mesh(om))
    n = om.n_elem
    small = ma.LinearOperator.slfmm(plan, cl, k, 12, 24, 5)
    assert small.slfmm_phase_mode() == 2
    small.close()
    op = ma.LinearOperator.slfmm(plan, cl, k, 20, 56, 5)
    assert op.slfmm_phase_mode() == 1
    ref = O.Slfmm(om, cl, k, 20, 56, 5)
    x = _xvec(n)
    y = op.apply(x); yr = ref.matvec(x)
    assert np.abs(y - yr).max() <= 1e-10 * np.abs(yr).max()
    yt = op.apply_transpose(x); ytr = ref.matvec(x, transpose=True)
    assert np.abs(yt - ytr).max() <= 1e-10 * np.abs(ytr).max()
    op.close(); plan.close()


def test_slfmm_on_the_box_and_input_validation(gpu):
    m = mm.generate_box_mesh(0.30, 0.40, 0.60, 5, 6, 9)
    om = O.Mesh(m.nodes, m.conn)
    k = mm.wave_number(1000.0)
    cl = grid_clusters(om.center, 0.12)
    assert cl.n > 20
    ref = O.Slfmm(om, cl, k, 8, 16, 6)
    plan = ma.BemPlan(to_ma_mesh(om))
    op = ma.LinearOperator.slfmm(plan, cl, k, 8, 16, 6)
    x = _xvec(om.n_elem)
    yr = ref.matvec(x)
    assert np.abs(op.apply(x) - yr).max() <= 1e-10 * np.abs(yr).max()
    ytr = ref.matvec(x, transpose=True)
    assert np.abs(op.apply_transpose(x) - ytr).max() <= 1e-10 * np.abs(ytr).max()
    op.close()
    with pytest.raises(ma.MaError) as e:                               # 9 is not a tabulated Gauss-Legendre order (gauss.rs:27-60)
        ma.LinearOperator.slfmm(plan, cl, k, 9, 16, 6)
    assert e.value.status == ma.MA_ERR_INVALID
    bad = grid_clusters(om.center, 0.12); bad.elem_idx = bad.elem_idx.copy(); bad.elem_idx[1] = bad.elem_idx[0]
    with pytest.raises(ma.MaError) as e:
        ma.LinearOperator.slfmm(plan, bad, k, 4, 8, 5)
    assert e.value.status == ma.MA_ERR_UNSUPPORTED
    plan.close()


def test_operator_apply_is_linear_and_reproducible_at_size(gpu):
    """5 120 panels (icosphere 4), 150 clusters: properties that do not depend on the size. The apply is linear; two applies of the
    same vector agree BIT FOR BIT (near blocks: partial sums gathered in entry order; translation: parts added in wavefront order;
    upward / downward passes: fixed lane-set reductions); the first-version kernels (MA_FMM_NEAR_BLOCKS=0, MA_FMM_DENSE_TRANSLATE=0,
    MA_FMM_STORE_PHASES=0) give the same vector to rounding."""
    import os
    om = O.icosphere(RADIUS, 4)
    k = 3.0 / RADIUS
    cl = grid_clusters(om.center, 0.035)
    plan = ma.BemPlan(to_ma_mesh(om))
    op = ma.LinearOperator.slfmm(plan, cl, k, 6, 12, 5)
    n = om.n_elem
    i = np.arange(n)
    x = np.sin(0.1 * i) + 1j * np.cos(0.2 * i); z = np.cos(0.3 * i) + 0.25j
    a, b = 0.7 - 0.2j, -1.3 + 0.4j
    y = op.apply(x)
    assert np.isfinite(y).all()
    assert (op.apply(x) == y).all() and (op.apply(x) == y).all()
    lin = op.apply(a * x + b * z) - (a * y + b * op.apply(z))
    assert np.abs(lin).max() <= 1e-12 * np.abs(y).max()
    yt = op.apply_transpose(x)
    assert (op.apply_transpose(x) == yt).all()
    old = {v: os.environ.get(v) for v in ("MA_FMM_NEAR_BLOCKS", "MA_FMM_DENSE_TRANSLATE", "MA_FMM_STORE_PHASES")}
    try:
        for v in old:
            os.environ[v] = "0"
        op0 = ma.LinearOperator.slfmm(plan, cl, k, 6, 12, 5)
    finally:
        for v, val in old.items():
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val
    y0 = op0.apply(x); yt0 = op0.apply_transpose(x)
    assert np.abs(y0 - y).max() <= 1e-12 * np.abs(y).max() and np.abs(yt0 - yt).max() <= 1e-12 * np.abs(yt).max()
    # MA_FMM_OVERLAP=0 (read at an operator's first apply): near and far field on one stream, the same vector to rounding
    old_ov = os.environ.get("MA_FMM_OVERLAP")
    try:
        os.environ["MA_FMM_OVERLAP"] = "0"
        op1 = ma.LinearOperator.slfmm(plan, cl, k, 6, 12, 5)
        y1 = op1.apply(x)
    finally:
        if old_ov is None:
            os.environ.pop("MA_FMM_OVERLAP", None)
        else:
            os.environ["MA_FMM_OVERLAP"] = old_ov
    assert np.abs(y1 - y).max() <= 1e-12 * np.abs(y).max() and (op1.apply(x) == y1).all()
    op1.close(); op0.close(); op.close()


def test_slfmm_over_octree_leaves(gpu):
    """The reference's own spatial partition (Octree::build + compute_interaction_lists, mesh/octree.rs, mirrored in
    math_audio_amd/octree.py) as the clusters of build_slfmm_system: device operator against the restatement on the same lists."""
    from math_audio_amd import octree as T
    om = O.icosphere(RADIUS, 3)
    k = k_from_ka(2.0)
    tree = T.Octree.build(om.center, 40, 5)
    tree.compute_interaction_lists(1.5)
    cl = T.slfmm_clusters(tree)
    assert cl.n == tree.stats().num_leaves > 8 and cl.far_ptr[-1] > 0
    ref = O.Slfmm(om, cl, k, 6, 12, 5)
    plan = ma.BemPlan(to_ma_mesh(om))
    op = ma.LinearOperator.slfmm(plan, cl, k, 6, 12, 5)
    x = _xvec(om.n_elem)
    y = op.apply(x); yr = ref.matvec(x)
    assert np.abs(y - yr).max() <= 1e-10 * np.abs(yr).max()
    yt = op.apply_transpose(x); ytr = ref.matvec(x, transpose=True)
    assert np.abs(yt - ytr).max() <= 1e-10 * np.abs(ytr).max()
    op.close(); plan.close()
