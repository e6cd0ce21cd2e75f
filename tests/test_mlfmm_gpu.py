"""GPU parity of the multi-level fast multipole operator (math-bem/src/core/assembly/mlfmm.rs: build_cluster_tree, build_mlfmm_system,
MlfmmSystem::matvec) against the numpy restatement (oracle/oracle_mlfmm.py): the centred icosphere (element centres on octant
boundaries: overlapping leaves), the same sphere moved off the planes, the box of BASELINE.json configs[4], a tree of one level
(near field only), the levels the device may skip, and what it refuses."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import mesh as mm
from helpers import to_ma_mesh, RADIUS
from mlfmm_common import assert_same_tree

pytestmark = pytest.mark.gpu


def _xvec(n):
    i = np.arange(n)
    return np.sin(0.1 * i) + 1j * np.cos(0.2 * i)


def _pair(om, target, k):
    M = O.mlfmm_module()
    ref_tree = M.build_cluster_tree(om.center, target, k)
    ref = M.MlfmmSystem(om, ref_tree, k, O)
    mesh = to_ma_mesh(om)
    plan = ma.BemPlan(mesh)
    tree = ma.ClusterTree(mesh, target, k)
    assert_same_tree(tree, ref_tree)
    return plan, tree, ref, ref_tree


@pytest.mark.parametrize("shifted,target,ka", [(False, 20, 1.0), (True, 20, 1.0), (False, 20, 3.0), (True, 40, 2.0), (False, 100, 1.0)])
def test_mlfmm_operator_matches_the_restatement(gpu, shifted, target, ka):
    om = O.icosphere(RADIUS, 3 if target < 100 else 1)
    if shifted:
        om = O.Mesh(om.nodes * np.array([1.0, 1.01, 0.99]) + np.array([0.0013, -0.0007, 0.0004]), om.conn)
    k = ka / RADIUS
    plan, tree, ref, ref_tree = _pair(om, target, k)
    leaf = ref_tree[-1].clusters
    cnt = np.zeros(om.n_elem, dtype=int)
    for c in leaf:
        cnt[c.element_indices] += 1
    assert cnt.max() == (1 if shifted or target >= 100 else 2)           # the centred sphere has elements in two leaves
    nfar = [sum(len(c.far_clusters) for c in lv.clusters) for lv in ref_tree]
    if target < 100:
        assert len(ref_tree) >= 3 and sum(nfar) > 0 and nfar[0] == 0
    else:
        assert len(ref_tree) == 1                                        # one level: the near field alone (mlfmm.rs:186)
    op = ma.LinearOperator.mlfmm(plan, tree, k)
    n = om.n_elem
    for x in (_xvec(n), np.ones(n, dtype=complex)):
        y = op.apply(x); yr = ref.matvec(x)
        assert np.abs(y - yr).max() <= 1e-10 * np.abs(yr).max()
    with pytest.raises(ma.MaError):                                      # fmm_interface.rs:131-134: unimplemented!()
        op.apply_transpose(_xvec(n))
    # the operator drives GMRES like any other (it is the reference's model operator, not the dense matrix: only convergence of the iteration is checked)
    b = np.ones(n, dtype=complex)
    xs, info = ma.gmres(op, b, restart=30, max_iterations=60, tol=1e-8)
    if info.converged:
        assert np.linalg.norm(op.apply(xs) - b) <= 1e-6 * np.linalg.norm(b)


def test_mlfmm_on_the_box(gpu):
    box = mm.generate_box_mesh(1.0, 1.3, 0.7, 10, 13, 7)                 # 1124 panels, flat faces, 90-degree edges
    om = O.Mesh(box.nodes, box.conn)
    k = 6.0
    plan, tree, ref, ref_tree = _pair(om, 24, k)
    assert sum(len(c.far_clusters) for lv in ref_tree for c in lv.clusters) > 0
    thetas = [lv.theta_points for lv in ref_tree]
    op = ma.LinearOperator.mlfmm(plan, tree, k)
    x = _xvec(om.n_elem)
    y = op.apply(x); yr = ref.matvec(x)
    assert np.abs(y - yr).max() <= 1e-10 * np.abs(yr).max(), thetas


def test_mlfmm_refuses_what_the_reference_would_garble(gpu):
    """A level that takes part in the far field with a theta_points outside the Gauss-Legendre tables: the reference builds its T and S
    with the next table's longer rule while D and the level's vectors keep theta_points * phi_points entries, and its length guards then
    drop stages silently. The device path says so instead."""
    om = O.icosphere(RADIUS, 3)
    M = O.mlfmm_module()
    mesh = to_ma_mesh(om)
    plan = ma.BemPlan(mesh)
    found = False
    for ka in (4.5, 5.0, 5.5, 6.0, 7.0, 8.0):
        k = ka / RADIUS
        T = M.build_cluster_tree(om.center, 20, k)
        first = next((l for l, lv in enumerate(T) if any(c.far_clusters for c in lv.clusters)), None)
        if first is None:
            continue
        bad = [lv.theta_points for lv in T[first:] if lv.theta_points not in (4, 5, 6, 7, 8, 10, 12, 16, 20)]
        if bad:
            tree = ma.ClusterTree(mesh, 20, k)
            with pytest.raises(ma.MaError) as e:
                ma.LinearOperator.mlfmm(plan, tree, k)
            assert e.value.status == ma.MA_ERR_UNSUPPORTED
            found = True
            break
    assert found
