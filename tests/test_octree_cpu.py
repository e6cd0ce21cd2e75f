"""The octree mirror (math_audio_amd/octree.py) against the reference's own tests (math-bem/src/core/mesh/octree.rs:424-540) and the
properties its construction promises; host-side setup, no GPU."""
import numpy as np
from math_audio_amd import octree as T

POINTS = [[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]]


def test_reference_unit_tests():
    c = T.AABB([0, 0, 0], [1, 1, 1]).center()
    assert all(abs(v - 0.5) < 1e-10 for v in c)
    a = T.AABB([0, 0, 0], [2, 2, 2])
    assert a.child_index([0.5, 0.5, 0.5]) == 0 and a.child_index([1.5, 1.5, 1.5]) == 7
    t = T.Octree.build(POINTS, 2, 4)
    assert t.nodes and sum(len(t.nodes[i].element_indices) for i in t.leaves()) == len(POINTS)
    assert T.Octree.build([], 2, 4).nodes == []
    one = T.Octree.build([[0.5, 0.5, 0.5]], 2, 4)
    assert len(one.nodes) == 1 and one.nodes[0].is_leaf() and len(one.nodes[0].element_indices) == 1
    t1 = T.Octree.build(POINTS, 1, 4)
    assert len(t1.level_nodes(0)) == 1 and t1.nodes[0].children is not None
    t.compute_interaction_lists(1.5)
    for leaf in t.leaves():
        assert leaf in t.nodes[leaf].near_clusters
    s = t.stats()
    assert s.num_leaves > 0 and s.num_levels >= 1


def test_construction_properties():
    """Cubic root box padded by 1 % of the largest extent; children halve the box; every element sits in exactly one leaf and inside its
    bounds; child_index and child_bounds agree; near + far = all leaves; the lists are symmetric (the criterion is)."""
    rng = np.random.default_rng(2)
    pts = rng.random((500, 3)) * np.array([2.0, 1.0, 0.5]) + np.array([0.3, -0.2, 1.0])
    t = T.Octree.build(pts, 12, 6)
    root = t.nodes[0].bounds
    ext = [root.max[i] - root.min[i] for i in range(3)]
    assert max(ext) - min(ext) < 1e-12 and abs(ext[0] - 2.0 * 1.02 * (pts[:, 0].max() - pts[:, 0].min()) / 2.0) < 1e-9
    seen = np.zeros(len(pts), dtype=int)
    for i in t.leaves():
        n = t.nodes[i]
        assert len(n.element_indices) <= 12 or n.level == 6
        for e in n.element_indices:
            seen[e] += 1
            assert n.bounds.contains(pts[e])
    assert (seen == 1).all()
    for n in t.nodes:
        if n.children is not None:
            for k, ci in enumerate(n.children):
                ch = t.nodes[ci]
                assert ch.level == n.level + 1 and ch.parent is not None
                assert n.bounds.child_index(ch.center) == k
                assert abs(ch.radius() - n.radius() / 2.0) < 1e-12
    t.compute_interaction_lists(1.5)
    leaves = t.leaves()
    for i in leaves:
        n = t.nodes[i]
        assert sorted(n.near_clusters + n.far_clusters) == leaves
        for j in n.far_clusters:
            assert i in t.nodes[j].far_clusters
    st = t.stats()
    assert st.num_nodes == len(t.nodes) and st.num_leaves == len(leaves) and st.min_elements_per_leaf >= 1
    assert abs(st.avg_elements_per_leaf * st.num_leaves - len(pts)) < 1e-9
    cl = T.slfmm_clusters(t)
    assert cl.n == len(leaves) and cl.elem_ptr[-1] == len(pts) and sorted(cl.elem_idx) == list(range(len(pts)))
    for c in range(cl.n):
        assert c not in cl.near_idx[cl.near_ptr[c]:cl.near_ptr[c + 1]]
        assert len(set(cl.near_idx[cl.near_ptr[c]:cl.near_ptr[c + 1]]) & set(cl.far_idx[cl.far_ptr[c]:cl.far_ptr[c + 1]])) == 0
