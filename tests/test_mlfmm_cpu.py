"""build_cluster_tree of the library (host code, no GPU needed) against the restatement of mlfmm.rs:954-1223."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import mesh as mm
from mlfmm_common import assert_same_tree

RADIUS = 0.1


def _both(om, target, k):
    M = O.mlfmm_module()
    ref = M.build_cluster_tree(om.center, target, k)
    lib_mesh = ma.MeshArrays(om.nodes, om.conn, om.center, om.normal, om.area)
    return ma.ClusterTree(lib_mesh, target, k), ref


@pytest.mark.parametrize("sub,target,ka", [(2, 10, 0.5), (3, 20, 1.0), (3, 20, 3.0), (3, 200, 1.0), (1, 100, 1.0)])
def test_cluster_tree_is_the_restatements(sub, target, ka):
    om = O.icosphere(RADIUS, sub)
    tree, ref = _both(om, target, ka / RADIUS)
    assert_same_tree(tree, ref)


def test_cluster_tree_of_a_box_and_of_the_reference_two_triangles():
    box = mm.generate_box_mesh(1.0, 1.3, 0.7, 6, 8, 4)
    om = O.Mesh(box.nodes, box.conn)
    tree, ref = _both(om, 16, 9.0)
    assert_same_tree(tree, ref)
    assert tree.num_levels() >= 3
    nodes = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.5, 1.0, 0.0], [1.5, 1.0, 0.0]])          # mlfmm.rs:1231-1265
    conn = np.array([[0, 1, 2, -1], [1, 3, 2, -1]], dtype=np.int32)
    two = O.Mesh(nodes, conn)
    tree, ref = _both(two, 10, O.wave_number(100.0, 343.0))
    assert_same_tree(tree, ref)
    lv = tree.level(0)
    assert tree.num_levels() >= 1 and lv["n_clusters"] >= 1 and lv["elem_ptr"][1] - lv["elem_ptr"][0] == 2     # test_build_cluster_tree (:1277-1286)


def test_cluster_tree_argument_checks():
    om = O.icosphere(RADIUS, 1)
    lib_mesh = ma.MeshArrays(om.nodes, om.conn, om.center, om.normal, om.area)
    with pytest.raises(ma.MaError):
        ma.ClusterTree(lib_mesh, 0, 1.0)
