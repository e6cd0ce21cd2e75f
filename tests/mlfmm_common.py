"""Shared by the MLFMM tests: the library's cluster tree (ma.ClusterTree, host code of libmathaudio_hip.so) against the restatement's
(oracle/oracle_mlfmm.py), list by list."""
import numpy as np


def assert_same_tree(lib_tree, oracle_levels):
    assert lib_tree.num_levels() == len(oracle_levels)
    for l, olv in enumerate(oracle_levels):
        lv = lib_tree.level(l)
        assert lv["n_clusters"] == len(olv.clusters)
        assert (lv["expansion_terms"], lv["theta_points"], lv["phi_points"]) == (olv.expansion_terms, olv.theta_points, olv.phi_points)
        for c, oc in enumerate(olv.clusters):
            assert np.array_equal(lv["center"][c], np.asarray(oc.center)), (l, c)            # bit for bit: the same arithmetic in the same order
            assert lv["radius"][c] == oc.radius
            assert lv["father"][c] == (-1 if oc.father is None else oc.father)
            for nm, ref in (("elem", oc.element_indices), ("near", oc.near_clusters), ("far", oc.far_clusters), ("son", oc.sons)):
                got = lv[nm + "_idx"][lv[nm + "_ptr"][c]:lv[nm + "_ptr"][c + 1]]
                assert list(got) == list(ref), (l, c, nm)
