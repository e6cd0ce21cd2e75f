"""Test-side builder of single-level FMM clusters: the INPUT of build_slfmm_system (slfmm.rs:417-470).

The reference builds its clusters with an octree (math-bem/src/core/mesh/octree.rs, out of scope: SURVEY 2c); the operator
takes them as given -- element_indices, centre, near_clusters, far_clusters per cluster (types.rs:445-488). Here: a uniform grid
of cells of edge `cell` over the mesh's bounding box, one cluster per non-empty cell (centre = centre of the cell), near = the
other non-empty cells among the 26 neighbours, far = every other cluster. Returned as the CSR-style lists the C side takes."""
import numpy as np


class Clusters:
    def __init__(self, center, elem_ptr, elem_idx, near_ptr, near_idx, far_ptr, far_idx):
        self.center = np.ascontiguousarray(center, dtype=np.float64)
        self.elem_ptr = np.ascontiguousarray(elem_ptr, dtype=np.int32); self.elem_idx = np.ascontiguousarray(elem_idx, dtype=np.int32)
        self.near_ptr = np.ascontiguousarray(near_ptr, dtype=np.int32); self.near_idx = np.ascontiguousarray(near_idx, dtype=np.int32)
        self.far_ptr = np.ascontiguousarray(far_ptr, dtype=np.int32); self.far_idx = np.ascontiguousarray(far_idx, dtype=np.int32)
        self.n = len(self.elem_ptr) - 1


def grid_clusters(centers, cell):
    centers = np.asarray(centers, dtype=np.float64)
    lo = centers.min(axis=0) - 1e-9
    ijk = np.floor((centers - lo) / cell).astype(np.int64)
    keys, inv = np.unique(ijk, axis=0, return_inverse=True)
    inv = inv.ravel()
    nc = len(keys)
    order = np.argsort(inv, kind="stable")
    counts = np.bincount(inv, minlength=nc)
    elem_ptr = np.concatenate([[0], np.cumsum(counts)])
    elem_idx = order
    ccenter = lo + (keys + 0.5) * cell
    index = {tuple(k): c for c, k in enumerate(keys)}
    near_ptr, near_idx, far_ptr, far_idx = [0], [], [0], []
    for c, k in enumerate(keys):
        near = set()
        for d in np.ndindex(3, 3, 3):
            q = (k[0] + d[0] - 1, k[1] + d[1] - 1, k[2] + d[2] - 1)
            if q in index and index[q] != c:
                near.add(index[q])
        near_idx += sorted(near); near_ptr.append(len(near_idx))
        far_idx += [j for j in range(nc) if j != c and j not in near]; far_ptr.append(len(far_idx))
    return Clusters(ccenter, elem_ptr, elem_idx, near_ptr, near_idx, far_ptr, far_idx)
