"""Host-side mirror of the reference's octree (math-bem/src/core/mesh/octree.rs): AABB, OctreeNode, Octree::build /
leaves / level_nodes / non_empty_nodes / compute_interaction_lists / stats, the same arithmetic in the same order (plain floats).
Like the mesh generators and file readers of this package it is setup the reference keeps on the host; `slfmm_clusters` turns the
leaves and their interaction lists into the lists `ma_op_create_slfmm` (LinearOperator.slfmm) takes."""
import math
import numpy as np


class AABB:                                     # octree.rs:11-117
    def __init__(self, mn, mx):
        self.min = [float(v) for v in mn]; self.max = [float(v) for v in mx]

    @staticmethod
    def empty():
        return AABB([math.inf] * 3, [-math.inf] * 3)

    def expand(self, p):
        for i in range(3):
            if p[i] < self.min[i]:
                self.min[i] = float(p[i])
            if p[i] > self.max[i]:
                self.max[i] = float(p[i])

    def center(self):
        return [(self.min[i] + self.max[i]) / 2.0 for i in range(3)]

    def half_size(self):
        return [(self.max[i] - self.min[i]) / 2.0 for i in range(3)]

    def max_dimension(self):
        s = [self.max[i] - self.min[i] for i in range(3)]
        return max(max(s[0], s[1]), s[2])

    def contains(self, p):
        return all(self.min[i] <= p[i] <= self.max[i] for i in range(3))

    def child_index(self, p):
        c = self.center()
        return (1 if p[0] >= c[0] else 0) | (2 if p[1] >= c[1] else 0) | (4 if p[2] >= c[2] else 0)

    def child_bounds(self, index):
        c = self.center()
        mn, mx = list(self.min), list(self.max)
        for d in range(3):
            if index & (1 << d):
                mn[d] = c[d]
            else:
                mx[d] = c[d]
        return AABB(mn, mx)


class OctreeNode:                               # :120-166
    def __init__(self, bounds, level, parent):
        self.bounds, self.center, self.level, self.parent = bounds, bounds.center(), level, parent
        self.children = None
        self.element_indices, self.near_clusters, self.far_clusters = [], [], []

    def is_leaf(self):
        return self.children is None

    def radius(self):
        h = self.bounds.half_size()
        return math.sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2])


class OctreeStats:                              # :404-418
    def __init__(self, num_nodes, num_leaves, num_levels, avg, mx, mn):
        self.num_nodes, self.num_leaves, self.num_levels = num_nodes, num_leaves, num_levels
        self.avg_elements_per_leaf, self.max_elements_per_leaf, self.min_elements_per_leaf = avg, mx, mn


class Octree:                                   # :169-401
    def __init__(self, max_per_leaf, max_depth):
        self.nodes, self.max_elements_per_leaf, self.max_depth, self.num_leaves, self.num_levels = [], max_per_leaf, max_depth, 0, 0

    @staticmethod
    def build(centers, max_per_leaf, max_depth):
        t = Octree(max_per_leaf, max_depth)
        centers = [[float(p[0]), float(p[1]), float(p[2])] for p in centers]
        if not centers:
            return t
        b = AABB.empty()
        for p in centers:
            b.expand(p)
        pad = b.max_dimension() * 0.01
        for i in range(3):
            b.min[i] -= pad; b.max[i] += pad
        md = b.max_dimension(); c = b.center(); half = md / 2.0
        b.min = [c[0] - half, c[1] - half, c[2] - half]; b.max = [c[0] + half, c[1] + half, c[2] + half]
        root = OctreeNode(b, 0, None)
        root.element_indices = list(range(len(centers)))
        t.nodes.append(root)
        t._subdivide(0, centers)
        t.num_leaves = sum(1 for n in t.nodes if n.is_leaf())
        t.num_levels = max(n.level for n in t.nodes) + 1
        return t

    def _subdivide(self, node_idx, centers):   # :245-291 (recursion depth <= max_depth)
        node = self.nodes[node_idx]
        if node.level >= self.max_depth or len(node.element_indices) <= self.max_elements_per_leaf:
            return
        bounds, level, elems = node.bounds, node.level, list(node.element_indices)
        first = len(self.nodes)
        child = [first + i for i in range(8)]
        for i in range(8):
            self.nodes.append(OctreeNode(bounds.child_bounds(i), level + 1, node_idx))
        for e in elems:
            self.nodes[child[bounds.child_index(centers[e])]].element_indices.append(e)
        node.element_indices = []
        node.children = child
        for ci in child:
            if self.nodes[ci].element_indices:
                self._subdivide(ci, centers)

    def leaves(self):
        return [i for i, n in enumerate(self.nodes) if n.is_leaf() and n.element_indices]

    def level_nodes(self, level):
        return [i for i, n in enumerate(self.nodes) if n.level == level]

    def non_empty_nodes(self):
        return [i for i, n in enumerate(self.nodes) if n.element_indices or n.children is not None]

    def compute_interaction_lists(self, separation_ratio):   # :327-370
        leaves = self.leaves()
        data = [(self.nodes[i].center, self.nodes[i].radius()) for i in leaves]
        for a, i in enumerate(leaves):
            ci, ri = data[a]
            near, far = [], []
            for b_, j in enumerate(leaves):
                if i == j:
                    near.append(j)
                    continue
                cj, rj = data[b_]
                d = [ci[0] - cj[0], ci[1] - cj[1], ci[2] - cj[2]]
                dist = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
                if dist > separation_ratio * (ri + rj):
                    far.append(j)
                else:
                    near.append(j)
            self.nodes[i].near_clusters, self.nodes[i].far_clusters = near, far

    def stats(self):
        leaves = self.leaves()
        per = [len(self.nodes[i].element_indices) for i in leaves]
        return OctreeStats(len(self.nodes), len(leaves), self.num_levels, (sum(per) / len(leaves)) if leaves else 0.0, max(per) if per else 0, min(per) if per else 0)


class Clusters:
    """The CSR-style cluster lists of ma_clusters_t (types.rs:445-488 per cluster: element_indices, center, near_clusters, far_clusters)."""

    def __init__(self, center, elem_ptr, elem_idx, near_ptr, near_idx, far_ptr, far_idx):
        self.center = np.ascontiguousarray(center, dtype=np.float64).reshape(-1, 3)
        self.elem_ptr = np.ascontiguousarray(elem_ptr, dtype=np.int32); self.elem_idx = np.ascontiguousarray(elem_idx, dtype=np.int32)
        self.near_ptr = np.ascontiguousarray(near_ptr, dtype=np.int32); self.near_idx = np.ascontiguousarray(near_idx, dtype=np.int32)
        self.far_ptr = np.ascontiguousarray(far_ptr, dtype=np.int32); self.far_idx = np.ascontiguousarray(far_idx, dtype=np.int32)
        self.n = len(self.elem_ptr) - 1


def slfmm_clusters(octree):
    """One cluster per non-empty leaf (after compute_interaction_lists): leaf node indices renumbered 0..n-1 in leaf order, a leaf's own
    index dropped from its near list (the operator always takes the self block)."""
    leaves = octree.leaves()
    num = {node: c for c, node in enumerate(leaves)}
    center, ep, ei, np_, ni, fp, fi = [], [0], [], [0], [], [0], []
    for node in leaves:
        n = octree.nodes[node]
        center.append(n.center)
        ei += n.element_indices; ep.append(len(ei))
        ni += [num[j] for j in n.near_clusters if j != node]; np_.append(len(ni))
        fi += [num[j] for j in n.far_clusters]; fp.append(len(fi))
    return Clusters(center, ep, ei, np_, ni, fp, fi)
