"""oracle_mlfmm.py -- CPU restatement (numpy) of the reference's multi-level fast multipole operator. TEST INFRASTRUCTURE ONLY:
only tests/ may import it; the product path (math_audio_amd/) never does.

Follows math-bem/src/core/assembly/mlfmm.rs function by function:
  estimate_num_levels (:954-974), build_cluster_tree (:979-1038), compute_bounding_box (:1041-1053), subdivide_level
  (:1056-1180), compute_near_far_lists (:1183-1223), build_mlfmm_system (:483-558), build_leaf_dof_mappings (:561-578),
  build_near_field / compute_near_block (:590-710, the integrals through the C restatement: mao_fmm_near_block),
  build_t_matrices_level (:713-800), build_d_matrices_level (:805-850), build_s_matrices_level (:853-935) and
  MlfmmSystem::matvec with upward_pass / translate_all_levels / downward_pass / evaluate_locals (:128-460).
The reference states its far field as a "simplified model" (:838-840); it is restated as it stands, including
  * the octant test `d * offset >= 0` on every axis (:1122-1128): a centre that lies ON a dividing plane joins every octant
    that touches it (its dof then appears in several leaf clusters and is summed several times);
  * near blocks stored for (i, i) and (i, j > i) only, the (j, i) block applied as the TRANSPOSE of (i, j) (:170-182), and no
    free term on the diagonal;
  * gauss_legendre's fall-back to the next tabulated order (gauss.rs:27-60): a level whose theta_points is not tabulated gets
    MORE sphere points than theta_points * phi_points, and the length guards of matvec (:229, :295, :341, :383, :447) then skip
    what no longer fits. Every guard is kept.
"""
import math
import numpy as np


def estimate_num_levels(num_elements, elements_per_leaf, min_levels, max_levels):        # :954-974
    if num_elements == 0:
        return min_levels
    levels, n = 1, num_elements
    while n > elements_per_leaf and levels < max_levels:
        n //= 8
        levels += 1
    return min(max(levels, min_levels), max_levels)


class Cluster:
    def __init__(self, center):                                                           # types.rs:490-516
        self.center = [float(center[0]), float(center[1]), float(center[2])]
        self.element_indices = []
        self.radius = 0.0
        self.near_clusters, self.far_clusters = [], []
        self.father, self.sons, self.level = None, [], 0

    def clone(self):
        c = Cluster(self.center)
        c.element_indices = list(self.element_indices); c.radius = self.radius
        c.near_clusters = list(self.near_clusters); c.far_clusters = list(self.far_clusters)
        c.father = self.father; c.sons = list(self.sons); c.level = self.level
        return c


class ClusterLevel:
    def __init__(self):                                                                   # types.rs:550-568
        self.clusters = []
        self.num_original = 0
        self.max_radius, self.avg_radius, self.min_radius = 0.0, 0.0, float(np.finfo(np.float64).max)
        self.expansion_terms, self.theta_points, self.phi_points = 4, 4, 8


def _expansion_terms(kr):
    """((kr + 6.0 * kr.ln().max(1.0)) as usize).clamp(4, 30): f64::max ignores a NaN, `as usize` truncates and saturates."""
    if kr > 0.0:
        ln = math.log(kr)
    elif kr == 0.0:
        ln = -math.inf
    else:
        ln = math.nan
    m = 1.0 if (ln != ln or ln < 1.0) else ln
    v = kr + 6.0 * m
    if v != v or v <= 0.0:
        u = 0
    elif v >= 1.8446744073709552e19:
        u = 2 ** 64 - 1
    else:
        u = int(v)
    return min(max(u, 4), 30)


def compute_near_far_lists(clusters, wave_number):                                        # :1183-1223
    n = len(clusters)
    for i in range(n):
        ci, ri = clusters[i].center, clusters[i].radius
        near, far = [], []
        for j in range(n):
            if i == j:
                continue
            cj, rj = clusters[j].center, clusters[j].radius
            d0 = ci[0] - cj[0]; d1 = ci[1] - cj[1]; d2 = ci[2] - cj[2]
            dist = math.sqrt(d0 * d0 + d1 * d1 + d2 * d2)                     # powi(2) is a product
            separation = dist / max(ri + rj, 1e-15)
            kr = wave_number * dist
            if separation > 2.0 and kr > 2.0:
                far.append(j)
            else:
                near.append(j)
        clusters[i].near_clusters, clusters[i].far_clusters = near, far


_OFFSETS = [(-1.0, -1.0, -1.0), (-1.0, -1.0, 1.0), (-1.0, 1.0, -1.0), (-1.0, 1.0, 1.0),
            (1.0, -1.0, -1.0), (1.0, -1.0, 1.0), (1.0, 1.0, -1.0), (1.0, 1.0, 1.0)]


def _subdivide_level(levels, centers, parent_level, target, k):                           # :1056-1180
    parents = [c.clone() for c in levels[parent_level].clusters]
    child_level = ClusterLevel()
    max_r, min_r, sum_r = 0.0, float(np.finfo(np.float64).max), 0.0
    for parent_idx, parent in enumerate(parents):
        if len(parent.element_indices) <= target:
            leaf = parent.clone()
            leaf.level = parent_level + 1; leaf.father = parent_idx
            max_r = max(max_r, leaf.radius); min_r = min(min_r, leaf.radius); sum_r += leaf.radius
            levels[parent_level].clusters[parent_idx].sons.append(len(child_level.clusters))
            child_level.clusters.append(leaf)
            continue
        half_size = parent.radius / 2.0
        for off in _OFFSETS:
            cc = [parent.center[0] + off[0] * half_size * 0.5, parent.center[1] + off[1] * half_size * 0.5, parent.center[2] + off[2] * half_size * 0.5]
            ce = []
            for idx in parent.element_indices:
                dx = centers[idx][0] - parent.center[0]; dy = centers[idx][1] - parent.center[1]; dz = centers[idx][2] - parent.center[2]
                if dx * off[0] >= 0.0 and dy * off[1] >= 0.0 and dz * off[2] >= 0.0:
                    ce.append(idx)
            if not ce:
                continue
            child = Cluster(cc)
            child.element_indices = ce; child.radius = half_size; child.level = parent_level + 1; child.father = parent_idx
            max_r = max(max_r, child.radius); min_r = min(min_r, child.radius); sum_r += child.radius
            levels[parent_level].clusters[parent_idx].sons.append(len(child_level.clusters))
            child_level.clusters.append(child)
    nc = len(child_level.clusters)
    child_level.num_original = nc
    child_level.max_radius, child_level.min_radius = max_r, min_r
    child_level.avg_radius = sum_r / float(nc) if nc > 0 else 0.0
    child_level.expansion_terms = _expansion_terms(k * child_level.avg_radius)
    child_level.theta_points = child_level.expansion_terms
    child_level.phi_points = 2 * child_level.expansion_terms
    levels.append(child_level)
    cur = len(levels) - 1
    if any(len(c.element_indices) > target for c in levels[cur].clusters) and cur < 7:
        _subdivide_level(levels, centers, cur, target, k)


def build_cluster_tree(centers, target_elements_per_leaf, wave_number):                   # :979-1038
    centers = [[float(v) for v in row] for row in np.asarray(centers, dtype=np.float64)]
    n = len(centers)
    num_levels = estimate_num_levels(n, target_elements_per_leaf, 1, 8)
    fmax = float(np.finfo(np.float64).max)
    lo, hi = [fmax] * 3, [-fmax] * 3          # f64::MIN is -f64::MAX
    for c in centers:
        for d in range(3):
            lo[d] = min(lo[d], c[d]); hi[d] = max(hi[d], c[d])
    root_center = [(lo[0] + hi[0]) / 2.0, (lo[1] + hi[1]) / 2.0, (lo[2] + hi[2]) / 2.0]
    e0 = hi[0] - lo[0]; e1 = hi[1] - lo[1]; e2 = hi[2] - lo[2]
    root_radius = math.sqrt(e0 * e0 + e1 * e1 + e2 * e2) / 2.0
    root = Cluster(root_center)
    root.element_indices = list(range(n)); root.radius = root_radius; root.level = 0
    level0 = ClusterLevel()
    level0.expansion_terms = _expansion_terms(wave_number * root_radius)
    level0.theta_points = level0.expansion_terms; level0.phi_points = 2 * level0.expansion_terms
    level0.clusters.append(root); level0.num_original = 1
    level0.max_radius = level0.avg_radius = level0.min_radius = root_radius
    levels = [level0]
    if num_levels > 1:
        _subdivide_level(levels, centers, 0, target_elements_per_leaf, wave_number)
    for lv in levels:
        compute_near_far_lists(lv.clusters, wave_number)
    return levels


class MlfmmSystem:
    """build_mlfmm_system(elements, nodes, cluster_levels, physics) and matvec. `mesh`: tests/oracle_lib.py's mesh arrays;
    `O`: tests/oracle_lib (near blocks, unit_sphere_quadrature, spherical_hankel_first_kind come from the C restatement)."""

    def __init__(self, mesh, levels, k, O, harmonic=1.0, tau=1.0):
        self.num_dofs = int(mesh.n_elem)                                                  # count_dofs: one dof per (non-evaluation) element
        self.num_levels = len(levels)
        self.levels = levels
        self.sphere_points_per_level = [lv.theta_points * lv.phi_points for lv in levels]
        dof = np.asarray(mesh.dof)
        leaf = levels[-1].clusters
        self.leaf_dof_indices = [[int(dof[e]) for e in c.element_indices] for c in leaf]   # :561-578
        # ---- near field (:590-644)
        self.near = []
        for i, ci in enumerate(leaf):
            self.near.append((i, i, O.fmm_near_block(mesh, ci.element_indices, ci.element_indices, True, k, harmonic, tau)))
            for j in ci.near_clusters:
                if j > i:
                    self.near.append((i, j, O.fmm_near_block(mesh, ci.element_indices, leaf[j].element_indices, False, k, harmonic, tau)))
        center = np.asarray(mesh.center, dtype=np.float64).reshape(-1, 3)
        # ---- T matrices, leaves to root, then reversed (:505-522)
        self.t = [None] * self.num_levels
        self.s = [None] * self.num_levels
        for li, lv in enumerate(levels):
            sc, sw = O.unit_sphere_quadrature(lv.theta_points, lv.phi_points)
            self.t[li] = self._level_matrices(lv, li, levels, center, k, sc, sw, -1.0)
            self.s[li] = [m.T.copy() for m in self._level_matrices(lv, li, levels, center, k, sc, sw, 1.0)]
        # ---- D entries per level (:805-850)
        self.d = []
        for li, lv in enumerate(levels):
            nsp = lv.theta_points * lv.phi_points
            ent = []
            for i, ci in enumerate(lv.clusters):
                for j in ci.far_clusters:
                    cj = lv.clusters[j]
                    diff = [ci.center[0] - cj.center[0], ci.center[1] - cj.center[1], ci.center[2] - cj.center[2]]
                    r = math.sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2])
                    if r < 1e-15:
                        continue
                    h = O.spherical_hankel_first_kind(max(lv.expansion_terms, 2), k * r, 1.0)
                    ent.append((i, j, np.full(nsp, h[0] * complex(0.0, k))))
            self.d.append(ent)

    @staticmethod
    def _level_matrices(lv, li, levels, center, k, sc, sw, sign):
        """T (sign -1: e^{-i k s.d}) of build_t_matrices_level, P x n; the S matrices (:853-935) are the same entries with e^{+i k s.d},
        transposed."""
        nsp = len(sc)
        is_leaf = li == len(levels) - 1
        out = []
        for cl in lv.clusters:
            if is_leaf:
                m = np.zeros((nsp, len(cl.element_indices)), dtype=np.complex128)
                for j, e in enumerate(cl.element_indices):
                    diff = center[e] - np.asarray(cl.center)
                    sd = sc[:, 0] * diff[0] + sc[:, 1] * diff[1] + sc[:, 2] * diff[2]
                    m[:, j] = (np.cos(k * sd) + 1j * sign * np.sin(k * sd)) * sw
            else:
                cnp = levels[li + 1].theta_points * levels[li + 1].phi_points if li + 1 < len(levels) else nsp
                m = np.zeros((nsp, len(cl.sons) * cnp), dtype=np.complex128)
                for ci, child_idx in enumerate(cl.sons):
                    ch = levels[li + 1].clusters[child_idx]
                    diff = np.asarray(ch.center) - np.asarray(cl.center)
                    sd = sc[:, 0] * diff[0] + sc[:, 1] * diff[1] + sc[:, 2] * diff[2]
                    col = (np.cos(k * sd) + 1j * sign * np.sin(k * sd)) * sw / float(cnp)
                    m[:, ci * cnp:(ci + 1) * cnp] = col[:, None]
            out.append(m)
        return out

    def matvec(self, x):                                                                  # :128-200
        x = np.asarray(x, dtype=np.complex128)
        y = np.zeros(self.num_dofs, dtype=np.complex128)
        if self.num_levels == 0:
            return y
        leaf_level = self.num_levels - 1
        nleaf = len(self.leaf_dof_indices)
        for (si, fi, B) in self.near:
            if si >= nleaf or fi >= nleaf:
                continue
            sd, fd = self.leaf_dof_indices[si], self.leaf_dof_indices[fi]
            yl = B @ x[fd] if len(fd) else np.zeros(len(sd), dtype=np.complex128)
            for li_, gi in enumerate(sd):
                if li_ < len(yl):
                    y[gi] += yl[li_]
            if si != fi:
                yf = B.T @ x[sd] if len(sd) else np.zeros(len(fd), dtype=np.complex128)
                for lj, gj in enumerate(fd):
                    if lj < len(yf):
                        y[gj] += yf[lj]
        if self.num_levels > 1 and len(self.t) > 0:
            mult = self._upward(x)
            loc = self._translate(mult)
            self._downward(loc)
            self._evaluate(loc, leaf_level, y)
        return y

    def _upward(self, x):                                                                 # :203-303
        mult = [[np.zeros(self.sphere_points_per_level[l], dtype=np.complex128) for _ in self.levels[l].clusters] for l in range(self.num_levels)]
        leaf = self.num_levels - 1
        for c in range(len(self.t[leaf])):
            if c >= len(self.leaf_dof_indices):
                continue
            dofs = self.leaf_dof_indices[c]
            if not dofs:
                continue
            T = self.t[leaf][c]
            if T.size == 0:
                continue
            xl = x[dofs]
            if len(xl) == T.shape[1]:
                mult[leaf][c] = T @ xl
        for l in range(leaf - 1, -1, -1):
            for c, cl in enumerate(self.levels[l].clusters):
                if c >= len(self.t[l]):
                    continue
                T = self.t[l][c]
                if T.size == 0 or not cl.sons:
                    continue
                cnp = self.sphere_points_per_level[l + 1]
                cm = np.zeros(len(cl.sons) * cnp, dtype=np.complex128)
                for s_i, ch in enumerate(cl.sons):
                    if ch < len(mult[l + 1]):
                        v = mult[l + 1][ch]
                        off = s_i * cnp
                        m = min(len(v), len(cm) - off)
                        if m > 0:
                            cm[off:off + m] = v[:m]
                if len(cm) == T.shape[1]:
                    mult[l][c] = T @ cm
        return mult

    def _translate(self, mult):                                                           # :306-359
        loc = [[np.zeros(self.sphere_points_per_level[l], dtype=np.complex128) for _ in self.levels[l].clusters] for l in range(self.num_levels)]
        for l, ent in enumerate(self.d):
            if l >= len(mult) or l >= len(loc):
                continue
            for (src, fld, diag) in ent:
                if src >= len(mult[l]) or fld >= len(loc[l]):
                    continue
                sm = mult[l][src]
                if len(sm) != len(diag):
                    continue
                m = min(len(diag), len(loc[l][fld]))
                loc[l][fld][:m] += diag[:m] * sm[:m]
        return loc

    def _downward(self, loc):                                                             # :362-417
        for l in range(max(self.num_levels - 1, 0)):
            for c, cl in enumerate(self.levels[l].clusters):
                if c >= len(self.s[l]) or not cl.sons:
                    continue
                S = self.s[l][c]
                if S.size == 0:
                    continue
                pl = loc[l][c]
                if len(pl) != S.shape[1]:
                    continue
                ch_loc = S @ pl
                cnp = self.sphere_points_per_level[l + 1]
                for s_i, ch in enumerate(cl.sons):
                    if ch >= len(loc[l + 1]):
                        continue
                    off = s_i * cnp
                    m = min(cnp, len(ch_loc) - off, len(loc[l + 1][ch]))
                    if m > 0:
                        loc[l + 1][ch][:m] += ch_loc[off:off + m]

    def _evaluate(self, loc, leaf, y):                                                    # :420-460
        for c in range(len(self.s[leaf])):
            if c >= len(self.leaf_dof_indices) or c >= len(loc[leaf]):
                continue
            dofs = self.leaf_dof_indices[c]
            if not dofs:
                continue
            S = self.s[leaf][c]
            if S.size == 0:
                continue
            le = loc[leaf][c]
            if len(le) != S.shape[1]:
                continue
            yl = S @ le
            for lj, gj in enumerate(dofs):
                if lj < len(yl) and gj < len(y):
                    y[gj] += yl[lj]
