"""P1 tetrahedral Helmholtz matrices on a Kuhn-split box — input generator for the sparse FEM path.

What the reference assembles once per mesh before its frequency sweep (math-fem/src/mesh/generators.rs:107-166
box_mesh_tetrahedra; assembly/stiffness.rs:143-190 with the 1-point rule; assembly/mass.rs:116-154 with the
4-point rule; assembler.rs:60-212: K and M merged on ONE sorted pattern, explicit zeros kept). Vectorised
NumPy/SciPy host plumbing; it produces the synthetic input of BASELINE.json configs[3], nothing here is timed.
K_e[i][j] = V grad(phi_i).grad(phi_j); M_e[i][j] = V/20 (1 + delta_ij) (what the 4-point rule integrates exactly).
"""
import numpy as np
import scipy.sparse as sp


def box_mesh_tetrahedra(xmin, xmax, ymin, ymax, zmin, zmax, nx, ny, nz):
    dx, dy, dz = (xmax - xmin) / nx, (ymax - ymin) / ny, (zmax - zmin) / nz
    k, j, i = np.meshgrid(np.arange(nz + 1), np.arange(ny + 1), np.arange(nx + 1), indexing="ij")
    nodes = np.stack([xmin + i.ravel() * dx, ymin + j.ravel() * dy, zmin + k.ravel() * dz], axis=1)

    def idx(ii, jj, kk):
        return kk * (ny + 1) * (nx + 1) + jj * (nx + 1) + ii
    kc, jc, ic = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    ic, jc, kc = ic.ravel(), jc.ravel(), kc.ravel()
    n000, n100, n010, n110 = idx(ic, jc, kc), idx(ic + 1, jc, kc), idx(ic, jc + 1, kc), idx(ic + 1, jc + 1, kc)
    n001, n101, n011, n111 = idx(ic, jc, kc + 1), idx(ic + 1, jc, kc + 1), idx(ic, jc + 1, kc + 1), idx(ic + 1, jc + 1, kc + 1)
    tets = np.stack([np.stack(t, axis=1) for t in ((n000, n100, n110, n111), (n000, n110, n010, n111), (n000, n010, n011, n111),
                                                   (n000, n011, n001, n111), (n000, n001, n101, n111), (n000, n101, n100, n111))], axis=1)
    return nodes, tets.reshape(-1, 4)           # cube-major, the 6 Kuhn tets of a cube consecutive


def assemble_p1(nodes, tets):
    """Returns (row_ptr int64, col int64, K float64, M float64) on the shared sorted pattern."""
    n = nodes.shape[0]
    p = nodes[tets]                                        # (ne, 4, 3)
    J = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]], axis=2)   # columns = edges
    det = np.linalg.det(J)
    vol = np.abs(det) / 6.0
    Jinv = np.linalg.inv(J)
    gref = np.array([[-1.0, -1.0, -1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]])
    g = np.einsum("ad,edk->eak", gref, Jinv)               # physical gradients (ne, 4, 3)
    Ke = np.einsum("eak,ebk->eab", g, g) * vol[:, None, None]
    Me = (np.ones((4, 4)) + np.eye(4))[None, :, :] * (vol / 20.0)[:, None, None]
    rows = np.repeat(tets, 4, axis=1).ravel()
    cols = np.tile(tets, (1, 4)).ravel()
    K = sp.coo_matrix((Ke.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    M = sp.coo_matrix((Me.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    K.sum_duplicates(); M.sum_duplicates(); K.sort_indices(); M.sort_indices()
    # K and M come from the same (row, col) triplet list, so they share one sorted pattern; sums that are
    # exactly zero stay stored, as in the reference's merged pattern (assembler.rs:101-212)
    if not (np.array_equal(K.indptr, M.indptr) and np.array_equal(K.indices, M.indices)):
        raise RuntimeError("K and M patterns differ")
    return K.indptr.astype(np.int64), K.indices.astype(np.int64), K.data.copy(), M.data.copy()


def helmholtz_box(nx, ny, nz, lx=5.0, ly=4.0, lz=2.5):
    """The F1M family of SURVEY §8: box 5 x 4 x 2.5 m, (nx+1)(ny+1)(nz+1) nodes, 6 nx ny nz tets."""
    nodes, tets = box_mesh_tetrahedra(0.0, lx, 0.0, ly, 0.0, lz, nx, ny, nz)
    return (nodes,) + assemble_p1(nodes, tets)


def boundary_nodes(tets):
    """Nodes on faces that belong to exactly one tetrahedron (Mesh::detect_boundaries, math-fem/src/mesh/types.rs:357-401)."""
    faces = np.concatenate([tets[:, [0, 1, 2]], tets[:, [0, 1, 3]], tets[:, [0, 2, 3]], tets[:, [1, 2, 3]]], axis=0)
    faces = np.sort(faces, axis=1)
    uniq, counts = np.unique(faces, axis=0, return_counts=True)
    return np.unique(uniq[counts == 1].ravel())


def apply_dirichlet_csr(row_ptr, col, values, rhs, nodes_idx, node_values):
    """apply_dirichlet (math-fem/src/boundary/dirichlet.rs:72-175) on a CSR system: b_j -= A_ji g_i for free rows j, Dirichlet
    rows become identity rows with b_i = g_i, Dirichlet columns are dropped. Returns (row_ptr, col, values, rhs) of the new system;
    host plumbing in front of the device solve, like the generator above."""
    n = len(row_ptr) - 1
    A = sp.csr_matrix((np.asarray(values, dtype=np.complex128), np.asarray(col), np.asarray(row_ptr)), shape=(n, n))
    g = np.zeros(n, dtype=np.complex128); g[nodes_idx] = node_values
    fixed = np.zeros(n, dtype=bool); fixed[nodes_idx] = True
    b = np.asarray(rhs, dtype=np.complex128) - A @ g                  # g is zero on the free nodes: only Dirichlet columns contribute
    b[fixed] = g[fixed]
    keep = sp.diags((~fixed).astype(np.float64))
    B = (keep @ A @ keep + sp.diags(fixed.astype(np.complex128))).tocsr()
    B.eliminate_zeros(); B.sort_indices()
    return B.indptr.astype(np.int64), B.indices.astype(np.int64), B.data.astype(np.complex128), b
