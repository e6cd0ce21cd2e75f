"""Test-side builder of AMG hierarchies for the F1M family (box, P1 Kuhn tets): the INPUT of AmgPreconditioner::apply.

The reference builds its hierarchy on the host (AmgPreconditioner::from_csr, amg.rs:276-372: strength, Ruge-Stuben / PMIS
coarsening, interpolation, Galerkin R A P) and that stays there; the device path and the restatement take the
levels as given. Here the levels come from the grid itself: every other node in each direction is a coarse point, P is
trilinear interpolation (rows sum to 1, <= 8 entries), R = P^T (amg.rs:326 transpose_csr) and A_c = R A P (amg.rs:329,
galerkin_product) -- the same operator algebra, with sorted columns and explicit zeros dropped.
"""
import numpy as np
import scipy.sparse as sp


def _interp_1d(nf):
    """(nf+1) fine nodes -> (nf//2+1) coarse nodes (nf even): linear interpolation."""
    nc = nf // 2
    rows, cols, vals = [], [], []
    for i in range(nf + 1):
        if i % 2 == 0:
            rows.append(i); cols.append(i // 2); vals.append(1.0)
        else:
            rows += [i, i]; cols += [i // 2, i // 2 + 1]; vals += [0.5, 0.5]
    return sp.csr_matrix((vals, (rows, cols)), shape=(nf + 1, nc + 1))


def csr_triplet(M, dtype=np.complex128):
    M = sp.csr_matrix(M); M.sort_indices()
    return M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(dtype)


def box_hierarchy(A, nx, ny, nz, nlevels):
    """A: scipy CSR (complex) on the (nx+1)(ny+1)(nz+1) grid, node index = k (ny+1)(nx+1) + j (nx+1) + i (fem.box_mesh_tetrahedra).
    Returns a list of dicts {A, P, R} of scipy CSR matrices (coarsest: A only)."""
    levels = []
    cur = sp.csr_matrix(A).astype(np.complex128)
    for _ in range(nlevels - 1):
        if nx % 2 or ny % 2 or nz % 2 or min(nx, ny, nz) < 2:
            break
        P = sp.kron(_interp_1d(nz), sp.kron(_interp_1d(ny), _interp_1d(nx))).tocsr().astype(np.complex128)
        R = P.T.tocsr()
        Ac = (R @ cur @ P).tocsr()
        Ac.eliminate_zeros(); Ac.sort_indices()
        levels.append({"A": cur, "P": P, "R": R})
        cur = Ac
        nx, ny, nz = nx // 2, ny // 2, nz // 2
    levels.append({"A": cur})
    return levels
