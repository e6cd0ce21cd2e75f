"""TEST INFRASTRUCTURE (oracle): numpy / plain-Python restatement of the math-fem pieces behind the reference's
`test_3d_plane_wave` (math-fem/tests/analytical_validation.rs:1237-1286): mesh, P1-tetrahedron element matrices, the Helmholtz
triplets, Dirichlet row elimination and the triplet -> CSR conversion, each in the reference's loop order. Only tests/ may import
this; the product path (math_audio_amd/) never does. Sizes: a few hundred elements (plain loops).

  box_mesh_tetrahedra        math-fem/src/mesh/generators.rs:107-166 (Kuhn split, 6 tets per cube, node index k (ny+1)(nx+1) + j (nx+1) + i)
  boundary_nodes             math-fem/src/mesh/types.rs:357-401 (faces that belong to one element), :404-431 (tet faces)
  element_stiffness_tet_p1   math-fem/src/assembly/stiffness.rs:143-190 with gauss_tetrahedron(1) (quadrature/gauss.rs:199-204) and
                             Jacobian::from_3d / transform_gradient (basis/shape.rs:123-183)
  element_mass_tet_p1        math-fem/src/assembly/mass.rs:116-154 with gauss_tetrahedron(2) (gauss.rs:205-216)
  helmholtz_triplets         math-fem/src/assembly/helmholtz.rs:36-70 (all K entries, then all -k^2 M entries)
  apply_dirichlet            math-fem/src/boundary/dirichlet.rs:72-175
  to_csr_matrix              tests/analytical_validation.rs:20-30 = to_compressed (helmholtz.rs:78-108: sums per (row, col) in
                             triplet order, entries with |v| <= 1e-15 dropped) + CsrMatrix::from_triplets (sorted rows / columns)
  l2_error                   tests/analytical_validation.rs:33-57 (nodal, relative)
"""
import math

import numpy as np


def box_mesh_tetrahedra(x_min, x_max, y_min, y_max, z_min, z_max, nx, ny, nz):
    dx = (x_max - x_min) / nx
    dy = (y_max - y_min) / ny
    dz = (z_max - z_min) / nz
    nodes = []
    for k in range(nz + 1):
        for j in range(ny + 1):
            for i in range(nx + 1):
                nodes.append((x_min + i * dx, y_min + j * dy, z_min + k * dz))

    def node_idx(i, j, k):
        return k * (ny + 1) * (nx + 1) + j * (nx + 1) + i
    tets = []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                n000, n100, n010, n110 = node_idx(i, j, k), node_idx(i + 1, j, k), node_idx(i, j + 1, k), node_idx(i + 1, j + 1, k)
                n001, n101, n011, n111 = node_idx(i, j, k + 1), node_idx(i + 1, j, k + 1), node_idx(i, j + 1, k + 1), node_idx(i + 1, j + 1, k + 1)
                tets += [(n000, n100, n110, n111), (n000, n110, n010, n111), (n000, n010, n011, n111),
                         (n000, n011, n001, n111), (n000, n001, n101, n111), (n000, n101, n100, n111)]
    return np.array(nodes, dtype=np.float64), np.array(tets, dtype=np.int64)


def boundary_nodes(tets):
    """Nodes of the faces that appear in exactly one element (types.rs:357-401); tet faces as types.rs:418-424 lists them."""
    count = {}
    for v in tets:
        for f in ((v[0], v[1], v[2]), (v[0], v[1], v[3]), (v[0], v[2], v[3]), (v[1], v[2], v[3])):
            key = tuple(sorted(int(t) for t in f))
            count[key] = count.get(key, 0) + 1
    out = set()
    for key, c in count.items():
        if c == 1:
            out.update(key)
    return out


_GRAD_REF = ((-1.0, -1.0, -1.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))   # P1 tet, basis/shape.rs


def _jacobian_3d(coords):
    j = [[0.0] * 3 for _ in range(3)]
    for i, g in enumerate(_GRAD_REF):
        for k in range(3):
            j[0][k] += g[k] * coords[i][0]
            j[1][k] += g[k] * coords[i][1]
            j[2][k] += g[k] * coords[i][2]
    det = (j[0][0] * (j[1][1] * j[2][2] - j[1][2] * j[2][1]) - j[0][1] * (j[1][0] * j[2][2] - j[1][2] * j[2][0])
           + j[0][2] * (j[1][0] * j[2][1] - j[1][1] * j[2][0]))
    inv_det = 1.0 / det
    inverse = [[(j[1][1] * j[2][2] - j[1][2] * j[2][1]) * inv_det, (j[0][2] * j[2][1] - j[0][1] * j[2][2]) * inv_det, (j[0][1] * j[1][2] - j[0][2] * j[1][1]) * inv_det],
               [(j[1][2] * j[2][0] - j[1][0] * j[2][2]) * inv_det, (j[0][0] * j[2][2] - j[0][2] * j[2][0]) * inv_det, (j[0][2] * j[1][0] - j[0][0] * j[1][2]) * inv_det],
               [(j[1][0] * j[2][1] - j[1][1] * j[2][0]) * inv_det, (j[0][1] * j[2][0] - j[0][0] * j[2][1]) * inv_det, (j[0][0] * j[1][1] - j[0][1] * j[1][0]) * inv_det]]
    return det, inverse


def element_stiffness_tet_p1(coords):
    det, inverse = _jacobian_3d(coords)
    det_j = abs(det)
    weight = 1.0 / 6.0                                              # gauss_tetrahedron(1)
    grads = []
    for g in _GRAD_REF:                                             # transform_gradient: result_i = sum_j inverse[j][i] g_j
        grads.append([sum(inverse[j][i] * g[j] for j in range(3)) for i in range(3)])
    k_local = [[0.0] * 4 for _ in range(4)]
    for i in range(4):
        for j in range(4):
            dot = sum(a * b for a, b in zip(grads[i], grads[j]))
            k_local[i][j] += dot * det_j * weight
    return k_local


def element_mass_tet_p1(coords):
    det, _ = _jacobian_3d(coords)
    det_j = abs(det)
    a = (5.0 - math.sqrt(5.0)) / 20.0
    b = (5.0 + 3.0 * math.sqrt(5.0)) / 20.0
    w = 1.0 / 24.0
    m_local = [[0.0] * 4 for _ in range(4)]
    for (xi, eta, zeta) in ((a, a, a), (b, a, a), (a, b, a), (a, a, b)):
        values = (1.0 - xi - eta - zeta, xi, eta, zeta)
        for i in range(4):
            for j in range(4):
                m_local[i][j] += values[i] * values[j] * det_j * w
    return m_local


def helmholtz_triplets(nodes, tets, k):
    """HelmholtzMatrix::new: every stiffness triplet (element by element, local i then j), then every -k^2 mass triplet."""
    k_sq = complex(k) * complex(k)
    rows, cols, vals = [], [], []
    mass = []
    for v in tets:
        coords = [tuple(nodes[int(t)]) for t in v]
        ke = element_stiffness_tet_p1(coords)
        me = element_mass_tet_p1(coords)
        for i in range(4):
            for j in range(4):
                rows.append(int(v[i])); cols.append(int(v[j])); vals.append(complex(ke[i][j], 0.0))
                mass.append((int(v[i]), int(v[j]), -k_sq * complex(me[i][j], 0.0)))
    for (r, c, m) in mass:
        rows.append(r); cols.append(c); vals.append(m)
    return rows, cols, vals


def apply_dirichlet(n, rows, cols, vals, rhs, dirichlet):
    """dirichlet: {node: value}. Returns the new triplets and right-hand side (dirichlet.rs:72-175)."""
    rhs = list(rhs)
    correction = [0j] * n
    for k in range(len(vals)):
        col = cols[k]
        if col in dirichlet:
            row = rows[k]
            if row not in dirichlet:
                correction[row] += vals[k] * dirichlet[col]
    for i in range(n):
        rhs[i] -= correction[i]
    for node, value in dirichlet.items():
        rhs[node] = value
    new_rows, new_cols, new_vals = [], [], []
    added = set()
    for k in range(len(vals)):
        row, col = rows[k], cols[k]
        if row in dirichlet:
            if row == col and row not in added:
                new_rows.append(row); new_cols.append(col); new_vals.append(1.0 + 0j); added.add(row)
        elif col in dirichlet:
            continue
        else:
            new_rows.append(row); new_cols.append(col); new_vals.append(vals[k])
    for node in dirichlet:
        if node not in added:
            new_rows.append(node); new_cols.append(node); new_vals.append(1.0 + 0j); added.add(node)
    return new_rows, new_cols, new_vals, rhs


def to_csr_matrix(n, rows, cols, vals):
    entries = {}
    for r, c, v in zip(rows, cols, vals):
        entries[(r, c)] = entries.get((r, c), 0j) + v
    keys = sorted(k for k, v in entries.items() if abs(v) > 1e-15)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    for (r, _) in keys:
        row_ptr[r + 1] += 1
    row_ptr = np.cumsum(row_ptr)
    col = np.array([c for (_, c) in keys], dtype=np.int64)
    val = np.array([entries[k] for k in keys], dtype=np.complex128)
    return row_ptr, col, val


def l2_error(nodes, solution, analytical):
    error_sq = 0.0
    norm_sq = 0.0
    for i, p in enumerate(nodes):
        exact = analytical(p[0], p[1], p[2])
        diff = complex(solution[i]) - exact
        error_sq += diff.real * diff.real + diff.imag * diff.imag
        norm_sq += exact.real * exact.real + exact.imag * exact.imag
    return math.sqrt(error_sq / norm_sq) if norm_sq > 1e-15 else math.sqrt(error_sq)


def plane_wave_3d_case(n_cells=4, k=2.0):
    """test_3d_plane_wave as the reference writes it: the system, its right-hand side and the analytic field."""
    theta, phi = math.pi / 4.0, math.pi / 3.0
    kx = k * math.sin(theta) * math.cos(phi)
    ky = k * math.sin(theta) * math.sin(phi)
    kz = k * math.cos(theta)

    def plane_wave(x, y, z):
        phase = kx * x + ky * y + kz * z
        return complex(math.cos(phase), math.sin(phase))
    nodes, tets = box_mesh_tetrahedra(0.0, 1.0, 0.0, 1.0, 0.0, 1.0, n_cells, n_cells, n_cells)
    n = nodes.shape[0]
    rows, cols, vals = helmholtz_triplets(nodes, tets, complex(k, 0.0))
    rhs = [0j] * n                                                  # source f = 0
    dirichlet = {node: plane_wave(*nodes[node]) for node in sorted(boundary_nodes(tets))}
    rows, cols, vals, rhs = apply_dirichlet(n, rows, cols, vals, rhs, dirichlet)
    row_ptr, col, val = to_csr_matrix(n, rows, cols, vals)
    return {"nodes": nodes, "tets": tets, "row_ptr": row_ptr, "col": col, "val": val, "rhs": np.array(rhs, dtype=np.complex128),
            "analytical": plane_wave, "dirichlet": dirichlet}
