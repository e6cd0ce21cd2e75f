"""Deterministic test geometries and physics helpers on the host side of the C-ABI.

What: the sphere generators, element geometry and the scalar physics helpers the reference's
drivers call before the hot path (math-bem/src/core/mesh/generators.rs:29-228, 434-602;
core/types.rs:39-219; math-xem-common/src/types.rs:290-302). They are host plumbing that feeds
`ma_mesh_t`; nothing here is timed. Scalar libm calls (math.sin/cos/sqrt) are used on purpose so
that node coordinates agree bit for bit with a libm-based restatement.
"""
import math
import numpy as np

from . import MeshArrays


def element_geometry(nodes, conn):
    """compute_element_geometry (generators.rs:513-602): centre, area, normal flipped outward."""
    nodes = np.asarray(nodes, dtype=np.float64)
    conn = np.asarray(conn, dtype=np.int32)
    ne = conn.shape[0]
    center = np.zeros((ne, 3)); normal = np.zeros((ne, 3)); area = np.zeros(ne)
    for e in range(ne):
        c = conn[e]
        nn = 3 if c[3] < 0 else 4
        acc = [0.0, 0.0, 0.0]
        for i in range(nn):
            for j in range(3):
                acc[j] += nodes[c[i], j]
        cen = [acc[j] / float(nn) for j in range(3)]
        if nn == 3:
            a = [nodes[c[1], j] - nodes[c[0], j] for j in range(3)]
            b = [nodes[c[2], j] - nodes[c[0], j] for j in range(3)]
        else:
            a = [nodes[c[2], j] - nodes[c[0], j] for j in range(3)]
            b = [nodes[c[3], j] - nodes[c[1], j] for j in range(3)]
        cr = [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
        ln = math.sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2])
        area[e] = ln / 2.0
        nr = [0.0, 0.0, 0.0]
        if ln > 1e-15:
            nr = [cr[0] / ln, cr[1] / ln, cr[2] / ln]
        if nr[0] * cen[0] + nr[1] * cen[1] + nr[2] * cen[2] < 0.0:
            nr = [-nr[0], -nr[1], -nr[2]]
        center[e] = cen; normal[e] = nr
    return center, normal, area


def _finish(nodes, conn):
    nodes = np.array(nodes, dtype=np.float64)
    conn = np.array(conn, dtype=np.int32)
    center, normal, area = element_geometry(nodes, conn)
    return MeshArrays(nodes, conn, center, normal, area)


def generate_sphere_mesh(radius, n_theta, n_phi):
    """UV sphere (generators.rs:29-98): 2 + (n_theta-1) n_phi nodes, 2 n_phi (n_theta-1) Tri3."""
    nodes = [[0.0, 0.0, radius]]
    for i in range(1, n_theta):
        theta = math.pi * float(i) / float(n_theta)
        st, ct = math.sin(theta), math.cos(theta)
        for j in range(n_phi):
            phi = 2.0 * math.pi * float(j) / float(n_phi)
            nodes.append([radius * st * math.cos(phi), radius * st * math.sin(phi), radius * ct])
    nodes.append([0.0, 0.0, -radius])
    south = len(nodes) - 1
    conn = []
    for j in range(n_phi):
        conn.append([0, 1 + j, 1 + (j + 1) % n_phi, -1])
    for i in range(n_theta - 2):
        rs, nrs = 1 + i * n_phi, 1 + (i + 1) * n_phi
        for j in range(n_phi):
            jn = (j + 1) % n_phi
            n0, n1, n2, n3 = rs + j, rs + jn, nrs + j, nrs + jn
            conn.append([n0, n2, n1, -1])
            conn.append([n1, n2, n3, -1])
    lrs = 1 + (n_theta - 2) * n_phi
    for j in range(n_phi):
        conn.append([lrs + j, south, lrs + (j + 1) % n_phi, -1])
    return _finish(nodes, conn)


def generate_icosphere_mesh(radius, subdivisions):
    """Icosphere (generators.rs:110-228); midpoints numbered in order of first use."""
    phi = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [[-1.0, phi, 0.0], [1.0, phi, 0.0], [-1.0, -phi, 0.0], [1.0, -phi, 0.0], [0.0, -1.0, phi], [0.0, 1.0, phi],
             [0.0, -1.0, -phi], [0.0, 1.0, -phi], [phi, 0.0, -1.0], [phi, 0.0, 1.0], [-phi, 0.0, -1.0], [-phi, 0.0, 1.0]]
    for v in verts:
        ln = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        v[0] /= ln; v[1] /= ln; v[2] /= ln
    faces = [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
             [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7],
             [9, 8, 1]]
    for _ in range(subdivisions):
        cache = {}

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key in cache:
                return cache[key]
            m = [(verts[a][d] + verts[b][d]) / 2.0 for d in range(3)]
            ln = math.sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2])
            verts.append([m[0] / ln, m[1] / ln, m[2] / ln])
            cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            m01, m12, m20 = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, m01, m20], [b, m12, m01], [c, m20, m12], [m01, m12, m20]]
        faces = nf
    nodes = [[v[0] * radius, v[1] * radius, v[2] * radius] for v in verts]
    conn = [[f[0], f[1], f[2], -1] for f in faces]
    return _finish(nodes, conn)


def generate_box_mesh(lx, ly, lz, nx, ny, nz):
    """Closed box centred at the origin, faces gridded nx x ny x nz cells, every quad split into two Tri3 with
    outward winding (the synthetic cabinet of BASELINE.json configs[4]; face gridding as RectangularRoom::
    generate_mesh, math-xem-common/src/geometry.rs:107-183): 4 (nx ny + nx nz + ny nz) panels."""
    xs = [-lx / 2.0 + lx * float(i) / float(nx) for i in range(nx + 1)]
    ys = [-ly / 2.0 + ly * float(j) / float(ny) for j in range(ny + 1)]
    zs = [-lz / 2.0 + lz * float(k) / float(nz) for k in range(nz + 1)]
    index = {}
    nodes = []

    def nid(i, j, k):
        key = (i, j, k)
        if key not in index:
            index[key] = len(nodes); nodes.append([xs[i], ys[j], zs[k]])
        return index[key]
    conn = []

    def quad(a, b, c, d):          # a-b-c-d counter-clockwise seen from outside
        conn.append([a, b, c, -1]); conn.append([a, c, d, -1])
    for i in range(nx):
        for j in range(ny):
            quad(nid(i, j, 0), nid(i, j + 1, 0), nid(i + 1, j + 1, 0), nid(i + 1, j, 0))                      # z = -lz/2 (normal -z)
            quad(nid(i, j, nz), nid(i + 1, j, nz), nid(i + 1, j + 1, nz), nid(i, j + 1, nz))                  # z = +lz/2
    for i in range(nx):
        for k in range(nz):
            quad(nid(i, 0, k), nid(i + 1, 0, k), nid(i + 1, 0, k + 1), nid(i, 0, k + 1))                      # y = -ly/2
            quad(nid(i, ny, k), nid(i, ny, k + 1), nid(i + 1, ny, k + 1), nid(i + 1, ny, k))                  # y = +ly/2
    for j in range(ny):
        for k in range(nz):
            quad(nid(0, j, k), nid(0, j, k + 1), nid(0, j + 1, k + 1), nid(0, j + 1, k))                      # x = -lx/2
            quad(nid(nx, j, k), nid(nx, j + 1, k), nid(nx, j + 1, k + 1), nid(nx, j, k + 1))                  # x = +lx/2
    return _finish(nodes, conn)


# ---------------------------------------------------------------- physics (types.rs:39-219)
def wave_number(frequency, speed_of_sound=343.0):
    omega = 2.0 * math.pi * frequency
    return omega / speed_of_sound


def burton_miller_beta_scaled(k, scale, harmonic=1.0, tau=1.0):
    return complex(0.0, harmonic * scale / k) if tau > 0.0 else 0j


def burton_miller_beta_adaptive(k, radius, harmonic=1.0, tau=1.0):
    if tau <= 0.0:
        return 0j, 1.0
    ka = k * radius
    scale = 1.0 if ka < 0.5 else (4.0 if ka < 1.2 else (8.0 if ka < 1.8 else 16.0))
    return complex(0.0, harmonic * scale / k), scale


def log_space(start, end, num):
    """math-xem-common/src/types.rs:290-302."""
    if num < 2:
        return [start]
    ls, le = math.log(start), math.log(end)
    return [math.exp(ls + (le - ls) * float(i) / float(num - 1)) for i in range(num)]
