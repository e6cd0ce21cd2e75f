"""GPU parity of the Quad4 path (ElementType::Quad4): bilinear panels, n x n Gauss rules by distance, quad-tree
subdivision, 4-edge singular term -- against the CPU restatement of regular.rs / singular.rs, on meshes of warped
(non-planar) quads and on a mixed Tri3 + Quad4 mesh."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from helpers import to_ma_mesh, k_from_ka, RADIUS

pytestmark = pytest.mark.gpu


def cube_sphere(radius, m, split_some=False):
    """Closed surface of 6 m^2 quads: a cube's face grids projected onto the sphere (warped bilinear quads).
    split_some: every third quad becomes two triangles (mixed mesh)."""
    idx = {}
    nodes = []

    def nid(p):
        key = tuple(np.round(p, 12))
        if key not in idx:
            idx[key] = len(nodes); nodes.append(p)
        return idx[key]
    conn = []
    g = np.linspace(-1.0, 1.0, m + 1)
    for axis in range(3):
        for sign in (-1.0, 1.0):
            for a in range(m):
                for b in range(m):
                    quad = []
                    for (u, v) in ((g[a], g[b]), (g[a + 1], g[b]), (g[a + 1], g[b + 1]), (g[a], g[b + 1])):
                        p = np.zeros(3); p[axis] = sign; p[(axis + 1) % 3] = u; p[(axis + 2) % 3] = v
                        quad.append(nid(radius * p / np.linalg.norm(p)))
                    if sign < 0:
                        quad = quad[::-1]
                    if split_some and (len(conn) % 3 == 0):
                        conn.append([quad[0], quad[1], quad[2], -1]); conn.append([quad[0], quad[2], quad[3], -1])
                    else:
                        conn.append(quad)
    return O.Mesh(np.array(nodes), np.array(conn, dtype=np.int32))


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("ka", [0.2, 2.5])
def test_quad_assembly_matches_oracle(gpu, mixed, ka):
    om = cube_sphere(RADIUS, 5, split_some=mixed)
    assert (om.conn[:, 3] >= 0).sum() > 0
    k = k_from_ka(ka)
    beta = complex(0.0, 4.0 / k)
    A_ref, rhs_ref = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    A, rhs = ma.assemble_tbem(to_ma_mesh(om), k, beta)
    assert np.all(np.isfinite(A.view(np.float64)))
    scale = np.abs(A_ref).max(axis=1, keepdims=True)
    err = np.abs(A - A_ref) / scale
    assert err.max() <= 1e-9, np.unravel_index(err.argmax(), err.shape)
    assert np.abs(rhs).max() == 0.0


def test_quad_near_list_leaf_counts_and_raw_integrals(gpu):
    om = cube_sphere(RADIUS, 6)
    n = om.n_elem
    k = k_from_ka(1.0)
    plan = ma.BemPlan(to_ma_mesh(om))
    near = plan.near_pairs()
    ref = set()
    for i in range(n):
        for j in range(n):
            if i != j:
                subs = O.generate_subelements(om.center[i], om.coords(j), om.area[j])
                if not (len(subs) == 1 and abs(subs[0].factor - 1.0) < 1e-10):
                    ref.add((i, j))
    assert set(map(tuple, near.tolist())) == ref
    rng = np.random.default_rng(3)
    pick = near[rng.choice(len(near), min(300, len(near)), replace=False)]
    out = plan.probe_pairs(k, pick.astype(np.int32))
    for q, (i, j) in enumerate(pick):
        r = O.regular_integration(om.center[i], om.normal[i], om.coords(j), om.area[j], k)[:4]
        assert int(round(out[q, 0].real)) == len(O.generate_subelements(om.center[i], om.coords(j), om.area[j]))
        assert np.all(np.abs(out[q, 1:5] - r) <= 1e-10 * np.abs(r).max()), (i, j)
    selfs = plan.probe_self(k)
    for e in range(0, n, 5):
        r = O.singular_integration(om.center[e], om.normal[e], om.coords(e), k)[:4]
        assert np.all(np.abs(selfs[e, 1:5] - r) <= 1e-10 * np.abs(r).max()), e
    plan.close()


def test_quad_very_near_point_hits_the_split_limit(gpu):
    """A collocation point hovering just above a big quad: the 15-splits-per-level limit of generate_subelements
    (singular.rs:556-562) drops pieces; the device must drop the same ones."""
    h = 0.02
    nodes = np.array([[1, 1, 0], [-1, 1, 0], [-1, -1, 0], [1, -1, 0],          # the big quad
                      [0.30, 0.20, h], [0.34, 0.20, h], [0.34, 0.24, h], [0.30, 0.24, h]], dtype=float)   # a tiny one above it
    om = O.Mesh(nodes, np.array([[0, 1, 2, 3], [4, 5, 6, 7]], dtype=np.int32))
    subs = O.generate_subelements(om.center[1], om.coords(0), om.area[0])
    assert len(subs) > 40
    plan = ma.BemPlan(to_ma_mesh(om))
    k = 3.0
    out = plan.probe_pairs(k, np.array([[1, 0]], dtype=np.int32))
    r = O.regular_integration(om.center[1], om.normal[1], om.coords(0), om.area[0], k)[:4]
    assert int(round(out[0, 0].real)) == len(subs)
    assert np.all(np.abs(out[0, 1:5] - r) <= 1e-10 * np.abs(r).max())
    plan.close()


@pytest.mark.parametrize("mixed", [False, True])
@pytest.mark.parametrize("ka", [0.2, 2.5])
def test_quad_matrix_free_operator_equals_dense_matvec(gpu, mixed, ka):
    """The streamed operator over Quad4 (and mixed) meshes: Quad4 columns stream the un-subdivided n x n rule the distance
    asks for (what the assembly's far kernel writes), Tri3 columns the 13-point rule; near pairs and the diagonal come in
    as corrections. Same entries as the dense matrix, different summation order -- for A x, row blocks, A^T x, A^H x and
    the diagonal preconditioner (fmm_interface.rs:177-212)."""
    om = cube_sphere(RADIUS, 6, split_some=mixed)
    n = om.n_elem
    k = k_from_ka(ka)
    beta = complex(0.0, 4.0 / k)
    mesh = to_ma_mesh(om)
    A, _ = ma.assemble_tbem(mesh, k, beta)
    plan = ma.BemPlan(mesh)
    op = ma.LinearOperator.tbem(plan, k, beta)
    i = np.arange(n)
    x = np.sin(0.1 * i) + 1j * np.cos(0.2 * i)
    y = op.apply(x)
    ref = A @ x
    assert np.all(np.isfinite(y.view(np.float64)))
    assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
    h = n // 3
    parts = np.zeros(n, dtype=complex)
    acc = np.zeros(n, dtype=complex)
    yt = op.apply_transpose(x)
    assert np.abs(yt - A.T @ x).max() <= 1e-12 * np.abs(A.T @ x).max()
    yh = op.apply_hermitian(x)
    assert np.abs(yh - A.conj().T @ x).max() <= 1e-12 * np.abs(A.conj().T @ x).max()
    for r0, r1 in ((0, h), (h, 2 * h), (2 * h, n)):
        blk = ma.LinearOperator.tbem(plan, k, beta, rows=(r0, r1))
        parts[r0:r1] = blk.apply(x)[r0:r1]
        acc += blk.apply_transpose(x)
        blk.close()
    assert np.array_equal(parts, y)
    assert np.abs(acc - yt).max() <= 1e-13 * np.abs(yt).max()
    Mp = ma.Preconditioner(op, kind="diagonal")
    z = Mp.apply(x)
    assert np.abs(z - x / np.diag(A)).max() <= 1e-12 * np.abs(z).max()
    b = ma.incident_rhs(om.center, om.normal, k, beta)
    xs, info = ma.gmres_preconditioned(op, Mp, b, restart=30, max_iterations=200, tol=1e-9)
    assert info.converged == 1
    xd = np.linalg.solve(A, b)
    assert np.linalg.norm(xs - xd) <= 1e-6 * np.linalg.norm(xd)
    Mp.close(); op.close(); plan.close()


@pytest.mark.parametrize("case", ["velocity_const", "nodal", "pressure_patch"])
def test_quad_boundary_values_reach_the_rhs(gpu, case):
    """rhs_contribution and free-term shares with Quad4 (and mixed) field panels: bilinear N_0..N_3, up to 4 values."""
    om = cube_sphere(RADIUS, 4, split_some=(case != "velocity_const"))
    n = om.n_elem
    nn = np.where(om.conn[:, 3] >= 0, 4, 3)
    rng = np.random.default_rng(8)
    if case == "velocity_const":
        om.bc_values[:, 0] = 1e-3 * (1.0 - 0.4j)
    elif case == "nodal":
        om.bc_len[:] = nn
        vals = 1e-3 * (rng.standard_normal((n, 4)) + 1j * rng.standard_normal((n, 4)))
        om.bc_values[:] = np.where(np.arange(4)[None, :] < nn[:, None], vals, 0.0)
    else:
        om.bc_type[10:40] = 1; om.bc_len[10:40] = nn[10:40]
        om.bc_values[10:40, :] = np.where(np.arange(4)[None, :] < nn[10:40, None], rng.standard_normal((30, 4)) + 0.2j, 0.0)
        om.bc_values[60:70, 0] = 2e-3
    for ka in (0.3, 2.2):
        k = k_from_ka(ka)
        beta = complex(0.0, 4.0 / k)
        A_ref, rhs_ref = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
        A, rhs = ma.assemble_tbem(to_ma_mesh(om), k, beta)
        assert np.abs(rhs_ref).max() > 0
        assert np.abs(rhs - rhs_ref).max() <= 1e-10 * np.abs(rhs_ref).max()
        assert (np.abs(A - A_ref) / np.abs(A_ref).max(axis=1, keepdims=True)).max() <= 1e-9


def test_quad_field_evaluation_follows_the_reference_quirk(gpu):
    """integrate_element_field (pressure.rs:154-258) uses the triangle of a quad's first three nodes."""
    om = cube_sphere(RADIUS, 4, split_some=True)
    k = k_from_ka(1.0)
    rng = np.random.default_rng(2)
    ps = rng.standard_normal(om.n_elem) + 1j * rng.standard_normal(om.n_elem)
    ep = 2.5 * RADIUS * np.array([[0, 0, 1.0], [1.0, 0, 0], [0.6, -0.8, 0], [0.0, 0.6, 0.8]])
    plan = ma.BemPlan(to_ma_mesh(om))
    ref = O.compute_scattered_field(ep, om, ps, k)
    got = ma.scattered_field(plan, k, ep, ps)
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    plan.close()


def test_golden_mixed_mesh_fixture(gpu):
    """Device vs the committed fixture (tests/golden/bem_golden.npz: mixq_*)."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bem_golden.npz"))
    om = O.Mesh(G["mixq_nodes"], G["mixq_conn"])
    om.bc_type = G["mixq_bc_type"].copy(); om.bc_len = G["mixq_bc_len"].copy(); om.bc_values = G["mixq_bc_values"].copy()
    A, rhs = ma.assemble_tbem(to_ma_mesh(om), float(G["mixq_k"][0]), complex(G["mixq_beta"][0]))
    Ag = G["mixq_A"]
    assert (np.abs(A - Ag) / np.abs(Ag).max(axis=1, keepdims=True)).max() <= 1e-9
    assert np.abs(rhs - G["mixq_rhs"]).max() <= 1e-10 * np.abs(G["mixq_rhs"]).max()
