"""GPU parity of field evaluation (postprocess/pressure.rs:81-258) and of the room-acoustics collocation
matrix (room_acoustics/solver.rs:448-493) against the CPU oracle."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from helpers import to_ma_mesh, k_from_ka, RADIUS

pytestmark = pytest.mark.gpu


def test_scattered_field_matches_oracle(gpu):
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(1.0)
    rng = np.random.default_rng(11)
    ps = rng.standard_normal(om.n_elem) + 1j * rng.standard_normal(om.n_elem)
    vs = np.zeros(om.n_elem, dtype=complex); vs[::7] = 0.3 - 0.2j           # a few non-zero velocities: single-layer branch
    th = np.linspace(0, np.pi, 37); ph = np.linspace(0, 2 * np.pi, 37)
    ep = np.stack([2 * RADIUS * np.sin(th) * np.cos(ph), 2 * RADIUS * np.sin(th) * np.sin(ph), 2 * RADIUS * np.cos(th)], axis=1)   # r = 2a
    plan = ma.BemPlan(to_ma_mesh(om))
    for v in (None, vs):
        ref = O.compute_scattered_field(ep, om, ps, k, surface_velocity=v)
        got = ma.scattered_field(plan, k, ep, ps, v)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    plan.close()


def test_scattered_field_of_solved_sphere_is_finite_and_decays(gpu):
    """compute_total_field usage (pressure.rs:273-311): scattered field of the solved ka = 0.2 sphere at r = 2a, 4a."""
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(0.2); beta, _ = O.beta_adaptive(k, RADIUS)
    mesh = to_ma_mesh(om)
    A, r0 = ma.assemble_tbem(mesh, k, beta)
    x = ma.zgesv(A, r0 + ma.incident_rhs(om.center, om.normal, k, beta))
    plan = ma.BemPlan(mesh)
    d = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
    p2 = ma.scattered_field(plan, k, 2 * RADIUS * d, x); p4 = ma.scattered_field(plan, k, 4 * RADIUS * d, x)
    assert np.all(np.isfinite(p2.view(float))) and np.all(np.abs(p4) < np.abs(p2))
    assert np.abs(p2 - O.compute_scattered_field(2 * RADIUS * d, om, x, k)).max() <= 1e-12 * np.abs(p2).max()
    plan.close()


@pytest.mark.parametrize("sub,k", [(1, 2.0), (2, 18.3)])
def test_room_collocation_matrix_matches_oracle(gpu, sub, k):
    om = O.icosphere(1.0, sub)
    ref = O.room_build_matrix(om.center, om.normal, om.area, k, nthreads=4)
    got = ma.room_build_matrix(om.center, om.normal, om.area, k)
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    assert np.array_equal(np.diag(got), np.diag(ref))


def _room_box(nx=6, ny=5, nz=4, lx=5.0, ly=4.0, lz=2.5, tri_every=4):
    """Room interior surface: box faces gridded into quads (RectangularRoom::generate_mesh, math-xem-common/src/geometry.rs:107-183),
    every `tri_every`-th quad split into two triangles; element normals as the node order gives them."""
    idx = {}; nodes = []; conn = []

    def nid(p):
        key = tuple(np.round(p, 12))
        if key not in idx:
            idx[key] = len(nodes); nodes.append(p)
        return idx[key]
    xs = np.linspace(0, lx, nx + 1); ys = np.linspace(0, ly, ny + 1); zs = np.linspace(0, lz, nz + 1)
    faces = []
    for i in range(nx):
        for j in range(ny):
            faces.append([(xs[i], ys[j], 0), (xs[i + 1], ys[j], 0), (xs[i + 1], ys[j + 1], 0), (xs[i], ys[j + 1], 0)])
            faces.append([(xs[i], ys[j], lz), (xs[i], ys[j + 1], lz), (xs[i + 1], ys[j + 1], lz), (xs[i + 1], ys[j], lz)])
    for i in range(nx):
        for k in range(nz):
            faces.append([(xs[i], 0, zs[k]), (xs[i], 0, zs[k + 1]), (xs[i + 1], 0, zs[k + 1]), (xs[i + 1], 0, zs[k])])
            faces.append([(xs[i], ly, zs[k]), (xs[i + 1], ly, zs[k]), (xs[i + 1], ly, zs[k + 1]), (xs[i], ly, zs[k + 1])])
    for j in range(ny):
        for k in range(nz):
            faces.append([(0, ys[j], zs[k]), (0, ys[j + 1], zs[k]), (0, ys[j + 1], zs[k + 1]), (0, ys[j], zs[k + 1])])
            faces.append([(lx, ys[j], zs[k]), (lx, ys[j], zs[k + 1]), (lx, ys[j + 1], zs[k + 1]), (lx, ys[j + 1], zs[k])])
    for q, f in enumerate(faces):
        ids = [nid(np.array(p, dtype=float)) for p in f]
        if q % tri_every == 0:
            conn.append([ids[0], ids[1], ids[2], -1]); conn.append([ids[0], ids[2], ids[3], -1])
        else:
            conn.append(ids)
    return np.array(nodes), np.array(conn, dtype=np.int32)


def test_room_element_data_and_adaptive_matrix_match_oracle(gpu):
    """element_center_and_normal / element_area / characteristic length (solver.rs:38-122, 600-611) and
    build_bem_matrix_adaptive (:500-597): collocation for far pairs, the double-layer part of the singular routine on
    the first three nodes for near pairs (also for quads -- the reference's ElementType::Tri3 quirk, :556)."""
    nodes, conn = _room_box()
    c, nr, a, cl = ma.room_element_data(nodes, conn)
    c0, n0, a0, l0 = O.room_element_data(nodes, conn)
    assert np.array_equal(c, c0) and np.array_equal(a, a0) and np.array_equal(cl, l0) and np.abs(nr - n0).max() <= 1e-15
    assert abs(a.sum() - 2 * (5 * 4 + 5 * 2.5 + 4 * 2.5)) < 1e-9
    for k in (0.7, 6.0):
        for adaptive in (True, False):
            ref = O.room_build_matrix_adaptive(nodes, conn, k, adaptive)
            got = ma.room_build_matrix_adaptive(nodes, conn, k, adaptive)
            assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()
        plain = ma.room_build_matrix(c, nr, a, k)
        assert np.abs(ma.room_build_matrix_adaptive(nodes, conn, k, False) - plain).max() <= 1e-12 * np.abs(plain).max()


def test_room_incident_derivative_and_field_pressure_match_oracle(gpu):
    nodes, conn = _room_box()
    c, nr, a, cl = ma.room_element_data(nodes, conn)
    n = len(a)
    k = 2 * np.pi * 250.0 / 343.0
    src = np.array([[1.2, 0.9, 1.1], [3.8, 3.1, 0.6]])
    rng = np.random.default_rng(4)
    for amp in (np.array([1.0, 0.35]), 0.5 + rng.random((2, n))):            # omnidirectional, or per-point (directivity x crossover)
        ref = O.room_incident_derivative(c, nr, src, amp, k)
        got = ma.room_incident_derivative(c, nr, src, amp, k)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    ps = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    pts = np.array([[2.5, 2.0, 1.25], [0.4, 3.5, 2.0], [4.6, 0.3, 0.2], [1.2, 0.9, 1.1]])     # the last one sits on a source: skipped term
    for amp in (np.array([1.0, 0.35]), 0.5 + rng.random((2, len(pts)))):
        ref = O.room_field_pressure(c, nr, a, ps, src, amp, pts, k)
        got = ma.room_field_pressure(c, nr, a, ps, src, amp, pts, k)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_incident_evaluate_and_total_field(gpu):
    """IncidentField::evaluate_pressure / evaluate_normal_derivative (incident.rs:93-280) and compute_total_field."""
    om = O.icosphere(RADIUS, 1)
    k = k_from_ka(1.3)
    rng = np.random.default_rng(1)
    pts = rng.standard_normal((50, 3)); nrm = rng.standard_normal((50, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    for kind, vec, amp in ((0, (0.6, 0.0, 0.8), 1.5 - 0.5j), (1, (0.2, -0.4, 3.0), 2.0 + 0j)):
        p, d = ma.incident_evaluate(pts, k, kind, vec, amp, normals=nrm)
        assert np.abs(p - O.incident_pressure(pts, k, kind, vec, amp)).max() <= 1e-13 * np.abs(p).max()
        assert np.abs(d - O.incident_normal_derivative(pts, nrm, k, kind, vec, amp)).max() <= 1e-13 * np.abs(d).max()
        assert np.array_equal(ma.incident_evaluate(pts, k, kind, vec, amp), p)
    plan = ma.BemPlan(to_ma_mesh(om))
    ps = rng.standard_normal(om.n_elem) + 0j
    ep = 3 * RADIUS * pts[:5] / np.linalg.norm(pts[:5], axis=1, keepdims=True)
    pi, psc = ma.total_field(plan, k, ep, ps)
    assert np.abs(pi - np.exp(1j * k * ep[:, 2])).max() <= 1e-13
    assert np.abs(psc - O.compute_scattered_field(ep, om, ps, k)).max() <= 1e-12 * np.abs(psc).max()
    plan.close()


def test_room_config_json_feeds_the_room_path(gpu):
    """RoomConfig JSON (math-xem-common/src/config.rs:583-604) -> RectangularRoom::generate_mesh -> the room-acoustics
    collocation path on the device, against the restatement: the reference's example_rectangular.json at its first frequency."""
    import os
    from math_audio_amd import io as mio
    cfg = mio.RoomConfig.from_file(os.path.join(os.path.dirname(__file__), "golden", "room_example_rectangular.json"))
    nodes, conn = cfg.generate_mesh()
    k = 2.0 * np.pi * cfg.generate_frequencies()[0] / 343.0
    c, nr, a, cl = ma.room_element_data(nodes, conn)
    c0, n0, a0, l0 = O.room_element_data(nodes, conn)
    assert np.array_equal(c, c0) and np.array_equal(nr, n0) and np.array_equal(a, a0) and np.array_equal(cl, l0)
    A = ma.room_build_matrix(c, nr, a, k)
    Ar = O.room_build_matrix(c0, n0, a0, k, nthreads=8)
    assert np.abs(A - Ar).max() <= 1e-12 * np.abs(Ar).max()
    src = np.array([cfg.sources[0]["position"]]); amp = np.array([cfg.sources[0]["amplitude"]])
    rhs = ma.room_incident_derivative(c, nr, src, amp, k)
    assert np.abs(rhs - O.room_incident_derivative(c0, n0, src, amp, k)).max() <= 1e-13 * np.abs(rhs).max()
