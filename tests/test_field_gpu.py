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
