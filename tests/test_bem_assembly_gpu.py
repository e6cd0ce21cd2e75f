"""GPU parity of the TBEM assembly (K1 far, K2 near, K3 self) against the CPU oracle.

Mirrors how the reference exercises build_tbem_system_with_beta: icosphere meshes from
generate_icosphere_mesh, rigid BC, beta from burton_miller_beta_adaptive (bin/qa_suite.rs:216-228).
Tolerances (complex f64): entries <= 1e-11 (far) / 1e-9 (near, self)
relative to the row's largest entry.
"""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from helpers import to_ma_mesh, k_from_ka, rowscaled_maxerr, RADIUS

pytestmark = pytest.mark.gpu

TOL_FAR = 1e-11
TOL_NEAR = 1e-9


@pytest.mark.parametrize("sub,ka", [(1, 0.2), (2, 0.2), (2, 1.0), (2, 3.0)])
def test_assemble_matches_oracle(gpu, sub, ka):
    om = O.icosphere(RADIUS, sub)
    k = k_from_ka(ka)
    beta, _ = O.beta_adaptive(k, RADIUS)
    A_ref, rhs_ref = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    A, rhs = ma.assemble_tbem(to_ma_mesh(om), k, beta)
    assert A.shape == A_ref.shape
    assert np.all(np.isfinite(A.view(np.float64)))
    plan = ma.BemPlan(to_ma_mesh(om))
    near = plan.near_pairs()
    mask = np.zeros(A.shape, dtype=bool)
    mask[near[:, 0], near[:, 1]] = True
    np.fill_diagonal(mask, True)
    scale = np.abs(A_ref).max(axis=1, keepdims=True)
    err = np.abs(A - A_ref) / scale
    assert err[~mask].max() <= TOL_FAR, "far entries"
    assert err[mask].max() <= TOL_NEAR, "near/self entries"
    assert np.abs(rhs).max() == 0.0 and np.abs(rhs_ref).max() == 0.0
    plan.close()


def test_near_pair_list_is_the_oracles(gpu):
    """The level-0 split decision (singular.rs:553-556) must agree pair for pair."""
    om = O.icosphere(RADIUS, 2)
    plan = ma.BemPlan(to_ma_mesh(om))
    near = set(map(tuple, plan.near_pairs().tolist()))
    ref = set()
    for i in range(om.n_elem):
        for j in range(om.n_elem):
            if i == j:
                continue
            subs = O.generate_subelements(om.center[i], om.coords(j), om.area[j])
            if not (len(subs) == 1 and abs(subs[0].factor - 1.0) < 1e-10):
                ref.add((i, j))
    assert near == ref
    plan.close()


def test_raw_integrals_match_oracle(gpu):
    """IntegrationResult level (G, H, H^T, E and sub-element count) for far, near and self pairs."""
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(1.0)
    plan = ma.BemPlan(to_ma_mesh(om))
    near = plan.near_pairs()
    rng = np.random.default_rng(7)
    far = []
    nearset = set(map(tuple, near.tolist()))
    while len(far) < 200:
        i, j = rng.integers(0, om.n_elem, 2)
        if i != j and (int(i), int(j)) not in nearset:
            far.append((int(i), int(j)))
    pick = near[rng.choice(len(near), 300, replace=False)]
    pairs = np.array(far + pick.tolist(), dtype=np.int32)
    out = plan.probe_pairs(k, pairs)
    for q, (i, j) in enumerate(pairs):
        ref = O.regular_integration(om.center[i], om.normal[i], om.coords(j), om.area[j], k)[:4]
        nsub = len(O.generate_subelements(om.center[i], om.coords(j), om.area[j]))
        assert int(round(out[q, 0].real)) == nsub
        got = out[q, 1:5]
        tol = 1e-12 if q < len(far) else 1e-10
        assert np.all(np.abs(got - ref) <= tol * np.abs(ref).max()), (i, j, got, ref)
    selfs = plan.probe_self(k)
    for e in range(0, om.n_elem, 7):
        ref = O.singular_integration(om.center[e], om.normal[e], om.coords(e), k)[:4]
        got = selfs[e, 1:5]
        assert np.all(np.abs(got - ref) <= 1e-10 * np.abs(ref).max()), (e, got, ref)
    plan.close()


def test_incident_rhs_matches_oracle(gpu):
    om = O.icosphere(RADIUS, 2)
    k = k_from_ka(1.0)
    beta, _ = O.beta_adaptive(k, RADIUS)
    ref = O.compute_rhs_with_beta(om.center, om.normal, k, beta)
    got = ma.incident_rhs(om.center, om.normal, k, beta)
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    src = (0.05, -0.3, 0.4)
    ref = O.compute_rhs_with_beta(om.center, om.normal, k, beta, kind=1, vec=src, amp=2.0 - 1.0j)
    got = ma.incident_rhs(om.center, om.normal, k, beta, kind=1, vec=src, amp=2.0 - 1.0j)
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize("case", ["velocity_const", "velocity_nodal", "pressure", "mixed_patch"])
def test_boundary_values_reach_the_rhs(gpu, case):
    """Non-zero boundary values: free-term share (tbem.rs:273-304) + rhs_contribution of every pair
    (regular.rs:157-177, singular.rs:360-392) -- far, subdivided and self regimes -- against the CPU restatement.
    Covers the reference's quirks: a single value is weighted by N_0 only, and beta inside rhs_contribution is the
    PhysicsParams' own i/k whatever beta the system is built with."""
    om = O.icosphere(RADIUS, 2)                           # 320 panels
    n = om.n_elem
    rng = np.random.default_rng(5)
    bc_type = np.zeros(n, dtype=np.uint8); bc_len = np.ones(n, dtype=np.int32); bc_values = np.zeros((n, 4), dtype=np.complex128)
    if case == "velocity_const":
        bc_values[:, 0] = 1e-3 * (1.0 + 0.5j)
    elif case == "velocity_nodal":
        bc_len[:] = 3; bc_values[:, :3] = 1e-3 * (rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3)))
    elif case == "pressure":
        bc_type[:] = 1; bc_len[:] = 3; bc_values[:, :3] = rng.standard_normal((n, 3)) + 1j * rng.standard_normal((n, 3))
    else:                                                 # a vibrating patch, a pressure-release patch, the rest rigid; one transfer-admittance panel
        bc_values[:40, 0] = 2e-3; bc_type[40:60] = 1; bc_values[40:60, 0] = 0.3 - 0.1j; bc_type[100] = 2; bc_values[100, 0] = 9.0
    om.bc_type = bc_type; om.bc_len = bc_len; om.bc_values = bc_values
    mesh = to_ma_mesh(om)
    for ka in (0.2, 1.7):
        k = k_from_ka(ka)
        beta = complex(0.0, 4.0 / k)                      # burton_miller_beta_scaled(k, 4), types.rs:144-150
        A_ref, rhs_ref = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
        A, rhs = ma.assemble_tbem(mesh, k, beta)
        assert np.abs(rhs_ref).max() > 0
        assert np.abs(rhs - rhs_ref).max() <= 1e-10 * np.abs(rhs_ref).max()
        assert rowscaled_maxerr(A, A_ref) <= 1e-9


def test_self_term_fourth_tier(gpu):
    """QuadratureParams::for_ka's last tier (6, 7, 10, 4), singular.rs:74-81: element ka >= 2. The 10 000-panel sweep only
    reaches the third tier at 8 kHz; coarse icospheres at 8 kHz put every panel in the fourth (edge 55 mm and 28 mm,
    k = 146.5: element ka = 8.0 and 4.1). Raw self integrals, then the assembled system, against the CPU restatement."""
    k = O.wave_number(8000.0, 343.0)
    for sub in (1, 2):
        om = O.icosphere(RADIUS, sub)
        edge = np.array([np.mean([np.linalg.norm(om.coords(e)[a] - om.coords(e)[(a + 1) % 3]) for a in range(3)]) for e in range(om.n_elem)])
        assert (k * edge).min() >= 2.0
        mesh = to_ma_mesh(om)
        plan = ma.BemPlan(mesh)
        selfs = plan.probe_self(k)
        for e in range(om.n_elem):
            ref = O.singular_integration(om.center[e], om.normal[e], om.coords(e), k)[:4]
            ref4 = O.singular_integration(om.center[e], om.normal[e], om.coords(e), k, params=(6, 7, 10, 4))[:4]
            assert np.array_equal(ref, ref4)
            assert np.all(np.abs(selfs[e, 1:5] - ref) <= 1e-10 * np.abs(ref).max()), (sub, e)
        assert int(round(selfs[0, 0].real)) == 3 * 10 * 6 + 3 * 4 * 7 * 7         # edge points + sub-triangle points
        plan.close()
        beta = complex(0.0, 4.0 / k)
        A_ref, _ = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
        A, _ = ma.assemble_tbem(mesh, k, beta)
        assert rowscaled_maxerr(A, A_ref) <= 1e-9


@pytest.mark.parametrize("nf", [2, 3, 5])
def test_multi_frequency_assembly_equals_the_single_one(gpu, nf):
    """ma_bem_plan_assemble_multi_dev: nf systems of one mesh, the far pairs of up to three per pass (geometry of a quadrature point
    shared): every system equals the one ma_bem_plan_assemble_dev builds alone (same operations per system: to rounding of a
    differently scheduled sum, 1e-13 of the row scale) and the oracle's within the usual tolerances; crosses the sign switch at ka = 0.5."""
    import torch
    om = O.icosphere(RADIUS, 2)
    mesh = to_ma_mesh(om)
    plan = ma.BemPlan(mesh)
    n = om.n_elem
    dev = torch.device("cuda", 0)
    kas = [0.2, 0.45, 1.0, 3.0, 0.7][:nf]
    ks = [k_from_ka(ka) for ka in kas]
    betas = [O.beta_scaled(k, 4.0) for k in ks]
    As = [torch.zeros(n * n, dtype=torch.complex128, device=dev) for _ in range(nf)]
    rs = [torch.ones(n, dtype=torch.complex128, device=dev) for _ in range(nf)]
    plan.assemble_multi_dev(ks, betas, [a.data_ptr() for a in As], [r.data_ptr() for r in rs])
    torch.cuda.synchronize()
    near = plan.near_pairs()
    mask = np.zeros((n, n), dtype=bool); mask[near[:, 0], near[:, 1]] = True; np.fill_diagonal(mask, True)
    for f in range(nf):
        A1 = torch.zeros(n * n, dtype=torch.complex128, device=dev); r1 = torch.ones(n, dtype=torch.complex128, device=dev)
        plan.assemble_dev(ks[f], betas[f], A1.data_ptr(), r1.data_ptr())
        torch.cuda.synchronize()
        Am = As[f].cpu().numpy().reshape(n, n); As1 = A1.cpu().numpy().reshape(n, n)
        scale = np.abs(As1).max(axis=1, keepdims=True)
        assert (np.abs(Am - As1) / scale).max() <= 1e-13
        assert np.abs(rs[f].cpu().numpy()).max() == 0.0
        A_ref, _ = O.build_tbem_system_with_beta(om, ks[f], betas[f], nthreads=8)
        err = np.abs(Am - A_ref) / np.abs(A_ref).max(axis=1, keepdims=True)
        assert err[~mask].max() <= TOL_FAR and err[mask].max() <= TOL_NEAR
    plan.close()


@pytest.mark.parametrize("nparts", [2, 5, 12])
def test_assembly_in_pieces_is_the_whole_one(gpu, nparts):
    """ma_bem_plan_assemble_multi_part_dev: the far pairs' rows in `nparts` slices (the first piece prepares the right-hand sides,
    the last runs the near and self pairs): issued in order on one stream they leave exactly the bits of assemble_multi_dev."""
    import torch
    om = O.icosphere(RADIUS, 2)
    mesh = to_ma_mesh(om)
    n = om.n_elem
    plan = ma.BemPlan(mesh)
    dev = torch.device("cuda", 0)
    ks = [k_from_ka(ka) for ka in (0.2, 1.0, 3.0)]
    betas = [O.beta_scaled(k, 4.0) for k in ks]
    whole = [torch.full((n * n,), 7.0 + 1j, dtype=torch.complex128, device=dev) for _ in ks]
    wr = [torch.full((n,), 3.0, dtype=torch.complex128, device=dev) for _ in ks]
    plan.assemble_multi_dev(ks, betas, [a.data_ptr() for a in whole], [r.data_ptr() for r in wr])
    parts = [torch.full((n * n,), -5.0 + 2j, dtype=torch.complex128, device=dev) for _ in ks]
    pr = [torch.full((n,), -1.0, dtype=torch.complex128, device=dev) for _ in ks]
    for p in range(nparts):
        plan.assemble_multi_part_dev(ks, betas, [a.data_ptr() for a in parts], [r.data_ptr() for r in pr], p, nparts)
    torch.cuda.synchronize()
    for a, b in zip(whole, parts):
        assert torch.equal(a, b)
    for a, b in zip(wr, pr):
        assert torch.equal(a, b)
    with pytest.raises(ma.MaError):
        plan.assemble_multi_part_dev(ks, betas, [a.data_ptr() for a in parts], [r.data_ptr() for r in pr], nparts, nparts)
