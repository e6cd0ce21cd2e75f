"""BASELINE config #2, literally and in one place (the reference's form: math-bem/bin/qa_suite.rs:199-326):
S1 = icosphere(0.1, 3) -- 1280 Tri3 panels --, rigid, ka = 1.0, c = 343, rho = 1.21, plane wave +z of amplitude 1, beta =
burton_miller_beta_adaptive(0.1). On the device: the whole system, the right-hand side, the dense solve (through the drop-in entry with
LAPACK's pivoting AND through the sweep's tournament plan) -- against the CPU restatement entry by entry, and against the Mie series at
the QA suite's threshold."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from helpers import to_ma_mesh, k_from_ka, RADIUS, rel_l2

pytestmark = pytest.mark.gpu


def test_config2_s1_device_against_restatement_and_mie(gpu):
    om = O.icosphere(RADIUS, 3)
    assert om.n_elem == 1280
    mesh = to_ma_mesh(om)
    k = k_from_ka(1.0)
    beta, sign = O.beta_adaptive(k, RADIUS)
    assert abs(beta - 4j / k) <= 1e-15 * abs(beta)                      # types.rs:183-194: ka >= 0.5 -> 4 i / k
    # --- the CPU restatement (tbem.rs:96-222, regular.rs, singular.rs; zgesv as LAPACK does it)
    A_ref, r0_ref = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    rhs_ref = r0_ref + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
    x_ref, _, rc = O.zgesv(A_ref, rhs_ref, nthreads=8)
    assert rc == 0
    # --- the device: assembly
    A, r0 = ma.assemble_tbem(mesh, k, beta)
    rhs = r0 + ma.incident_rhs(mesh.center, mesh.normal, k, beta)
    plan = ma.BemPlan(mesh)
    near = plan.near_pairs()
    mask = np.zeros(A.shape, dtype=bool)
    mask[near[:, 0], near[:, 1]] = True
    np.fill_diagonal(mask, True)
    scale = np.abs(A_ref).max(axis=1, keepdims=True)
    err = np.abs(A - A_ref) / scale
    assert err[~mask].max() <= 1e-11, ("far entries", err[~mask].max())
    assert err[mask].max() <= 1e-9, ("near / self entries", err[mask].max())
    assert (~mask).sum() > 1_500_000 and mask.sum() > 10_000           # both classes are populated at this size
    assert np.abs(rhs - rhs_ref).max() <= 1e-12 * np.abs(rhs_ref).max()
    # row sums of the exterior Burton-Miller operator on a closed surface (tbem.rs:487-520): the same global checksum on both sides
    assert np.abs(A.sum(axis=1) - A_ref.sum(axis=1)).max() <= 1e-9 * np.abs(A_ref).max()
    # --- the device: the solve, both pivoting modes, the restatement's system and the device's own
    for Amat, bvec in ((A_ref, rhs_ref), (A, rhs)):
        x_p = ma.zgesv(Amat, bvec)                                       # ma_zgesv: LAPACK's pivots (speculative panels verified)
        x_t = ma.zgesv(Amat, bvec, pivoting="tournament")
        assert rel_l2(x_p, x_ref) <= 1e-8 and rel_l2(x_t, x_ref) <= 1e-8
        assert rel_l2(x_p, x_t) <= 1e-12
    # --- the device end to end through the plan API (what the sweep runs per frequency)
    import torch
    dev = torch.device("cuda", 0)
    n = om.n_elem
    dA = torch.empty(n * n, dtype=torch.complex128, device=dev); dx = torch.empty(n, dtype=torch.complex128, device=dev)
    lu = ma.LuPlan(n, pivoting="tournament")
    st = lu.main_stream() or torch.cuda.current_stream().cuda_stream
    plan.assemble_dev(k, beta, dA.data_ptr(), dx.data_ptr(), stream=st)
    plan.incident_rhs_dev(k, beta, dx.data_ptr(), accumulate=True, stream=st)
    lu.factor_solve_dev(dA.data_ptr(), dx.data_ptr(), 1, st)
    assert lu.status(st) == ma.MA_OK
    x_dev = dx.cpu().numpy()
    acc, wid, rej = lu.speculation_stats()
    assert acc + wid + rej == 40 and rej <= 2, (acc, wid, rej)           # 1280 / 32 half-panels; an icosphere has no poles
    lu.close(); plan.close()
    assert rel_l2(x_dev, x_ref) <= 1e-8
    # --- against the Mie series at the collocation points, the QA suite's measure and threshold (qa_suite.rs:175-179, 216)
    r = np.linalg.norm(om.center, axis=1)
    theta = np.arccos(om.center[:, 2] / r)
    mie = np.array([O.sphere_scattering_3d(k, RADIUS, 50, [r[i]], [theta[i]])[0, 0] for i in range(n)])
    for xx in (x_dev, x_ref):
        assert rel_l2(xx, mie) < 0.30
    assert abs(rel_l2(x_dev, mie) - rel_l2(x_ref, mie)) <= 1e-8
