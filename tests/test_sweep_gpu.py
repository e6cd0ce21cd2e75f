"""GPU tests of the device-resident path (plan -> assemble -> incident RHS -> LU) at the reference's
QA configurations, against the committed golden vectors, and at BASELINE.json's full size (10 000
panels) through size-independent properties."""
import os
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import mesh as mm
from helpers import to_ma_mesh, k_from_ka, RADIUS, rel_l2

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "bem_golden.npz"))


def _device_solve(mesh, k, beta):
    import torch
    n = mesh.n_elem
    dev = torch.device("cuda", 0)
    plan = ma.BemPlan(mesh); lu = ma.LuPlan(n)
    A = torch.empty(n * n, dtype=torch.complex128, device=dev); x = torch.empty(n, dtype=torch.complex128, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    plan.assemble_dev(k, beta, A.data_ptr(), x.data_ptr(), stream=st)
    plan.incident_rhs_dev(k, beta, x.data_ptr(), accumulate=True, stream=st)
    Acopy = A.clone(); b = x.clone()
    lu.factor_solve_dev(A.data_ptr(), x.data_ptr(), 1, stream=st)
    assert lu.status(st) == ma.MA_OK
    plan.close(); lu.close()
    return Acopy.view(n, n), b, x


@pytest.mark.parametrize("ka,sub,tol", [(0.2, 2, 0.05), (1.0, 3, 0.30), (3.0, 3, 0.30)])
def test_qa_suite_on_device(gpu, ka, sub, tol):
    """bin/qa_suite.rs:199-326 end to end on the GPU: L2 vs the reference's Mie series below its threshold,
    and the solution within 1e-8 of the CPU restatement's (config #2)."""
    om = O.icosphere(RADIUS, sub)
    k = k_from_ka(ka)
    beta, _ = O.beta_adaptive(k, RADIUS)
    A, b, x = _device_solve(to_ma_mesh(om), k, beta)
    x = x.cpu().numpy()
    A_ref, r0 = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    rhs_ref = r0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
    x_ref, _, rc = O.zgesv(A_ref, rhs_ref, nthreads=8)
    assert rc == 0
    assert rel_l2(x, x_ref) <= 1e-8
    r = np.linalg.norm(om.center, axis=1); theta = np.arccos(om.center[:, 2] / r)
    mie = np.array([O.sphere_scattering_3d(k, RADIUS, 50, [r[i]], [theta[i]])[0, 0] for i in range(om.n_elem)])
    assert rel_l2(x, mie) < tol


def test_golden_vectors_on_device(gpu):
    om = O.Mesh(GOLD["ico1_nodes"], GOLD["ico1_conn"])
    for tag in ("ka1", "ka02"):
        k = float(GOLD["ico1_%s_k" % tag][0]); beta = complex(GOLD["ico1_%s_beta" % tag][0])
        A, rhs0 = ma.assemble_tbem(to_ma_mesh(om), k, beta)
        Ag = GOLD["ico1_%s_A" % tag]
        assert (np.abs(A - Ag) / np.abs(Ag).max(axis=1, keepdims=True)).max() <= 1e-9
        rhs = rhs0 + ma.incident_rhs(om.center, om.normal, k, beta)
        assert np.abs(rhs - GOLD["ico1_%s_rhs" % tag]).max() <= 1e-13 * np.abs(rhs).max()
        x = ma.zgesv(A, rhs)
        assert rel_l2(x, GOLD["ico1_%s_x" % tag]) <= 1e-9
    om2 = O.icosphere(RADIUS, 2)
    plan = ma.BemPlan(to_ma_mesh(om2))
    pairs = GOLD["ico2_pairs"]
    off = pairs[pairs[:, 0] != pairs[:, 1]]
    for tag in ("ka1", "ka3"):
        k = float(GOLD["ico2_%s_k" % tag][0])
        got = plan.probe_pairs(k, off)
        selfs = plan.probe_self(k)
        qo = 0
        for q, (i, j) in enumerate(pairs):
            ref = GOLD["ico2_%s_integrals" % tag][q][:4]
            if i == j:
                g = selfs[i, 1:5]
            else:
                g = got[qo, 1:5]
                assert int(round(got[qo, 0].real)) == int(GOLD["ico2_%s_nsub" % tag][q]); qo += 1
            assert np.all(np.abs(g - ref) <= 1e-10 * np.abs(ref).max()), (tag, i, j)
    plan.close()


def test_level_overflow_quirk_on_device(gpu):
    """Two parallel unit triangles 0.05 apart: the collocation point of one sits so close to the other that
    a subdivision level wants more than 15 splits; the reference abandons the rest of that level
    (singular.rs:556-562) and the device kernel must integrate exactly the same 109 leaves."""
    nodes = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 0.05], [1, 0, 0.05], [0, 1, 0.05],
                      [0.02, 0.01, 0.8], [1.02, 0.01, 0.8], [0.02, 1.01, 0.8]], dtype=float)
    conn = np.array([[0, 1, 2, -1], [3, 4, 5, -1], [6, 7, 8, -1]], dtype=np.int32)
    om = O.Mesh(nodes, conn)
    # collocation points are the stored centres; override them so that panel 1 sees panel 0 from (0.3, 0.3, 0.05)
    om.center[1] = [0.3, 0.3, 0.05]
    plan = ma.BemPlan(to_ma_mesh(om))
    k = 5.0
    got = plan.probe_pairs(k, np.array([[1, 0], [2, 0], [0, 2]], dtype=np.int32))
    for q, (i, j) in enumerate([(1, 0), (2, 0), (0, 2)]):
        ref = O.regular_integration(om.center[i], om.normal[i], om.coords(j), om.area[j], k)[:4]
        nsub = len(O.generate_subelements(om.center[i], om.coords(j), om.area[j]))
        assert int(round(got[q, 0].real)) == nsub
        assert np.all(np.abs(got[q, 1:5] - ref) <= 1e-10 * np.abs(ref).max())
    assert int(round(got[0, 0].real)) == 109          # the truncated level
    assert int(round(got[1, 0].real)) == 16           # two complete levels
    plan.close()


@pytest.mark.parametrize("fidx", [0, 32, 63])
def test_full_size_sphere_properties(gpu, fidx):
    """S10 (10 000 panels, BASELINE.json configs[2]) at the two ends and the middle of the 64-point sweep -- 100 Hz
    (ka = 0.18: dg_dn sign +1, tbem.rs:123; self terms in for_ka's first tier), 926 Hz, 8 kHz (ka = 14.7: sign -1, the
    equatorial triangles' self terms in the third tier): sampled rows against the CPU restatement, the self terms of
    sampled panels, residual and linearity of the device solve."""
    import torch
    mesh = mm.generate_sphere_mesh(RADIUS, 51, 100)
    n = mesh.n_elem
    f = mm.log_space(100.0, 8000.0, 64)[fidx]
    k = mm.wave_number(f); beta = mm.burton_miller_beta_scaled(k, 4.0)
    assert (k * RADIUS < 0.5) == (fidx == 0)
    A, b, x = _device_solve(mesh, k, beta)
    # (1) sampled rows vs the oracle (first, last, and three interior rows incl. polar caps)
    om = O.uv_sphere(RADIUS, 51, 100)
    assert np.abs(om.nodes - mesh.nodes).max() <= 2e-17
    om.nodes[:] = mesh.nodes; om.center[:] = mesh.center; om.normal[:] = mesh.normal; om.area[:] = mesh.area
    for r0 in (0, 137, 5000, 9999):
        S, _ = O.build_tbem_rows(om, k, beta, r0, r0 + 1)
        row = A[r0].cpu().numpy()
        assert np.abs(row - S[0]).max() <= 1e-9 * np.abs(S[0]).max()
    # the for_ka tier of the self terms (singular.rs:48-82) at this frequency: polar cap and equator panels
    tiers = set()
    plan = ma.BemPlan(mesh)
    selfs = plan.probe_self(k)
    for e in (0, 137, 4950, 5000, 9999):
        ka_el = k * np.mean([np.linalg.norm(om.coords(e)[a] - om.coords(e)[(a + 1) % 3]) for a in range(3)])
        tiers.add(0 if ka_el < 0.3 else 1 if ka_el < 1.0 else 2 if ka_el < 2.0 else 3)
        ref = O.singular_integration(om.center[e], om.normal[e], om.coords(e), k)[:4]
        assert np.all(np.abs(selfs[e, 1:5] - ref) <= 1e-10 * np.abs(ref).max()), e
    plan.close()
    assert tiers == ({1, 2} if fidx == 63 else {0}), tiers     # 8 kHz: 7600 panels in the second tier, 2400 (equator) in the third
    # (2) residual of the solve, computed on the device with the saved copy of A
    res = torch.linalg.norm(A @ x - b) / torch.linalg.norm(b)
    assert float(res) <= 1e-10
    # (3) linearity: solving for 2b - i b gives (2 - i) x
    lu = ma.LuPlan(n)
    A2 = A.clone().reshape(-1); b2 = ((2 - 1j) * b).clone()
    st = torch.cuda.current_stream().cuda_stream
    lu.factor_solve_dev(A2.data_ptr(), b2.data_ptr(), 1, stream=st)
    assert lu.status(st) == ma.MA_OK
    assert float(torch.linalg.norm(b2 - (2 - 1j) * x) / torch.linalg.norm(x)) <= 1e-10
    lu.close()


def test_solve_sweep_driver_matches_per_frequency_path(gpu):
    """ma_bem_solve_sweep = the drivers' loop (room_simulator_bem.rs:329-360): per frequency assembly + incident RHS + solve,
    several systems per interleaved batch, against the one-shot path and the CPU restatement."""
    om = O.icosphere(RADIUS, 2)
    mesh = to_ma_mesh(om)
    plan = ma.BemPlan(mesh)
    freqs = [150.0, 545.9, 900.0, 1400.0, 2100.0]
    X, st = ma.solve_sweep(plan, freqs, speed_of_sound=343.0, beta_scale=4.0, slots=2)
    assert np.all(st == ma.MA_OK)
    for fi, f in enumerate(freqs):
        k = O.wave_number(f, 343.0); beta = complex(0.0, 4.0 / k)
        A, r0 = ma.assemble_tbem(mesh, k, beta)
        x1 = ma.zgesv(A, r0 + ma.incident_rhs(om.center, om.normal, k, beta))
        assert rel_l2(X[fi], x1) <= 1e-10
    k = O.wave_number(freqs[1], 343.0); beta = complex(0.0, 4.0 / k)
    A_ref, rhs_ref = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    x_ref, _, rc = O.zgesv(A_ref, rhs_ref + O.compute_rhs_with_beta(om.center, om.normal, k, beta), nthreads=8)
    assert rc == 0 and rel_l2(X[1], x_ref) <= 1e-8
    plan.close()


def test_two_host_threads_one_frequency_each(gpu):
    """The reference's drivers solve frequencies from rayon workers (room_simulator_fem.rs:1143-1158; traits are Send + Sync):
    two host threads, each with its own plans and buffers, must reproduce the serial results bit for bit (the panel
    kernels of both go through the launcher's sequencer; errors are thread-local)."""
    import threading
    om = O.icosphere(RADIUS, 3)
    mesh = to_ma_mesh(om)
    ks = [k_from_ka(1.0), k_from_ka(3.0)]

    def solve(k):
        beta, _ = O.beta_adaptive(k, RADIUS)
        A, rhs0 = ma.assemble_tbem(mesh, k, beta)
        b = rhs0 + ma.incident_rhs(om.center, om.normal, k, beta)
        return ma.zgesv(A, b)

    serial = [solve(k) for k in ks]
    out = [None, None]; err = []

    def worker(t):
        try:
            for _ in range(3):
                out[t] = solve(ks[t])
        except Exception as e:       # surfaced below: a thread must not swallow a failure
            err.append(e)

    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for t in range(2):
        assert np.array_equal(out[t], serial[t])


def test_multi_device_sweep_inside_the_library(gpu):
    """ma_bem_solve_sweep_multi: frequency f on devices[f mod ndev], one host thread, BEM plan, LU plan and stream per device
    inside the library (SURVEY 8e.1 behind the C-ABI). On a one-GPU box the 'devices' are the same GPU twice / three times -- which
    only the DIAGNOSTIC build of the library accepts (MA_TEST_ALLOW_DUPLICATE_DEVICES; a process of its own): the sharding rule, the
    threads, the strided solution scatter and the per-frequency status are the multi-GPU code path; results must equal the
    single-device sweep (same kernels, same order per system). The shipped library refuses a device listed twice."""
    from test_lu_gpu import _run_with_diagnostic_library
    om = O.icosphere(RADIUS, 2)
    mesh = to_ma_mesh(om)
    freqs = [150.0, 545.9, 900.0, 1400.0, 2100.0, 2600.0, 3100.0]
    plan = ma.BemPlan(mesh)
    X1, s1 = ma.solve_sweep(plan, freqs, slots=2)
    plan.close()
    Xa, sa = ma.solve_sweep_multi(mesh, [0], freqs, slots=2)
    assert np.array_equal(Xa, X1) and np.array_equal(sa, s1)
    h = ma.BemSweepMulti(mesh, [0], len(freqs), slots=2)      # the reusable handle on the shipped library: one device
    for _ in range(2):
        Xh, sh = h.run(freqs)
        assert np.array_equal(Xh, X1) and np.array_equal(sh, s1)
    with pytest.raises(ma.MaError) as e:
        h.run(freqs + [3500.0])                               # more frequencies than the handle was made for
    assert e.value.status == ma.MA_ERR_INVALID
    h.close()
    code = r'''
import numpy as np
import oracle_lib as O
import math_audio_amd as ma
from helpers import to_ma_mesh, RADIUS, rel_l2
mesh = to_ma_mesh(O.icosphere(RADIUS, 2))
freqs = [150.0, 545.9, 900.0, 1400.0, 2100.0, 2600.0, 3100.0]
plan = ma.BemPlan(mesh)
X1, s1 = ma.solve_sweep(plan, freqs, slots=2)
plan.close()
for devs in ([0, 0], [0, 0, 0]):
    Xm, sm = ma.solve_sweep_multi(mesh, devs, freqs, slots=2)
    assert np.all(sm == ma.MA_OK)
    for f in range(len(freqs)):
        assert rel_l2(Xm[f], X1[f]) <= 1e-12, (devs, f)
    # the reusable handle: two runs (the second with fewer frequencies) on plans and sweep handles made once
    h = ma.BemSweepMulti(mesh, devs, len(freqs), slots=2)
    Xh, sh = h.run(freqs)
    Xh2, sh2 = h.run(freqs[:4])
    secs, cnt = h.last_timing()
    h.close()
    assert np.array_equal(Xh, Xm) and np.array_equal(Xh2, Xm[:4]) and np.all(sh == 0) and np.all(sh2 == 0)
    assert int(cnt.sum()) == 4 and np.all(secs[cnt > 0] > 0.0)
print("ok")
'''
    r = _run_with_diagnostic_library(code, {"MA_TEST_ALLOW_DUPLICATE_DEVICES": 1})
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    os.environ["MA_TEST_ALLOW_DUPLICATE_DEVICES"] = "1"           # the shipped library has no such switch
    try:
        with pytest.raises(ma.MaError) as e:
            ma.solve_sweep_multi(mesh, [0, 0], freqs)
        assert e.value.status == ma.MA_ERR_INVALID
    finally:
        del os.environ["MA_TEST_ALLOW_DUPLICATE_DEVICES"]
    with pytest.raises(ma.MaError) as e:
        ma.solve_sweep_multi(mesh, [ma.device_count()], freqs)
    assert e.value.status == ma.MA_ERR_INVALID


def test_sweep_runs_on_its_plans_device_not_the_callers(gpu):
    """ma_bem_solve_sweep took the calling thread's current device for the LU plan and the buffers. It now runs on
    the plan's device and restores the caller's; with one GPU the observable part is that the call leaves the current device
    and torch's current stream untouched and that the plan reports its device."""
    import torch
    om = O.icosphere(RADIUS, 1)
    plan = ma.BemPlan(to_ma_mesh(om))
    assert plan.device == 0
    before = torch.cuda.current_device()
    X, st = ma.solve_sweep(plan, [300.0, 600.0], slots=2)
    assert torch.cuda.current_device() == before and np.all(st == ma.MA_OK) and np.all(np.isfinite(X.view(np.float64)))
    plan.close()
