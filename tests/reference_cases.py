"""The reference's own end-to-end integration tests restated as data + one runner.

What: math-bem/tests/test_accuracy_parity.rs and tests/test_bem_sphere_integration.rs drive
`BemSolver::new().solve(&BemProblem::rigid_sphere_scattering_custom(..))` (bem_solver.rs:150-170,
273-322: UV sphere, rigid BC, plane wave +z, beta = 4 i / k, TBEM, lu_solve) and compare
`evaluate_pressure_field` (compute_total_field, postprocess/pressure.rs:273-311) or the surface pressure
with the Mie series of math-wave under the thresholds held in those files. They are the only
reference-held numbers that exercise UV-sphere + assembly + solve + field evaluation end to end, so they
pin the CPU restatement (tests/test_reference_integration.py) and the device path
(tests/test_reference_integration_gpu.py) alike.

`run_case(case, backend)` returns the measured quantity; `backend` supplies
solve(n_theta, n_phi, k, beta) -> (mesh centres, surface pressure, total_field(points) callable).
"""
import math
import numpy as np

RADIUS = 0.1
C_SOUND = 343.0


def _k_of_ka(ka):
    """`let k = ka / radius; let frequency = k * c / (2 pi)` then PhysicsParams::new recomputes
    k = 2 pi f / c (test_accuracy_parity.rs:69-70, types.rs:39-58)."""
    k = ka / RADIUS
    f = k * C_SOUND / (2.0 * math.pi)
    return 2.0 * math.pi * f / C_SOUND, k


def _arc_points(r, count):
    """theta_i = pi i / (count-1); points (r sin, 0, r cos) (test_accuracy_parity.rs:92-100)."""
    th = [math.pi * float(i) / float(count - 1) for i in range(count)]
    pts = np.array([[r * math.sin(t), 0.0, r * math.cos(t)] for t in th])
    return th, pts


# kind "field": max over the arc of | |p_bem| - |p_mie| | / |p_mie|        (relative_error, :29-35)
# kind "surface": closest-element surface |p| vs Mie at 1.001 a, |ana| < 0.1 skipped (:186-233)
# kind "field_skip_small": as field, entries with |ana| < 0.1 skipped              (:631-639)
# kind "phase": max wrapped phase difference                                       (:520-540)
# kind "ratio": forward / back ratio at 3a                                         (:417-452)
# kind "point": one point at theta = pi/4, r = 2a                                  (:340-402)
CASES = [
    # test_accuracy_parity.rs::test_accuracy_rayleigh_regime (:60-140): 8x16, r = 2a, 9 points, 30 terms, < 20 %
    dict(name="parity_rayleigh_ka0.1", ka=0.1, mesh=(8, 16), kind="field", r=2.0, npts=9, terms=30, limit=0.20),
    dict(name="parity_rayleigh_ka0.2", ka=0.2, mesh=(8, 16), kind="field", r=2.0, npts=9, terms=30, limit=0.20),
    dict(name="parity_rayleigh_ka0.3", ka=0.3, mesh=(8, 16), kind="field", r=2.0, npts=9, terms=30, limit=0.20),
    # ::test_accuracy_mie_regime (:151-254): 10x20, surface vs Mie at 1.001 a, 13 angles, 50 terms, < 30 %
    dict(name="parity_mie_ka1.0", ka=1.0, mesh=(10, 20), kind="surface", r=1.001, npts=13, terms=50, limit=0.30),
    dict(name="parity_mie_ka1.2", ka=1.2, mesh=(10, 20), kind="surface", r=1.001, npts=13, terms=50, limit=0.30),
    # ::test_accuracy_higher_frequency (:261-325): 12x24, ka = 2, r = 2a, 17 points, 50 terms, < 35 %
    dict(name="parity_high_ka2.0", ka=2.0, mesh=(12, 24), kind="field", r=2.0, npts=17, terms=50, limit=0.35),
    # ::test_mesh_convergence (:327-402): ka = 1, finest mesh 12x24 at theta = pi/4, < 25 %
    dict(name="parity_convergence_12x24", ka=1.0, mesh=(12, 24), kind="point", r=2.0, terms=50, limit=0.25),
    # ::test_forward_backscatter_ratio (:404-452): 10x20, ka = 1, r = 3a, 40 terms, ratio error < 50 %
    dict(name="parity_fwd_back_ratio", ka=1.0, mesh=(10, 20), kind="ratio", r=3.0, terms=40, limit=0.50),
    # ::test_pressure_phase (:458-548): 10x20, ka = 1, r = 2a, 9 points, 40 terms, < 45 degrees
    dict(name="parity_phase", ka=1.0, mesh=(10, 20), kind="phase", r=2.0, npts=9, terms=40, limit=math.pi / 4.0),
    # ::test_accuracy_summary (:556-690): r = 2a, 13 points, 50 terms, small references skipped; 10 % / 70 % / 35 %
    dict(name="parity_summary_rayleigh", ka=0.3, mesh=(8, 16), kind="field_skip_small", r=2.0, npts=13, terms=50, limit=0.10),
    dict(name="parity_summary_mie", ka=1.0, mesh=(10, 20), kind="field_skip_small", r=2.0, npts=13, terms=50, limit=0.70),
    dict(name="parity_summary_high", ka=2.0, mesh=(12, 24), kind="field_skip_small", r=2.0, npts=13, terms=50, limit=0.35),
    # test_bem_sphere_integration.rs::test_bem_vs_analytical_rayleigh (:24-117): f = 100 Hz, 6x12, 9 points, 20 terms, < 50 %
    dict(name="sphere_rayleigh_100Hz", freq=100.0, mesh=(6, 12), kind="field", r=2.0, npts=9, terms=20, limit=0.50),
    # ::test_bem_vs_analytical_mie (:122-204): f = 546 Hz, 8x16, 13 points, 30 terms, < 75 %
    dict(name="sphere_mie_546Hz", freq=546.0, mesh=(8, 16), kind="field", r=2.0, npts=13, terms=30, limit=0.75),
]


def case_wavenumbers(case):
    """(k the solver uses, k the test hands to the Mie series)."""
    if "freq" in case:          # test_bem_sphere_integration.rs:30: k = 2 pi f / c for both
        k = 2.0 * math.pi * case["freq"] / C_SOUND
        return k, k
    return _k_of_ka(case["ka"])


def _rel(computed, reference):
    return abs(computed) if abs(reference) < 1e-15 else abs(computed - reference) / abs(reference)


def run_case(case, backend, mie):
    """backend.solve(n_theta, n_phi, k, beta) -> (centers, surface_pressure, total_field(points));
    mie(k, radius, terms, r, thetas) -> complex array. Returns the measured error (compare with case['limit'])."""
    k_solver, k_mie = case_wavenumbers(case)
    beta = complex(0.0, 4.0 / k_solver)                    # BemSolver::default beta_scale = 4 (bem_solver.rs:225, 366)
    centers, p_surf, total_field = backend.solve(case["mesh"][0], case["mesh"][1], k_solver, beta)
    kind = case["kind"]
    r_eval = case["r"] * RADIUS
    if kind in ("field", "field_skip_small", "phase"):
        th, pts = _arc_points(r_eval, case["npts"])
        bem = total_field(pts)
        ana = mie(k_mie, RADIUS, case["terms"], r_eval, th)
        if kind == "phase":
            worst = 0.0
            for b, a in zip(bem, ana):
                d = abs(math.atan2(b.imag, b.real) - math.atan2(a.imag, a.real))
                if d > math.pi:
                    d = 2.0 * math.pi - d
                worst = max(worst, d)
            return worst
        worst = 0.0
        for b, a in zip(bem, ana):
            if kind == "field_skip_small" and abs(a) < 0.1:
                continue
            if "freq" in case:              # test_bem_sphere_integration.rs:91-95: reference below 1e-10 counts as 0 error
                e = abs(abs(b) - abs(a)) / abs(a) if abs(a) > 1e-10 else 0.0
            else:
                e = _rel(abs(b), abs(a))
            worst = max(worst, e)
        return worst
    if kind == "point":
        t = math.pi / 4.0
        pts = np.array([[r_eval * math.sin(t), 0.0, r_eval * math.cos(t)]])
        return _rel(abs(total_field(pts)[0]), abs(mie(k_mie, RADIUS, case["terms"], r_eval, [t])[0]))
    if kind == "ratio":
        pts = np.array([[0.0, 0.0, r_eval], [0.0, 0.0, -r_eval]])
        bem = total_field(pts)
        af = abs(mie(k_mie, RADIUS, case["terms"], r_eval, [0.0])[0]); ab = abs(mie(k_mie, RADIUS, case["terms"], r_eval, [math.pi])[0])
        assert abs(bem[0]) > 0.0 and abs(bem[1]) > 0.0
        return _rel(abs(bem[0]) / abs(bem[1]), af / ab)
    if kind == "surface":
        th = [math.pi * float(i) / float(case["npts"] - 1) for i in range(case["npts"])]
        ana = mie(k_mie, RADIUS, case["terms"], r_eval, th)
        rr = np.sqrt((centers ** 2).sum(axis=1))
        el_theta = np.arccos(centers[:, 2] / rr)
        worst = 0.0
        for t, a in zip(th, ana):
            best = int(np.argmin(np.abs(el_theta - t)))           # first minimum, as the strict `<` scan (:205-216)
            if abs(a) < 0.1:
                continue
            worst = max(worst, _rel(abs(p_surf[best]), abs(a)))
        return worst
    raise ValueError(kind)
