"""The high-level API mirror (math_audio_amd/bem_solver.py: BemProblem / BemSolver / BemSolution of math-bem/src/core/bem_solver.rs) on
the GPU: the flow of the reference's integration tests written as the reference writes it, the solver / assembly method switches,
and what the reference itself refuses."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import bem_solver as B
from math_audio_amd import mesh as mm
import reference_cases as RC
from test_reference_integration import OracleBackend, oracle_mie

pytestmark = pytest.mark.gpu


def test_solve_reads_like_the_reference_and_lands_on_its_thresholds(gpu):
    """test_accuracy_parity.rs:60-140 (Rayleigh, ka = 0.2): BemProblem::rigid_sphere_scattering_custom(a, f, c, rho, 8, 16),
    BemSolver::new().solve(&problem), solution.evaluate_pressure_field(points at r = 2a) against the Mie series, < 20 %."""
    case = next(c for c in RC.CASES if c["name"] == "parity_rayleigh_ka0.2")
    k_solver, k_mie = RC.case_wavenumbers(case)
    freq = k_solver * 343.0 / (2.0 * np.pi)
    problem = B.BemProblem.rigid_sphere_scattering_custom(RC.RADIUS, freq, 343.0, 1.21, case["mesh"][0], case["mesh"][1])
    solution = B.BemSolver().solve(problem)
    assert solution.num_dofs() == problem.mesh.n_elem and solution.max_surface_pressure() >= solution.mean_surface_pressure() > 0.0
    th, pts = RC._arc_points(case["r"] * RC.RADIUS, case["npts"])
    bem = solution.evaluate_pressure_field(pts)
    ana = oracle_mie(k_mie, RC.RADIUS, case["terms"], case["r"] * RC.RADIUS, th)
    worst = max(RC._rel(abs(b), abs(a)) for b, a in zip(bem, ana))
    assert worst < case["limit"]
    ref = RC.run_case(case, OracleBackend(), oracle_mie)                 # the same number as the restatement's flow
    assert abs(worst - ref) <= 1e-6
    assert abs(solution.evaluate_pressure(pts[3]) - bem[3]) <= 1e-12 * abs(bem[3])
    assert abs(problem.ka() - k_solver * RC.RADIUS) < 1e-12


def test_solver_and_assembly_switches(gpu):
    problem = B.BemProblem.rigid_sphere_scattering(0.1, 300.0, 343.0, 1.21)     # ka = 0.55: icosphere(2), 320 panels
    assert problem.mesh.n_elem == 320
    direct = B.BemSolver().solve(problem)
    for method in (B.SolverMethod.BiCgStab, B.SolverMethod.Cgs):                  # both run bicgstab (bem_solver.rs:459-474)
        it = B.BemSolver().with_solver_method(method).with_tolerance(1e-10).solve(problem)
        assert np.linalg.norm(it.surface_pressure - direct.surface_pressure) <= 1e-7 * np.linalg.norm(direct.surface_pressure)
    with pytest.raises(B.BemError) as e:                                           # :408-410
        B.BemSolver().with_assembly_method(B.AssemblyMethod.Mlfmm).solve(problem)
    assert e.value.kind == "NotImplemented"
    # SLFMM in the high-level API: one cluster, 6 x 12 points, 5 terms; Direct = LU of the extracted near-field matrix (:375-399, :481-489)
    s_dir = B.BemSolver().with_assembly_method(B.AssemblyMethod.Slfmm).solve(problem)
    s_it = B.BemSolver().with_assembly_method(B.AssemblyMethod.Slfmm).with_solver_method(B.SolverMethod.BiCgStab).with_tolerance(1e-10).solve(problem)
    assert np.linalg.norm(s_it.surface_pressure - s_dir.surface_pressure) <= 1e-6 * np.linalg.norm(s_dir.surface_pressure)
    om = O.icosphere(0.1, 2)
    from fmm_clusters import Clusters
    one = Clusters([[0.0, 0.0, 0.0]], [0, 320], np.arange(320), [0, 0], [], [0, 0], [])
    k = problem.physics.wave_number
    N = O.Slfmm(om, one, k, 6, 12, 5).near_matrix()
    rhs = O.compute_rhs_with_beta(om.center, om.normal, k, problem.physics.burton_miller_beta_scaled(4.0))
    xo, _, rc = O.zgesv(N, rhs, nthreads=4)
    assert rc == 0 and np.linalg.norm(s_dir.surface_pressure - xo) <= 1e-8 * np.linalg.norm(xo)
    # without Burton-Miller only the right-hand side changes (-gamma p_inc, incident.rs:303-308); the matrix keeps beta (:370-373)
    nb = B.BemSolver().solve(B.BemProblem.rigid_sphere_scattering(0.1, 300.0, 343.0, 1.21).with_burton_miller(False))
    A, r0 = O.build_tbem_system_with_beta(om, k, problem.physics.burton_miller_beta_scaled(4.0), nthreads=4)
    xr, _, rc = O.zgesv(A, r0 + O.compute_rhs_with_beta(om.center, om.normal, k, 0j), nthreads=4)
    assert rc == 0 and np.linalg.norm(nb.surface_pressure - xr) <= 1e-8 * np.linalg.norm(xr)
    with pytest.raises(B.BemError):
        B.BemSolver().solve(B.BemProblem.rigid_sphere_scattering(0.1, 300.0, 343.0, 1.21).with_boundary_condition(B.BoundaryConditionType.Impedance))


def test_soft_sphere(gpu):
    """BoundaryConditionType::Soft: Pressure(0) on every element (bem_solver.rs:336-339); against the restatement's system."""
    problem = B.BemProblem.rigid_sphere_scattering(0.1, 300.0, 343.0, 1.21).with_boundary_condition(B.BoundaryConditionType.Soft)
    sol = B.BemSolver().solve(problem)
    om = O.icosphere(0.1, 2)
    om.bc_type[:] = 1
    k = problem.physics.wave_number; beta = problem.physics.burton_miller_beta_scaled(4.0)
    A, r0 = O.build_tbem_system_with_beta(om, k, beta, nthreads=4)
    xr, _, rc = O.zgesv(A, r0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta), nthreads=4)
    assert rc == 0 and np.linalg.norm(sol.surface_pressure - xr) <= 1e-8 * np.linalg.norm(xr)
