"""Host-side mirror of the reference's high-level API (math-bem/src/core/bem_solver.rs): BemProblem / BemSolver / BemSolution.

The orchestration is the reference's, step by step (`BemSolver::solve`, :273-324): prepare_elements (:327-361) -> assemble_system
(:364-412) -> add_incident_field_rhs (:415-447) -> solve_dense_system / solve_fmm_system (:450-496); every numeric step runs on the
GPU through the C-ABI (assembly, incident right-hand side, LU, BiCGSTAB, the single-level operator, field evaluation). There is no
CPU fallback: without the library or a GPU every call raises."""
import enum
import math
import numpy as np
import math_audio_amd as ma
from . import mesh as mm


class SolverMethod(enum.Enum):              # bem_solver.rs:49-59
    Direct = 0
    Cgs = 1
    BiCgStab = 2


class AssemblyMethod(enum.Enum):            # :61-71
    Tbem = 0
    Slfmm = 1
    Mlfmm = 2


class BoundaryConditionType(enum.Enum):     # :73-82
    Rigid = 0
    Soft = 1
    Impedance = 2


class BemError(RuntimeError):               # :566-587
    def __init__(self, kind, text):
        super().__init__("%s: %s" % (kind, text))
        self.kind = kind


class IncidentField:
    """incident.rs:17-39: plane_wave_z / plane_wave(direction, amplitude) / point_source(position, strength)."""

    def __init__(self, kind=0, vec=(0.0, 0.0, 1.0), amp=1.0):
        self.kind, self.vec, self.amp = kind, tuple(float(v) for v in vec), complex(amp)

    @staticmethod
    def plane_wave_z():
        return IncidentField()

    @staticmethod
    def plane_wave(direction, amplitude=1.0):
        d = [float(v) for v in direction]
        l = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        return IncidentField(0, (d[0] / l, d[1] / l, d[2] / l) if l > 1e-10 else (0.0, 0.0, -1.0), amplitude)

    @staticmethod
    def point_source(position, strength=1.0):
        return IncidentField(1, position, strength)


class PhysicsParams:
    """types.rs:16-219, what the solver reads: PhysicsParams::new(frequency, speed_of_sound, density, is_internal)."""

    def __init__(self, frequency, speed_of_sound, density, is_internal=False):
        self.frequency, self.speed_of_sound, self.density = float(frequency), float(speed_of_sound), float(density)
        self.omega = 2.0 * math.pi * self.frequency
        self.wave_number = self.omega / self.speed_of_sound
        self.harmonic_factor = 1.0
        self.tau = -1.0 if is_internal else 1.0

    def burton_miller_beta_scaled(self, scale):
        return mm.burton_miller_beta_scaled(self.wave_number, scale, self.harmonic_factor, self.tau)


class BemProblem:                           # :84-199
    def __init__(self, mesh, physics, incident_field=None, bc_type=BoundaryConditionType.Rigid, use_burton_miller=True):
        self.mesh, self.physics = mesh, physics
        self.incident_field = incident_field or IncidentField.plane_wave_z()
        self.bc_type, self.use_burton_miller = bc_type, use_burton_miller

    @staticmethod
    def rigid_sphere_scattering(radius, frequency, speed_of_sound, density):
        ka = 2.0 * math.pi * frequency / speed_of_sound * radius
        subdivisions = 2 if ka < 1.0 else (3 if ka < 5.0 else 4)      # :118-124
        return BemProblem(mm.generate_icosphere_mesh(radius, subdivisions), PhysicsParams(frequency, speed_of_sound, density, False))

    @staticmethod
    def rigid_sphere_scattering_custom(radius, frequency, speed_of_sound, density, n_theta, n_phi):
        return BemProblem(mm.generate_sphere_mesh(radius, n_theta, n_phi), PhysicsParams(frequency, speed_of_sound, density, False))

    def with_incident_field(self, field):
        self.incident_field = field
        return self

    def with_boundary_condition(self, bc_type):
        self.bc_type = bc_type
        return self

    def with_burton_miller(self, use_bm):
        self.use_burton_miller = use_bm
        return self

    def ka(self):                           # :181-198: wave number times the largest node radius
        n = np.asarray(self.mesh.nodes, dtype=np.float64)
        return self.physics.wave_number * float(np.sqrt((n * n).sum(axis=1)).max())


class BemSolution:                          # :499-563
    def __init__(self, surface_pressure, plan, incident_field, physics):
        self.surface_pressure, self._plan, self.incident_field, self.physics = surface_pressure, plan, incident_field, physics

    def evaluate_pressure_field(self, points):
        """compute_total_field(points, elements, nodes, surface_pressure, None, incident_field, physics): p_total per point."""
        f = self.incident_field
        p_inc, p_scat = ma.total_field(self._plan, self.physics.wave_number, np.asarray(points, dtype=np.float64), self.surface_pressure,
                                       kind=f.kind, vec=f.vec, amp=f.amp)
        return p_inc + p_scat

    def evaluate_pressure(self, point):
        return complex(self.evaluate_pressure_field(np.asarray([point], dtype=np.float64))[0])

    def max_surface_pressure(self):
        return float(np.abs(self.surface_pressure).max())

    def mean_surface_pressure(self):
        return float(np.abs(self.surface_pressure).sum() / len(self.surface_pressure))

    def num_dofs(self):
        return len(self.surface_pressure)


class BemSolver:                            # :201-265
    def __init__(self):
        self.solver_method, self.assembly_method = SolverMethod.Direct, AssemblyMethod.Tbem
        self.max_iterations, self.tolerance, self.verbose, self.beta_scale = 1000, 1e-8, False, 4.0

    def with_solver_method(self, m):
        self.solver_method = m
        return self

    def with_assembly_method(self, m):
        self.assembly_method = m
        return self

    def with_max_iterations(self, n):
        self.max_iterations = int(n)
        return self

    def with_tolerance(self, t):
        self.tolerance = float(t)
        return self

    def with_verbose(self, v):
        self.verbose = bool(v)
        return self

    def solve(self, problem):               # :273-324
        m, ph, f = problem.mesh, problem.physics, problem.incident_field
        k = ph.wave_number
        n = m.n_elem
        # prepare_elements (:327-361): one boundary condition for every element, dof i for element i
        if problem.bc_type == BoundaryConditionType.Impedance:
            raise BemError("NotImplemented", "VelocityWithAdmittance elements are outside what the device assembly takes")
        bc_type = np.full(n, 0 if problem.bc_type == BoundaryConditionType.Rigid else 1, dtype=np.uint8)
        mesh = ma.MeshArrays(m.nodes, m.conn, m.center, m.normal, m.area, dof=np.arange(n), bc_type=bc_type, bc_values=np.zeros((n, 4), dtype=np.complex128),
                             bc_len=np.ones(n, dtype=np.int32))
        plan = ma.BemPlan(mesh)
        beta = ph.burton_miller_beta_scaled(self.beta_scale)
        # add_incident_field_rhs (:415-447): -(gamma p_inc + beta tau dp_inc/dn), or -gamma p_inc without Burton-Miller
        rhs_beta = beta if problem.use_burton_miller else 0j
        inc = ma.incident_rhs(mesh.center, mesh.normal, k, rhs_beta, kind=f.kind, vec=f.vec, amp=f.amp, harmonic=ph.harmonic_factor, tau=ph.tau)
        if self.assembly_method == AssemblyMethod.Tbem:              # :370-374
            A, rhs = ma.assemble_tbem(mesh, k, beta, harmonic=ph.harmonic_factor, tau=ph.tau)
            rhs = rhs + inc
            if self.solver_method == SolverMethod.Direct:            # :456-458
                try:
                    x = ma.lu_solve(A, rhs)
                except ma.MaError as e:
                    raise BemError("SolverFailed", str(e))
            else:                                                      # :459-474: Cgs and BiCgStab both run bicgstab
                x, info = ma.bicgstab(ma.LinearOperator.dense(A), rhs, self.max_iterations, self.tolerance)
                if not info.converged:
                    raise BemError("SolverFailed", "BiCGSTAB did not converge: residual = %g" % info.residual)
        elif self.assembly_method == AssemblyMethod.Slfmm:           # :375-399: ONE cluster at the origin holding every element, 6 x 12 points, 5 terms
            class _One:
                center = np.zeros((1, 3)); elem_ptr = np.array([0, n], dtype=np.int32); elem_idx = np.arange(n, dtype=np.int32)
                near_ptr = np.zeros(2, dtype=np.int32); near_idx = np.zeros(0, dtype=np.int32); far_ptr = np.zeros(2, dtype=np.int32); far_idx = np.zeros(0, dtype=np.int32)
            op = ma.LinearOperator.slfmm(plan, _One, k, 6, 12, 5, harmonic=ph.harmonic_factor, tau=ph.tau)
            rhs = inc                                                  # SlfmmSystem.rhs is zero (:289)
            if self.solver_method == SolverMethod.Direct:            # :481-489
                if n > 2000:
                    raise BemError("NotImplemented", "Direct solver not available for large FMM problems")
                try:
                    x = ma.lu_solve(op.slfmm_near_matrix(), rhs)
                except ma.MaError as e:
                    raise BemError("SolverFailed", str(e))
            else:
                x, info = ma.bicgstab(op, rhs, self.max_iterations, self.tolerance)
                if not info.converged:
                    raise BemError("SolverFailed", "BiCGSTAB did not converge: residual = %g" % info.residual)
        else:                                                          # :408-410
            raise BemError("NotImplemented", "MLFMM not yet integrated in high-level API")
        return BemSolution(x, plan, f, ph)
