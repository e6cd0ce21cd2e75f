"""The reference's integration tests (tests/reference_cases.py: test_accuracy_parity.rs, test_bem_sphere_integration.rs)
through the device path: UV-sphere mesh -> ma_bem_solve_sweep (BemSolver::solve's assembly + incident RHS + lu_solve) ->
compute_total_field on the device. Every case must stay under the reference's own threshold AND reproduce the number the CPU
restatement measures for the same case (the measured errors are properties of the discretisation, not noise, so the two
paths must agree on them to 1e-6)."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import mesh as mm
import reference_cases as RC
from test_reference_integration import OracleBackend, oracle_mie

pytestmark = pytest.mark.gpu


class DeviceBackend:
    def solve(self, n_theta, n_phi, k, beta):
        mesh = mm.generate_sphere_mesh(RC.RADIUS, n_theta, n_phi)
        plan = ma.BemPlan(mesh)
        f = k * RC.C_SOUND / (2.0 * np.pi)
        X, st = ma.solve_sweep(plan, [f], speed_of_sound=RC.C_SOUND, beta_scale=4.0, slots=1)
        assert st[0] == ma.MA_OK
        x = X[0]

        def total_field(points):
            p_inc, p_sc = ma.total_field(plan, k, points, x)
            return p_inc + p_sc
        self.plan = plan
        return mesh.center, x, total_field


@pytest.mark.parametrize("case", RC.CASES, ids=[c["name"] for c in RC.CASES])
def test_reference_threshold_met_on_the_device(gpu, case):
    dev = DeviceBackend()
    err = RC.run_case(case, dev, oracle_mie)
    dev.plan.close()
    assert np.isfinite(err) and err < case["limit"], (case["name"], err, case["limit"])
    ref = RC.run_case(case, OracleBackend(), oracle_mie)
    assert abs(err - ref) <= 1e-6 * max(1.0, abs(ref)), (case["name"], err, ref)
