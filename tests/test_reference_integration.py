"""The reference's integration tests (math-bem/tests/test_accuracy_parity.rs, tests/test_bem_sphere_integration.rs)
run through the CPU restatement: every threshold those files hold must be met by the oracle, which pins the
oracle's UV-sphere generator + TBEM assembly + LU + compute_total_field chain on reference-held numbers
(tests/reference_cases.py lists the cases with their file:line). The device path runs the same cases in
tests/test_reference_integration_gpu.py."""
import numpy as np
import pytest
import oracle_lib as O
import reference_cases as RC


class OracleBackend:
    def solve(self, n_theta, n_phi, k, beta):
        om = O.uv_sphere(RC.RADIUS, n_theta, n_phi)
        A, r0 = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
        rhs = r0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
        x, _, rc = O.zgesv(A, rhs, nthreads=4)
        assert rc == 0

        def total_field(points):
            return O.incident_pressure(points, k) + O.compute_scattered_field(points, om, x, k)
        return om.center, x, total_field


def oracle_mie(k, radius, terms, r, thetas):
    return O.sphere_scattering_3d(k, radius, terms, [r], list(thetas))[0]


@pytest.mark.parametrize("case", RC.CASES, ids=[c["name"] for c in RC.CASES])
def test_reference_threshold_met_by_the_restatement(case):
    err = RC.run_case(case, OracleBackend(), oracle_mie)
    assert np.isfinite(err)
    assert err < case["limit"], (case["name"], err, case["limit"])
