"""math-fem/tests/analytical_validation.rs:1237-1286 (`test_3d_plane_wave`) as the reference writes it, for either solver:
unit cube, box_mesh_tetrahedra(0,1,0,1,0,1,4,4,4), P1, k = 2, plane wave exp(i k.x) (theta = pi/4, phi = pi/3) imposed on every
boundary face by row elimination, f = 0, GMRES(restart 50, max 500, tol 1e-10), nodal relative L2 error < 0.05.
The system comes from the restatement in oracle/oracle_fem.py; `solve(row_ptr, col, val, rhs) -> (x, converged)` is the solver
under test (the oracle's GMRES on the CPU, the library's device GMRES on the GPU)."""
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def oracle_fem():
    spec = importlib.util.spec_from_file_location("oracle_fem", os.path.join(_HERE, "..", "oracle", "oracle_fem.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


RESTART, MAX_ITERATIONS, TOLERANCE, THRESHOLD = 50, 500, 1e-10, 0.05


def run(solve, n_cells=4, k=2.0):
    fem = oracle_fem()
    case = fem.plane_wave_3d_case(n_cells, k)
    x, converged = solve(case["row_ptr"], case["col"], case["val"], case["rhs"])
    assert converged, "GMRES should converge"
    err = fem.l2_error(case["nodes"], x, case["analytical"])
    assert err < THRESHOLD, "3D plane wave error %g should be < 0.05" % err
    return err, case, x
