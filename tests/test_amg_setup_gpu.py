"""GPU path of AmgPreconditioner::from_csr (math-solvers/src/preconditioners/amg.rs:276-372) behind ma_precond_create_amg_from_csr:
the hierarchy the library builds (host side, the reference's steps in their order) against the restatement's, level by level; the
cycle over it against the restated cycle; the reference's own unit tests (amg.rs:1158-1266) and SolverType::GmresAmg
(math-fem/src/solver/mod.rs:667) through the device path."""
import numpy as np
import pytest
import scipy.sparse as sp
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import fem

pytestmark = pytest.mark.gpu

S = O.amg_setup_module()


def _xvec(n):
    i = np.arange(n)
    return np.sin(0.1 * i) + 1j * np.cos(0.2 * i)


def _helmholtz(nx, ny, nz, k):
    nodes, rp, ci, K, M = fem.helmholtz_box(nx, ny, nz)
    n = len(rp) - 1
    A = sp.csr_matrix((K - (k * k) * M, ci, rp), shape=(n, n)).astype(np.complex128)
    A.sort_indices()
    return A


def _laplacian(n):
    return S.to_scipy(S.laplacian_1d(n))


def _operator(A):
    A = sp.csr_matrix(A).astype(np.complex128); A.sort_indices()
    return ma.CsrOperator(A.indptr.astype(np.int64), A.indices.astype(np.int64), values=A.data)


def _oracle_config(c):
    return dict(coarsening=c.coarsening, interpolation=c.interpolation, smoother=c.smoother, cycle=c.cycle, strong_threshold=c.strong_threshold,
                max_levels=c.max_levels, coarse_size=c.coarse_size, num_pre_smooth=c.num_pre_smooth, num_post_smooth=c.num_post_smooth,
                jacobi_weight=c.jacobi_weight, trunc_factor=c.trunc_factor, max_interp_elements=c.max_interp_elements,
                aggressive_coarsening_levels=c.aggressive_coarsening_levels)


def _same_matrix(dev, ora, what):
    assert dev["shape"] == (ora.nr, ora.nc), what
    assert (dev["row_ptrs"] == np.array(ora.ptr)).all(), what + ": row pointers"
    assert (dev["col_indices"] == np.array(ora.col, dtype=np.int64)).all(), what + ": columns"
    v = np.array(ora.val, dtype=np.complex128)
    if len(v):
        assert np.abs(dev["values"] - v).max() <= 1e-14 * max(1e-300, np.abs(v).max()), what + ": values"


CASES = [
    ("laplacian 100, default", lambda: _laplacian(100), "default", {}),
    ("laplacian 300, default, coarse_size 10", lambda: _laplacian(300), "default", {"coarse_size": 10}),
    ("laplacian 100, Pmis", lambda: _laplacian(100), "default", {"coarsening": 1}),
    ("laplacian 200, for_bem", lambda: _laplacian(200), "for_bem", {}),
    ("helmholtz 6x6x6 k=1.832+0.01i, for_fem", lambda: _helmholtz(6, 6, 6, 1.832 + 0.01j), "for_fem", {}),
    ("helmholtz 8x6x5, for_parallel", lambda: _helmholtz(8, 6, 5, 1.832), "for_parallel", {}),
    ("helmholtz 6x6x6, for_difficult_problems (Extended)", lambda: _helmholtz(6, 6, 6, 1.832 + 0.01j), "for_difficult_problems", {}),
    ("helmholtz 7x5x6, Direct with truncation", lambda: _helmholtz(7, 5, 6, 0.5), "default", {"interpolation": 2, "trunc_factor": 0.3, "max_interp_elements": 2}),
    ("helmholtz 6x6x6, Standard with truncation, Hmis", lambda: _helmholtz(6, 6, 6, 1.0), "default", {"coarsening": 2, "trunc_factor": 0.2, "max_interp_elements": 3}),
]


@pytest.mark.parametrize("name,make,preset,overrides", CASES, ids=[c[0] for c in CASES])
def test_hierarchy_equals_the_restatement(gpu, name, make, preset, overrides):
    A = make()
    cfg = ma.AmgConfig.preset(preset, **overrides)
    op = _operator(A)
    amg = ma.AmgFromCsr(op, cfg)
    lv, gc, oc = S.from_csr(S.from_scipy(A), _oracle_config(cfg))
    d = amg.diagnostics()
    assert d["num_levels"] == len(lv)
    assert abs(d["grid_complexity"] - gc) <= 1e-15 * gc and abs(d["operator_complexity"] - oc) <= 1e-15 * oc and d["setup_time_ms"] >= 0.0
    assert d["level_dofs"] == [l["A"].nr for l in lv] and d["level_nnz"] == [l["A"].nnz() for l in lv]
    for l in range(len(lv)):
        dl = amg.level(l)
        _same_matrix(dl["A"], lv[l]["A"], "level %d A" % l)
        if lv[l]["P"] is None:
            assert dl["P"] is None and dl["R"] is None
        else:
            _same_matrix(dl["P"], lv[l]["P"], "level %d P" % l)
            _same_matrix(dl["R"], lv[l]["R"], "level %d R" % l)
    # the cycle over it: AmgPreconditioner::apply, against the restated cycle over the restated hierarchy
    r = _xvec(A.shape[0])
    z = amg.apply(r)
    sm = 0 if cfg.smoother == 3 else cfg.smoother
    ref = O.AmgHierarchy(O.amg_levels_as_triplets(lv), smoother=sm, jacobi_weight=cfg.jacobi_weight, num_pre_smooth=cfg.num_pre_smooth,
                         num_post_smooth=cfg.num_post_smooth, cycle=cfg.cycle).apply(r)
    assert np.isfinite(z).all()
    assert np.abs(z - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    amg.close(); op.close()


def test_reference_unit_tests_on_the_device(gpu):
    """amg.rs:1158-1266: creation (>= 2 levels, complexities >= 1), apply changes the vector, PMIS, the three smoothers, ten
    preconditioned Richardson steps reduce the residual of the 1-D Laplacian below a tenth, diagnostics."""
    op = _operator(_laplacian(100))
    amg = ma.AmgFromCsr(op)
    d = amg.diagnostics()
    assert d["num_levels"] >= 2 and d["grid_complexity"] >= 1.0 and d["operator_complexity"] >= 1.0
    assert len(d["level_dofs"]) == d["num_levels"] and len(d["level_nnz"]) == d["num_levels"] and d["setup_time_ms"] >= 0.0
    amg.close()
    pm = ma.AmgFromCsr(op, ma.AmgConfig.preset("default", coarsening=1))
    assert pm.info()["num_levels"] >= 2
    pm.close(); op.close()
    op = _operator(_laplacian(50))
    r = np.arange(50, dtype=np.complex128)
    for smoother in (0, 1, 2, 3):
        a = ma.AmgFromCsr(op, ma.AmgConfig.preset("default", smoother=smoother))
        z = a.apply(r)
        assert z.shape == r.shape and np.abs(z - r).sum() > 1e-10
        a.close()
    op.close()
    n = 64
    A = _laplacian(n); op = _operator(A); amg = ma.AmgFromCsr(op)
    b = np.sin(np.arange(n)).astype(np.complex128); x = np.zeros(n, dtype=np.complex128)
    r0 = np.linalg.norm(b - A @ x)
    for _ in range(10):
        x = x + amg.apply(b - A @ x)
    assert np.linalg.norm(b - A @ x) < 0.1 * r0
    amg.close(); op.close()


def test_gmres_amg_from_the_matrix(gpu):
    """SolverType::GmresAmg (math-fem/src/solver/mod.rs:667): AmgPreconditioner::from_csr + gmres_preconditioned on the F1M family
    (12 x 10 x 8 box, k = 1.832 + 0.01i): converges, in the restatement's number of iterations, to the restatement's solution; and
    in fewer iterations than without the preconditioner."""
    A = _helmholtz(12, 10, 8, 1.832 + 0.01j)
    n = A.shape[0]
    b = A @ _xvec(n)
    cfg = ma.AmgConfig.preset("for_parallel")
    op = _operator(A); lin = ma.LinearOperator.csr(op)
    amg = ma.AmgFromCsr(op, cfg)
    x, info = ma.gmres_preconditioned(lin, amg, b, restart=30, max_iterations=200, tol=1e-8)
    lv, _, _ = S.from_csr(S.from_scipy(A), _oracle_config(cfg))
    H = O.AmgHierarchy(O.amg_levels_as_triplets(lv), smoother=cfg.smoother, jacobi_weight=cfg.jacobi_weight, num_pre_smooth=cfg.num_pre_smooth,
                       num_post_smooth=cfg.num_post_smooth, cycle=cfg.cycle)
    xr, ir = H.gmres(b, restart=30, max_iterations=200, tol=1e-8)
    assert info.converged and ir.converged and info.iterations == ir.iterations
    assert np.abs(x - xr).max() <= 1e-8 * np.abs(xr).max()
    assert np.linalg.norm(A @ x - b) <= 1e-6 * np.linalg.norm(b)
    x0, i0 = ma.gmres(lin, b, restart=30, max_iterations=200, tol=1e-8)
    assert info.iterations < i0.iterations
    amg.close(); op.close()


def test_setup_follows_the_current_values_and_refuses_bad_input(gpu):
    """from_csr reads the operator's CURRENT values (K - k^2 M after set_wavenumber); a matrix at or below coarse_size gives one
    level (:296-298); bad enums and a NULL config are MA_ERR_INVALID."""
    nodes, rp, ci, K, M = fem.helmholtz_box(6, 5, 4)
    n = len(rp) - 1
    op = ma.CsrOperator(rp, ci, K=K, M=M)
    for k in (0.5, 1.832 + 0.01j):
        op.set_wavenumber(k)
        amg = ma.AmgFromCsr(op, ma.AmgConfig.preset("for_parallel", coarse_size=20))
        A = sp.csr_matrix((K - (k * k) * M, ci, rp), shape=(n, n)).astype(np.complex128); A.sort_indices()
        lv, _, _ = S.from_csr(S.from_scipy(A), _oracle_config(amg.config))
        assert amg.info()["num_levels"] == len(lv) >= 2
        _same_matrix(amg.level(1)["A"], lv[1]["A"], "k = %s" % k)
        amg.close()
    one = ma.AmgFromCsr(op, ma.AmgConfig.preset("default", coarse_size=n))
    assert one.info()["num_levels"] == 1 and one.info()["grid_complexity"] == 1.0
    r = _xvec(n)
    assert np.isfinite(one.apply(r)).all()
    one.close()
    for bad in ({"coarsening": 3}, {"interpolation": -1}, {"smoother": 4}, {"cycle": 3}, {"max_levels": 0}):
        with pytest.raises(ma.MaError):
            ma.AmgFromCsr(op, ma.AmgConfig.preset("default", **bad))
    op.close()


def test_full_size_hierarchy_properties(gpu):
    """BASELINE.json config #4 at full size (F1M family, 96^3 cells = 912 673 DoF, k = 1.832 + 0.01i), where the Python restatement
    would take hours: properties of from_csr that do not depend on the size. Level sizes shrink; P is the identity on its C-points
    (every coarse column is the single entry 1 of some row) and no row is longer than a row of A; R = P^T entry for
    entry; A_c x = R (A (P x)) for a fixed x up to the 1e-15 cut of CsrMatrix::matmul; GMRES + AMG converges on it."""
    from math_audio_amd import fem
    _, rp, ci, K, M = fem.helmholtz_box(96, 96, 96)
    n = len(rp) - 1
    assert n == 912673
    k = 1.832 + 0.01j
    op = ma.CsrOperator(rp, ci, K=K, M=M); op.set_wavenumber(k)
    amg = ma.AmgFromCsr(op, ma.AmgConfig.preset("for_parallel"))
    d = amg.diagnostics()
    assert d["num_levels"] >= 5 and d["level_dofs"][0] == n and all(a > b for a, b in zip(d["level_dofs"], d["level_dofs"][1:]))
    assert 1.0 < d["grid_complexity"] < 2.5 and 1.0 < d["operator_complexity"] < 4.0
    l0 = amg.level(0); l1 = amg.level(1)
    P = sp.csr_matrix((l0["P"]["values"], l0["P"]["col_indices"], l0["P"]["row_ptrs"]), shape=l0["P"]["shape"])
    R = sp.csr_matrix((l0["R"]["values"], l0["R"]["col_indices"], l0["R"]["row_ptrs"]), shape=l0["R"]["shape"])
    Ac = sp.csr_matrix((l1["A"]["values"], l1["A"]["col_indices"], l1["A"]["row_ptrs"]), shape=l1["A"]["shape"])
    A = sp.csr_matrix((K - (k * k) * M, ci, rp), shape=(n, n))
    nc = P.shape[1]
    assert P.shape == (n, nc) and R.shape == (nc, n) and Ac.shape == (nc, nc) and d["level_dofs"][1] == nc
    assert (R - P.T).nnz == 0                                            # transpose_csr, entry for entry
    ones_rows = np.flatnonzero((np.diff(P.indptr) == 1) & (P.data[P.indptr[:-1].clip(max=len(P.data) - 1)] == 1.0))
    cols_of_ones = P.indices[P.indptr[ones_rows]]
    assert len(np.unique(cols_of_ones)) == nc                            # every coarse column is some C-point's identity row
    assert np.diff(P.indptr).max() <= 14                                 # never more than the row's strong neighbours
    x = _xvec(nc)
    y = R @ (A @ (P @ x)); z = Ac @ x
    assert np.abs(y - z).max() <= 1e-12 * np.abs(z).max()
    b = A @ _xvec(n)
    r0 = np.linalg.norm(b)
    lin = ma.LinearOperator.csr(op)
    xg, info = ma.gmres_preconditioned(lin, amg, b, restart=30, max_iterations=300, tol=1e-8)
    assert info.converged and np.linalg.norm(A @ xg - b) <= 1e-6 * r0
    amg.close(); lin.close(); op.close()


def test_hierarchy_does_not_depend_on_the_host_thread_count(gpu):
    """MA_HOST_THREADS (default min(16, cores)): the setup's products run over row blocks on host threads from 8 192 rows on; every
    row's result sits in its own slot, so one thread and the default count build the SAME hierarchy, bit for bit (27 x 27 x 27 box:
    21 952 nodes, the sparse products of the Galerkin operator split in five blocks)."""
    import os
    A = _helmholtz(27, 27, 27, 1.832 + 0.01j)
    assert A.shape[0] >= 5 * 4096
    op = _operator(A)
    cfg = ma.AmgConfig.preset("for_parallel")
    amg = ma.AmgFromCsr(op, cfg)
    old = os.environ.get("MA_HOST_THREADS")
    os.environ["MA_HOST_THREADS"] = "1"
    try:
        amg1 = ma.AmgFromCsr(op, cfg)
    finally:
        if old is None:
            del os.environ["MA_HOST_THREADS"]
        else:
            os.environ["MA_HOST_THREADS"] = old
    d, d1 = amg.diagnostics(), amg1.diagnostics()
    assert d["num_levels"] == d1["num_levels"] >= 2 and d["level_dofs"] == d1["level_dofs"] and d["level_nnz"] == d1["level_nnz"]
    for l in range(d["num_levels"]):
        a, b = amg.level(l), amg1.level(l)
        for key in ("A", "P", "R"):
            if a[key] is None:
                assert b[key] is None
                continue
            assert a[key]["shape"] == b[key]["shape"] and (a[key]["row_ptrs"] == b[key]["row_ptrs"]).all()
            assert (a[key]["col_indices"] == b[key]["col_indices"]).all() and (a[key]["values"] == b[key]["values"]).all(), (l, key)
    r = _xvec(A.shape[0])
    assert (amg.apply(r) == amg1.apply(r)).all()
    amg.close(); amg1.close(); op.close()
