"""GPU parity of AmgPreconditioner::apply / v_cycle (math-solvers/src/preconditioners/amg.rs:981-1065, 1068-1103) over host-built
hierarchies, against the CPU restatement, and of GMRES preconditioned by it (SolverType::GmresAmg, math-fem/src/solver/mod.rs:667:
`AmgPreconditioner::from_csr` + `gmres_preconditioned`) on the F1M family of BASELINE.json configs[3]."""
import numpy as np
import pytest
import scipy.sparse as sp
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import fem
from amg_hierarchy import box_hierarchy, csr_triplet

pytestmark = pytest.mark.gpu


def _xvec(n):
    i = np.arange(n)
    return np.sin(0.1 * i) + 1j * np.cos(0.2 * i)


def _helmholtz(nx, ny, nz, k):
    nodes, rp, ci, K, M = fem.helmholtz_box(nx, ny, nz)
    n = len(rp) - 1
    A = sp.csr_matrix((K - (k * k) * M, ci, rp), shape=(n, n))
    return A


def _device_levels(levels):
    out = []
    for lv in levels:
        d = {"A": ma.CsrOperator(*csr_triplet(lv["A"])[:2], values=csr_triplet(lv["A"])[2])}
        if "P" in lv:
            d["P"] = ma.CsrOperator.rect(lv["P"].shape[0], lv["P"].shape[1], *csr_triplet(lv["P"]))
            d["R"] = ma.CsrOperator.rect(lv["R"].shape[0], lv["R"].shape[1], *csr_triplet(lv["R"]))
        out.append(d)
    return out


def _oracle_levels(levels):
    return [{key: csr_triplet(lv[key]) for key in lv} for lv in levels]


def test_rectangular_operator_matvec(gpu):
    """The transfer operators are rectangular CSR handles (amg.rs:236-243): y = P x with x of ncols entries, against SciPy and
    the restatement's matvec; tall (P) and wide (R), with rows longer than the column count of a slice and empty rows."""
    rng = np.random.default_rng(3)
    for shape in ((1000, 130), (130, 1000), (70, 1), (3, 5000)):
        Mx = sp.random(shape[0], shape[1], density=min(1.0, 8.0 / shape[1]), random_state=rng.integers(1 << 30), format="csr").astype(np.complex128)
        Mx.data = rng.standard_normal(Mx.nnz) + 1j * rng.standard_normal(Mx.nnz)
        rp, ci, v = csr_triplet(Mx)
        h = ma.CsrOperator.rect(shape[0], shape[1], rp, ci, v)
        x = _xvec(shape[1])
        y = h.matvec(x)
        assert y.shape == (shape[0],)
        ref = O.csr_matvec(rp, ci, v, x)
        assert np.abs(y - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max())
        h.close()
    with pytest.raises(ma.MaError):
        ma.CsrOperator.rect(4, 3, [0, 1, 2, 3, 4], [0, 1, 2, 3], np.ones(4, dtype=complex))     # column 3 of a 4 x 3 operator


@pytest.mark.parametrize("smoother,cycle,pre,post", [("jacobi", "V", 1, 1), ("jacobi", "V", 2, 2), ("l1", "V", 1, 1), ("sgs", "V", 1, 1),
                                                     ("jacobi", "W", 1, 1), ("l1", "F", 2, 1)])
def test_v_cycle_matches_the_restatement(gpu, smoother, cycle, pre, post):
    """One preconditioner application z = M^-1 r, level by level the reference's sequence: pre-smooth, residual, restrict,
    recurse from zero, prolongate, correct, post-smooth; coarsest level 20 / 20 / 10 sweeps."""
    nx, ny, nz = 8, 8, 4
    k = complex(0.3, 0.01)                               # below the box's first mode: every level's Jacobi iteration contracts
    A = _helmholtz(nx, ny, nz, k)
    A = (A + 0.05 * sp.identity(A.shape[0])).tocsr()
    levels = box_hierarchy(A, nx, ny, nz, 3)
    assert len(levels) == 3 and levels[2]["A"].shape[0] == 3 * 3 * 2
    sm = {"jacobi": 0, "l1": 1, "sgs": 2}[smoother]; cy = {"V": 0, "W": 1, "F": 2}[cycle]
    w = 0.8 if pre == 2 else 0.6667                      # AmgConfig::for_parallel / default (amg.rs:150-203)
    ref = O.AmgHierarchy(_oracle_levels(levels), smoother=sm, jacobi_weight=w, num_pre_smooth=pre, num_post_smooth=post, cycle=cy)
    dl = _device_levels(levels)
    M = ma.AmgPreconditioner(dl, smoother=smoother, jacobi_weight=w, num_pre_smooth=pre, num_post_smooth=post, cycle=cycle)
    n = A.shape[0]
    for r in (_xvec(n), np.ones(n, dtype=complex)):
        z = M.apply(r); zr = ref.apply(r)
        assert np.all(np.isfinite(z.view(np.float64)))
        assert np.abs(z - zr).max() <= 1e-11 * np.abs(zr).max(), (smoother, cycle)
    # a single level degenerates to the coarsest-level smoother (amg.rs:985: level == levels.len() - 1)
    M1 = ma.AmgPreconditioner([{"A": dl[0]["A"]}], smoother=smoother, jacobi_weight=w)
    ref1 = O.AmgHierarchy([{"A": csr_triplet(levels[0]["A"])}], smoother=sm, jacobi_weight=w)
    r = _xvec(n)
    assert np.abs(M1.apply(r) - ref1.apply(r)).max() <= 1e-11 * np.abs(ref1.apply(r)).max()
    M.close(); M1.close()
    for lv in dl:
        for h in lv.values():
            h.close()


def test_v_cycle_at_the_wavenumber_of_config_4(gpu):
    """k = 2 pi 100 / 343 + 0.01 i on the coarse levels of this box is indefinite: the coarsest level's 20 Jacobi sweeps amplify
    (|z| ~ 1e9 |r|) -- the reference's algorithm does the same, which is why its FEM driver preconditions with the shifted
    Laplacian (math-fem/src/solver/mod.rs:1161-1290). Parity must hold on that growth too, relative to |z|."""
    nx, ny, nz = 16, 16, 8
    k = complex(2.0 * np.pi * 100.0 / 343.0, 0.01)
    A = _helmholtz(nx, ny, nz, k)
    levels = box_hierarchy(A, nx, ny, nz, 3)
    ref = O.AmgHierarchy(_oracle_levels(levels), smoother=0, jacobi_weight=0.8, num_pre_smooth=2, num_post_smooth=2)
    dl = _device_levels(levels)
    M = ma.AmgPreconditioner(dl, smoother="jacobi", jacobi_weight=0.8, num_pre_smooth=2, num_post_smooth=2)
    r = _xvec(A.shape[0])
    z = M.apply(r); zr = ref.apply(r)
    assert np.abs(zr).max() > 1e3 * np.abs(r).max()
    assert np.abs(z - zr).max() <= 1e-9 * np.abs(zr).max()
    M.close()
    for lv in dl:
        for h in lv.values():
            h.close()


def test_gmres_with_the_amg_preconditioner(gpu):
    """SolverType::GmresAmg: left-preconditioned GMRES(30) with the V-cycle; same iteration counts as the restatement, and a
    solution that satisfies the unpreconditioned system. Positive-definite regime (k below the first mode of the box) so that
    the multigrid cycle is a contraction, plus the damped k of config #4."""
    nx, ny, nz = 16, 16, 8
    for k in (0.3, complex(0.5, 0.05)):
        A = _helmholtz(nx, ny, nz, k)
        A = (A + 0.05 * sp.identity(A.shape[0])).tocsr()   # the pure Neumann Laplacian is singular at k = 0: shifted
        levels = box_hierarchy(A, nx, ny, nz, 3)
        n = A.shape[0]
        b = A @ _xvec(n)
        ref = O.AmgHierarchy(_oracle_levels(levels), smoother=0, jacobi_weight=0.8, num_pre_smooth=2, num_post_smooth=2)
        xr, ir = ref.gmres(b, restart=30, max_iterations=10, tol=1e-8)
        dl = _device_levels(levels)
        M = ma.AmgPreconditioner(dl, smoother="jacobi", jacobi_weight=0.8, num_pre_smooth=2, num_post_smooth=2)
        op = ma.LinearOperator.csr(dl[0]["A"])
        x, info = ma.gmres_preconditioned(op, M, b, restart=30, max_iterations=10, tol=1e-8)
        assert info.converged == ir.converged
        assert abs(info.iterations - ir.iterations) <= 1 and info.restarts == ir.restarts, (info.iterations, ir.iterations)
        if ir.converged:
            assert np.linalg.norm(A @ x - b) <= 1e-5 * np.linalg.norm(b)
            assert np.linalg.norm(x - xr) <= 1e-6 * np.linalg.norm(xr)
        op.close(); M.close()
        for lv in dl:
            for h in lv.values():
                h.close()
