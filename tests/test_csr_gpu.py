"""GPU parity of the sparse FEM side: CSR SpMV, residual, Jacobi / l1-Jacobi sweeps vs the CPU oracle.

Known answers from math-solvers/src/sparse/csr.rs:659-736 and math-fem/src/assembly/helmholtz.rs:354-390;
P1-tet Helmholtz matrices of the F1M family at a test size. Tolerance: 1e-13 relative
to the row scale (same products, different summation order inside a row)."""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import fem

pytestmark = pytest.mark.gpu


def _x0(n):
    i = np.arange(n)
    return np.sin(0.1 * i) + 1j * np.cos(0.2 * i)       # the deterministic pattern of tests/test_fmm_validation.rs:121


def test_csr_known_answers(gpu):
    A = ma.CsrOperator([0, 2, 4], [0, 1, 0, 1], values=[1, 2, 3, 4])
    assert np.allclose(A.matvec([1, 2]), [5, 11])
    B = ma.CsrOperator([0, 2, 3, 5], [0, 2, 1, 0, 2], values=[1, 2, 3, 4, 5])
    assert np.allclose(B.matvec([1, 1, 1]), [3, 3, 9])
    A.close(); B.close()


@pytest.mark.parametrize("layout", ["sell", "csr"])
@pytest.mark.parametrize("nxyz", [(6, 5, 4), (21, 17, 13)])
def test_helmholtz_spmv_and_smoothers_match_oracle(gpu, nxyz, layout, monkeypatch):
    """Both device layouts of the same operator: sliced ELLPACK (the default for near-uniform rows) and CSR-vector."""
    monkeypatch.setenv("MA_CSR_SELL", "1" if layout == "sell" else "0")
    nodes, rp, ci, K, M = fem.helmholtz_box(*nxyz)
    n = len(rp) - 1
    assert n == (nxyz[0] + 1) * (nxyz[1] + 1) * (nxyz[2] + 1)
    op = ma.CsrOperator(rp, ci, K=K, M=M)
    x = _x0(n); b = np.cos(0.3 * np.arange(n)) + 0.5j
    for k in (0.0, 2 * np.pi * 100.0 / 343.0, 1.832 + 0.01j):
        op.set_wavenumber(k)
        vals = O.helmholtz_values(K, M, k)
        if k == 0.0:
            assert np.array_equal(vals, K.astype(complex))            # helmholtz.rs:354-390
        y_ref = O.csr_matvec(rp, ci, vals, x, nthreads=4)
        scale = np.abs(y_ref).max()
        assert np.abs(op.matvec(x) - y_ref).max() <= 1e-13 * scale
        assert np.abs(op.residual(x, b) - (b - y_ref)).max() <= 1e-13 * max(scale, 1.0)
        xj_ref = O.amg_jacobi(rp, ci, vals, x, b, 0.8, 2, nthreads=4)   # AmgConfig::for_parallel: omega 0.8, 2 sweeps
        assert np.abs(op.jacobi(x, b, 0.8, 2) - xj_ref).max() <= 1e-12 * np.abs(xj_ref).max()
        xl_ref = O.amg_l1_jacobi(rp, ci, vals, x, b, 2, nthreads=4)
        assert np.abs(op.l1_jacobi(x, b, 2) - xl_ref).max() <= 1e-12 * np.abs(xl_ref).max()
    op.close()


def test_generic_complex_csr_with_ragged_rows(gpu):
    """Empty rows, a dense row, long and short rows: every group width path."""
    rng = np.random.default_rng(4)
    n = 500
    rows = []
    for i in range(n):
        m = [0, 1, 3, 17, 70, 300][i % 6]
        rows.append(np.sort(rng.choice(n, size=m, replace=False)))
    rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    ci = np.concatenate(rows).astype(np.int64)
    vals = rng.standard_normal(len(ci)) + 1j * rng.standard_normal(len(ci))
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    op = ma.CsrOperator(rp, ci, values=vals)
    y_ref = O.csr_matvec(rp, ci, vals, x)
    assert np.abs(op.matvec(x) - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
    op.close()


def test_jacobi_sweeps_on_device_buffers(gpu):
    """Device-resident ping-pong form: equals the oracle sweep for sweep and, as in smoother.rs:192-237 /
    the amg.rs tests, drives the residual down (K + 0.25 M is positive definite)."""
    import torch
    nodes, rp, ci, K, M = fem.helmholtz_box(12, 10, 8)
    n = len(rp) - 1
    op = ma.CsrOperator(rp, ci, K=K, M=M)
    op.set_wavenumber(0.5j)
    vals = O.helmholtz_values(K, M, 0.5j)
    dev = torch.device("cuda", 0)
    bh = np.ones(n, dtype=complex)
    b = torch.tensor(bh, device=dev); x = torch.zeros_like(b); tmp = torch.empty_like(b); r = torch.empty_like(b)
    st = torch.cuda.current_stream().cuda_stream
    op.residual_dev(x.data_ptr(), b.data_ptr(), r.data_ptr(), st); r0 = float(torch.linalg.norm(r))
    op.jacobi_dev(x.data_ptr(), b.data_ptr(), 2.0 / 3.0, 5, tmp.data_ptr(), st)           # odd count: result copied back
    x_ref = O.amg_jacobi(rp, ci, vals, np.zeros(n, dtype=complex), bh, 2.0 / 3.0, 5)
    assert np.abs(x.cpu().numpy() - x_ref).max() <= 1e-12 * np.abs(x_ref).max()
    op.l1_jacobi_dev(x.data_ptr(), b.data_ptr(), 4, tmp.data_ptr(), st)
    x_ref = O.amg_l1_jacobi(rp, ci, vals, x_ref, bh, 4)
    assert np.abs(x.cpu().numpy() - x_ref).max() <= 1e-12 * np.abs(x_ref).max()
    op.l1_jacobi_dev(x.data_ptr(), b.data_ptr(), 200, tmp.data_ptr(), st)
    op.residual_dev(x.data_ptr(), b.data_ptr(), r.data_ptr(), st); r1 = float(torch.linalg.norm(r))
    assert r1 < 0.7 * r0                                   # Jacobi is a smoother, not a solver
    op.close()


def test_csr_argument_errors(gpu):
    with pytest.raises(ma.MaError) as e:
        ma.CsrOperator([0, 1, 2], [0, 5], values=[1, 1])          # column out of range
    assert e.value.status == ma.MA_ERR_INVALID
    A = ma.CsrOperator([0, 1, 2], [0, 1], values=[1, 1])
    with pytest.raises(ma.MaError):
        A.set_wavenumber(1.0)                                      # complex-valued handle has no K/M
    A.close()


def test_fem_coo_smoother_matches_oracle(gpu):
    """math-fem's HelmholtzMatrix is COO with unsummed duplicates (helmholtz.rs:22-33); Jacobi sweep and residual of
    multigrid/smoother.rs:120-176 vs the CPU restatement, including a zero-diagonal row (left untouched, :143-146)."""
    rng = np.random.default_rng(12)
    n = 400
    rows, cols, vals = [], [], []
    for i in range(n):
        if i != 7:                                        # row 7 has no diagonal entry at all
            for part in (2.0 + 0.3j, 1.5 - 0.1j):         # the diagonal arrives as two triplets
                rows.append(i); cols.append(i); vals.append(part)
        for j in rng.choice(n, size=5, replace=False):
            if j != i:
                for _ in range(2):                        # duplicated off-diagonals
                    rows.append(i); cols.append(int(j)); vals.append(0.1 * (rng.standard_normal() + 1j * rng.standard_normal()))
    perm = rng.permutation(len(rows))                     # triplet order is arbitrary
    rows = np.array(rows)[perm]; cols = np.array(cols)[perm]; vals = np.array(vals)[perm]
    x = _x0(n); b = np.cos(0.3 * np.arange(n)) + 0.5j
    A = ma.CsrOperator.from_coo(n, rows, cols, vals)
    r_ref = O.fem_residual(n, rows, cols, vals, x, b)
    assert np.abs(A.fem_residual(x, b) - r_ref).max() <= 1e-13 * np.abs(r_ref).max()
    for omega, its in ((2.0 / 3.0, 2), (0.8, 3)):
        x_ref = O.fem_smooth(n, rows, cols, vals, x, b, kind=1, iterations=its, omega=omega)
        x_dev = A.fem_smooth(x, b, kind=1, iterations=its, omega=omega)
        assert np.abs(x_dev - x_ref).max() <= 1e-12 * np.abs(x_ref).max()
        assert x_dev[7] == x[7]                           # zero diagonal: skipped
    # Gauss-Seidel (the reference's default smoother, smoother.rs:31-39, 71-117) and its symmetric form: level-scheduled
    # sweeps on the device equal the sequential sweep of the restatement (duplicates are pre-summed: rounding only)
    for kind, its in ((0, 2), (0, 5), (2, 2)):
        x_ref = O.fem_smooth(n, rows, cols, vals, x, b, kind=kind, iterations=its)
        x_dev = A.fem_smooth(x, b, kind=kind, iterations=its)
        assert np.abs(x_dev - x_ref).max() <= 1e-12 * np.abs(x_ref).max()
        assert x_dev[7] == x[7]                           # zero diagonal: skipped (smoother.rs:100-102)
    with pytest.raises(ma.MaError) as ei:
        A.fem_smooth(x, b, kind=3)
    assert ei.value.status == ma.MA_ERR_INVALID
    A.close()


def test_sym_gauss_seidel_matches_oracle(gpu):
    """smooth_sym_gauss_seidel (amg.rs:932-978): forward + backward sweeps by dependency levels == the sequential sweeps.
    Cases: FEM operator in fused K - k^2 M form and with stored complex values; an unsymmetric pattern (a_ij stored without
    a_ji: the levels must still order the two rows) with a row that stores no diagonal (uses 1) and a zero diagonal (skipped)."""
    _, rp, col, K, M = fem.helmholtz_box(9, 8, 7)
    n = len(rp) - 1
    k = 2.0 + 0.05j
    vals = O.helmholtz_values(K, M, k)
    x = _x0(n); b = np.sin(0.2 * np.arange(n)) + 1j * np.cos(0.1 * np.arange(n))
    for mode in ("km", "values"):
        A = ma.CsrOperator(rp, col, K=K, M=M) if mode == "km" else ma.CsrOperator(rp, col, values=vals)
        if mode == "km":
            A.set_wavenumber(k)
        for sweeps in (1, 3):
            x_ref = O.amg_sym_gauss_seidel(rp, col, vals, x, b, sweeps)
            x_dev = A.sym_gauss_seidel(x, b, sweeps)
            assert np.abs(x_dev - x_ref).max() <= 1e-12 * np.abs(x_ref).max(), (mode, sweeps)
        nf, nb = A.gauss_seidel_levels()
        assert 1 < nf < n and 1 < nb < n
        A.close()
    rng = np.random.default_rng(5)
    n = 300
    rows = []
    for i in range(n):
        c = set(int(j) for j in rng.choice(n, size=6, replace=False)) - {i}
        if i == 11 or i != 20:
            c.add(i)                    # row 20 stores no diagonal at all (amg.rs:944 uses 1)
        rows.append(sorted(c))
    rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    col = np.concatenate(rows).astype(np.int64)
    vals = 0.2 * (rng.standard_normal(len(col)) + 1j * rng.standard_normal(len(col)))
    for i in range(n):
        for idx in range(rp[i], rp[i + 1]):
            if col[idx] == i:
                vals[idx] = 0.0 if i == 11 else 4.0 + 0.5j   # row 11: stored zero diagonal, left alone (amg.rs:952)
    x = _x0(n); b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    A = ma.CsrOperator(rp, col, values=vals)
    x_ref = O.amg_sym_gauss_seidel(rp, col, vals, x, b, 2)
    x_dev = A.sym_gauss_seidel(x, b, 2)
    assert np.abs(x_dev - x_ref).max() <= 1e-12 * np.abs(x_ref).max()
    assert x_dev[11] == x[11]
    # as a preconditioner: z = M^-1 r from z = 0
    Mp = ma.Preconditioner(A, kind="sgs", sweeps=2)
    z_ref = O.amg_sym_gauss_seidel(rp, col, vals, np.zeros(n, dtype=complex), b, 2)
    assert np.abs(Mp.apply(b) - z_ref).max() <= 1e-12 * np.abs(z_ref).max()
    Mp.close()
    A.close()


def test_gauss_seidel_level_counts(gpu):
    """A tridiagonal operator is one long dependency chain (n levels); a diagonal one has a single level."""
    n = 50
    rp = [0]; col = []
    for i in range(n):
        col += [j for j in (i - 1, i, i + 1) if 0 <= j < n]; rp.append(len(col))
    A = ma.CsrOperator(rp, col, values=np.ones(len(col)))
    assert A.gauss_seidel_levels() == (n, n)
    A.close()
    D = ma.CsrOperator(list(range(n + 1)), list(range(n)), values=np.full(n, 2.0))
    assert D.gauss_seidel_levels() == (1, 1)
    x = D.sym_gauss_seidel(np.zeros(n), np.arange(n) + 1j, 1)
    assert np.abs(x - (np.arange(n) + 1j) / 2.0).max() < 1e-15
    D.close()


@pytest.mark.parametrize("layout", ["sell", "csr"])
def test_helmholtz_assemble_with_boundary_terms(gpu, layout, monkeypatch):
    """HelmholtzAssembler::assemble (assembler.rs:216-257): A = K - k^2 M + sum_t c_t B_t with two boundary matrices (an
    impedance wall and a second tag that gets no coefficient), SpMV, smoothers and transpose on the assembled operator."""
    monkeypatch.setenv("MA_CSR_SELL", "1" if layout == "sell" else "0")
    nodes, rp, ci, K, M = fem.helmholtz_box(7, 6, 5)
    n = len(rp) - 1
    rng = np.random.default_rng(6)
    wall = nodes[:, 0] < 1e-12                                       # entries coupling two nodes of the x = 0 wall
    rows = np.repeat(np.arange(n), np.diff(rp))
    B1 = np.where(wall[rows] & wall[ci], 0.01 * (1.0 + rng.random(len(ci))), 0.0)
    B2 = np.where(nodes[rows, 2] > 2.4, 0.02, 0.0) * (nodes[ci, 2] > 2.4)
    op = ma.CsrOperator(rp, ci, K=K, M=M)
    op.add_boundary(3, B1); op.add_boundary(9, B2)
    k = 1.9 + 0.05j
    x = _x0(n); b = np.cos(0.3 * np.arange(n)) + 0.5j
    for coeffs in ({3: 1j * k * 0.7}, {3: 1j * k * 0.7, 9: -0.3 + 0.1j}, {}, {5: 2.0}):
        op.assemble(k, coeffs)
        vals = O.helmholtz_values(K, M, k)
        if 3 in coeffs: vals = vals + coeffs[3] * B1
        if 9 in coeffs: vals = vals + coeffs[9] * B2
        y_ref = O.csr_matvec(rp, ci, vals, x, nthreads=4)
        assert np.abs(op.matvec(x) - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
        xj = O.amg_jacobi(rp, ci, vals, x, b, 0.8, 2, nthreads=4)
        assert np.abs(op.jacobi(x, b, 0.8, 2) - xj).max() <= 1e-12 * np.abs(xj).max()
        xl = O.amg_l1_jacobi(rp, ci, vals, x, b, 2, nthreads=4)
        assert np.abs(op.l1_jacobi(x, b, 2) - xl).max() <= 1e-12 * np.abs(xl).max()
    op.assemble(k, {3: 1j * k * 0.7})
    lo = ma.LinearOperator.csr(op)
    vals = O.helmholtz_values(K, M, k) + 1j * k * 0.7 * B1
    dense = np.zeros((n, n), dtype=complex); dense[rows, ci] = vals
    assert np.abs(lo.apply_transpose(x) - dense.T @ x).max() <= 1e-12 * np.abs(dense.T @ x).max()
    op.set_wavenumber(k)                                             # back to the fused K - k^2 M path
    y0 = O.csr_matvec(rp, ci, O.helmholtz_values(K, M, k), x, nthreads=4)
    assert np.abs(op.matvec(x) - y0).max() <= 1e-13 * np.abs(y0).max()
    assert np.abs(lo.apply_transpose(x) - y0).max() <= 1e-12 * np.abs(y0).max()     # the cached transpose follows (K, M symmetric)
    op.set_wavenumber(0.5 * k)
    y1 = O.csr_matvec(rp, ci, O.helmholtz_values(K, M, 0.5 * k), x, nthreads=4)
    assert np.abs(lo.apply_transpose(x) - y1).max() <= 1e-12 * np.abs(y1).max()
    lo.close(); op.close()


def test_full_size_fem_box(gpu):
    """BASELINE.json config #4 at full size (F1M: 100^3 nodes, 1.48e7 non-zeros): checksums that do not depend on the size
    (K annihilates constants, M's entries sum to the volume, so (K - k^2 M) 1 sums to -k^2 V; linearity) and every device
    smoother against the CPU restatement, whose sequential sweeps finish in well under a second per sweep."""
    _, rp, ci, K, M = fem.helmholtz_box(99, 99, 99)
    n = len(rp) - 1
    assert n == 1000000
    op = ma.CsrOperator(rp, ci, K=K, M=M)
    ones = np.ones(n, dtype=complex)
    op.set_wavenumber(0.0)
    y0 = op.matvec(ones)
    assert np.abs(y0).max() <= 1e-12 * np.abs(K).max()                 # stiffness rows sum to zero
    k = 2 * np.pi * 100.0 / 343.0 + 0.01j
    op.set_wavenumber(k)
    y1 = op.matvec(ones)
    vol = 5.0 * 4.0 * 2.5
    assert abs(y1.sum() - (-(k * k) * vol)) <= 1e-10 * abs(k * k * vol)  # sum of the mass matrix = volume of the box
    x = _x0(n); z = np.cos(0.3 * np.arange(n)) + 0.25j
    a, b_ = 0.7 - 0.2j, -1.3 + 0.4j
    lin = op.matvec(a * x + b_ * z) - (a * op.matvec(x) + b_ * op.matvec(z))
    assert np.abs(lin).max() <= 1e-12 * np.abs(op.matvec(x)).max()
    vals = O.helmholtz_values(K, M, k)
    b = np.sin(0.2 * np.arange(n)) + 1j * np.cos(0.1 * np.arange(n))
    y_ref = O.csr_matvec(rp, ci, vals, x, nthreads=8)
    assert np.abs(op.matvec(x) - y_ref).max() <= 1e-13 * np.abs(y_ref).max()
    xj = O.amg_jacobi(rp, ci, vals, x, b, 0.8, 2, nthreads=8)
    assert np.abs(op.jacobi(x, b, 0.8, 2) - xj).max() <= 1e-12 * np.abs(xj).max()
    xl = O.amg_l1_jacobi(rp, ci, vals, x, b, 2, nthreads=8)
    assert np.abs(op.l1_jacobi(x, b, 2) - xl).max() <= 1e-12 * np.abs(xl).max()
    xs = O.amg_sym_gauss_seidel(rp, ci, vals, x, b, 1)
    xd = op.sym_gauss_seidel(x, b, 1)
    assert np.abs(xd - xs).max() <= 1e-11 * np.abs(xs).max()
    # the sweep is sequential: the row updated last (row 0 of the backward sweep) satisfies its equation exactly
    r = op.residual(xd, b)
    assert abs(r[0]) <= 1e-12 * abs(b[0])
    assert op.gauss_seidel_levels() == (298, 298)
    op.close()


def test_persistent_gauss_seidel_sweep_is_the_level_launches(gpu, monkeypatch):
    """The three schedules of a sweep -- one persistent launch in which the new value is its own flag (the default), one persistent
    launch with a device-wide barrier per level, a launch per level -- handle the same rows with the same 16-lane reduction, so the
    iterates must agree bit for bit: forward, backward, both modes, the fused K - k^2 M operator and stored complex values, on a box
    with 46 levels and on an unsymmetric random pattern."""
    def _xvec(m):
        i = np.arange(m)
        return np.sin(0.1 * i) + 1j * np.cos(0.2 * i)
    nodes, rp, ci, K, M = fem.helmholtz_box(12, 10, 8)
    n = len(rp) - 1
    b = _xvec(n); x0 = 0.3 * _xvec(n)[::-1].copy()
    k = 1.3 + 0.2j
    outs = {}
    for pers in ("1", "0", "flags"):
        monkeypatch.setenv("MA_CSR_GS_PERSISTENT", "1" if pers == "1" else "0")
        monkeypatch.setenv("MA_CSR_GS_FLAGS", "1" if pers == "flags" else "0")
        h = ma.CsrOperator(rp, ci, K=K, M=M); h.set_wavenumber(k)
        fwd, bwd = h.gauss_seidel_levels()
        assert fwd >= 8 and bwd >= 8
        x1 = h.sym_gauss_seidel(x0, b, 2)
        hc = ma.CsrOperator.from_coo(n, np.repeat(np.arange(n), np.diff(rp)), ci, O.helmholtz_values(K, M, k))
        x2 = hc.fem_smooth(x0, b, kind=0, iterations=3)
        x3 = hc.fem_smooth(x0, b, kind=2, iterations=1)
        outs[pers] = (x1, x2, x3)
        h.close(); hc.close()
    for a, c, f in zip(outs["1"], outs["0"], outs["flags"]):
        assert np.array_equal(a, c) and np.array_equal(f, c)
    vals = O.helmholtz_values(K, M, k)
    ref = O.amg_sym_gauss_seidel(rp, ci, vals, x0, b, 2)
    assert np.abs(outs["1"][0] - ref).max() <= 1e-12 * np.abs(ref).max()
    rng = np.random.default_rng(11)
    import scipy.sparse as sp
    R = sp.random(700, 700, density=0.01, random_state=5, format="csr").astype(np.complex128)
    R.data = rng.standard_normal(R.nnz) + 1j * rng.standard_normal(R.nnz)
    R = (R + sp.diags(8.0 + rng.standard_normal(700))).tocsr(); R.sort_indices()
    bb = _xvec(700); res = {}
    for pers in ("1", "0", "flags"):
        monkeypatch.setenv("MA_CSR_GS_PERSISTENT", "1" if pers == "1" else "0")
        monkeypatch.setenv("MA_CSR_GS_FLAGS", "1" if pers == "flags" else "0")
        h = ma.CsrOperator(R.indptr, R.indices, values=R.data)
        res[pers] = h.sym_gauss_seidel(np.zeros(700, dtype=complex), bb, 3)
        h.close()
    assert np.array_equal(res["1"], res["0"]) and np.array_equal(res["flags"], res["0"])
    refu = O.amg_sym_gauss_seidel(R.indptr, R.indices, R.data, np.zeros(700, dtype=complex), bb, 3)
    assert np.abs(res["1"] - refu).max() <= 1e-12 * np.abs(refu).max()


def test_sweep_survives_the_sentinel_pattern_in_its_input(gpu):
    """The default sweep marks "not there yet" with a NaN payload (0x7FFC0DE0DEADBEEF). Inputs that carry that very pattern -- or any
    NaN -- must come out as NaNs where they propagate, not leave a row waiting: the writer never stores the sentinel itself."""
    nodes, rp, ci, K, M = fem.helmholtz_box(10, 9, 8)
    n = len(rp) - 1
    h = ma.CsrOperator(rp, ci, K=K, M=M); h.set_wavenumber(1.1 + 0.1j)
    assert min(h.gauss_seidel_levels()) >= 8
    sentinel = np.frombuffer(np.array([0x7FFC0DE0DEADBEEF], dtype=np.uint64).tobytes(), dtype=np.float64)[0]
    i = np.arange(n)
    b = (np.sin(0.2 * i) + 1j * np.cos(0.1 * i)).astype(np.complex128)
    x0 = (np.sin(0.1 * i) + 1j * np.cos(0.2 * i)).astype(np.complex128)
    x0[5] = complex(sentinel, 1.0); x0[n // 2] = complex(2.0, sentinel); b[n - 3] = complex(sentinel, sentinel); b[7] = complex(np.nan, 0.0)
    x = h.sym_gauss_seidel(x0, b, 2)
    assert x.shape == (n,) and np.isnan(x).any()
    ma.check(ma.lib().ma_csr_status(h.h))                    # no wait was abandoned
    clean = h.sym_gauss_seidel(np.nan_to_num(x0, nan=0.5), np.nan_to_num(b, nan=0.25), 1)
    assert np.isfinite(clean).all()
    h.close()
