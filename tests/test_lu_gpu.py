"""GPU parity of the dense complex solve (ma_zgesv / LU plan) against LAPACK semantics.

Mirrors math-solvers/src/direct/lu.rs:163-240 (residual <= 1e-10 on small real/complex systems,
identity, singular => Err) and adds BEM-sized systems checked against the CPU oracle's zgesv and
NumPy's LAPACK.
"""
import numpy as np
import pytest
import oracle_lib as O
import math_audio_amd as ma

pytestmark = pytest.mark.gpu


def _rand(n, seed, cond_shift=0.0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A += cond_shift * np.eye(n)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    return A, b


@pytest.mark.parametrize("M,N,K", [(128, 128, 8), (128, 128, 128), (200, 333, 96), (37, 515, 19), (1000, 1, 128)])
def test_zgemm_sub_kernel(gpu, M, N, K):
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((M, K)) + 1j * rng.standard_normal((M, K))
    B = rng.standard_normal((K, N)) + 1j * rng.standard_normal((K, N))
    Cm = rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))
    got = ma.test_zgemm_sub(A, B, Cm)
    ref = Cm - A @ B
    assert np.abs(got - ref).max() <= 1e-12 * K * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("M,N,K", [(64, 128, 8), (65, 129, 16), (1, 1, 8), (300, 70, 384), (129, 1000, 24), (1000, 1000, 64), (2100, 4100, 16), (40, 66000, 8)])
def test_zgemm_dma_kernels_are_bitwise_the_register_staged_one(gpu, M, N, K):
    """K a multiple of 8: the update runs in zgemm3m_dma_kernel (operands global -> LDS by LDS-DMA, 32 x 64 of C per wavefront);
    MA_ZGEMM_DMA=0 is the register-staged zgemm3m_sub_kernel (the kernel for ragged K), =2 the 128 x 128-tile form. Every entry of C
    accumulates the same products in the same order in all three: equal bits, ragged edges included (rows and columns beyond the
    matrix are fetched from the last valid one and never stored)."""
    rng = np.random.default_rng(M * 3 + N * 5 + K)
    A = rng.standard_normal((M, K)) + 1j * rng.standard_normal((M, K))
    B = rng.standard_normal((K, N)) + 1j * rng.standard_normal((K, N))
    Cm = rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))
    ref = Cm - A @ B
    got = {}
    for mode in (0, 1, 2):
        with _with_env(MA_ZGEMM_DMA=mode):
            got[mode] = ma.test_zgemm_sub(A, B, Cm)
        assert np.abs(got[mode] - ref).max() <= 1e-12 * K * max(1.0, np.abs(ref).max())
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
    # from 512 tiles on the tiles are dealt out XCD by XCD in blocks of 4 x 4 (a one-dimensional grid): every tile once, the same bits
    with _with_env(MA_ZGEMM_DMA=1, MA_ZGEMM_TILE_ORDER=0):
        assert np.array_equal(ma.test_zgemm_sub(A, B, Cm), got[0])


@pytest.mark.parametrize("M,N,K,persist", [(700, 1100, 64, 0), (1300, 900, 256, 0), (515, 2100, 40, 1), (64, 64, 8, 0)])
def test_zgemm_sub_kernel_drawing_its_tiles(gpu, M, N, K, persist):
    """Large updates draw their tiles XCD by XCD (8 x 8 blocks of tiles per XCD, counters per launch, stealing when an XCD runs
    dry; lu_kernels.hip): every tile exactly once whatever the placement, ragged edges, blocks that are partly outside the
    matrix, and the counters are left at zero for the launch that reuses them (the same shape twice)."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)) + 1j * rng.standard_normal((M, K))
    B = rng.standard_normal((K, N)) + 1j * rng.standard_normal((K, N))
    Cm = rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))
    ref = Cm - A @ B
    with _with_env(MA_ZGEMM_XCD_TILES=1, MA_ZGEMM_XCD_PERSIST=persist):
        for _ in range(2):
            got = ma.test_zgemm_sub(A, B, Cm)
            assert np.abs(got - ref).max() <= 1e-12 * K * max(1.0, np.abs(ref).max())
    assert np.array_equal(ma.test_zgemm_sub(A, B, Cm), got)      # the plain tile order gives the same bits: tiles are independent


def test_lu_solve_real_2x2(gpu):            # lu.rs:163-175
    A = np.array([[4.0, 1.0], [1.0, 3.0]]); b = np.array([1.0, 2.0])
    x = ma.zgesv(A, b)
    assert np.abs(A @ x - b).max() <= 1e-10


def test_lu_solve_complex_2x2(gpu):         # lu.rs:177-193
    A = np.array([[4 + 1j, 1 + 0j], [1 + 0j, 3 - 1j]]); b = np.array([1 + 1j, 2 - 1j])
    x = ma.zgesv(A, b)
    assert np.abs(A @ x - b).max() <= 1e-10


def test_lu_identity(gpu):                  # lu.rs:195-206
    n = 5
    x = ma.zgesv(np.eye(n), np.arange(1, n + 1, dtype=float))
    assert np.abs(x - np.arange(1, n + 1)).max() <= 1e-10


def test_lu_singular_is_an_error(gpu):      # lu.rs:208-216
    with pytest.raises(ma.MaError) as e:
        ma.zgesv(np.array([[1.0, 2.0], [2.0, 4.0]]), np.array([1.0, 2.0]))
    assert e.value.status == ma.MA_ERR_SINGULAR


def test_lu_dimension_mismatch(gpu):
    with pytest.raises(ma.MaError) as e:
        ma.zgesv(np.eye(3), np.ones(2))
    assert e.value.status == ma.MA_ERR_DIM


@pytest.mark.parametrize("n", [1, 3, 17, 64, 127, 128, 129, 300, 777, 1280])
def test_lu_random_matches_lapack(gpu, n):
    A, b = _rand(n, n)
    x = ma.zgesv(A, b)
    xr = np.linalg.solve(A, b)
    xo, _, rc = O.zgesv(A, b, nthreads=8)
    assert rc == 0
    res = np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x))
    assert res <= 1e-14 * n
    # forward error relative to LAPACK within a few condition-number-scaled ulps
    kappa = np.linalg.cond(A) if n <= 400 else 1e4
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) <= 1e-13 * kappa
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-13 * kappa


@pytest.mark.parametrize("n", [33, 65, 66, 70, 97, 129, 130, 193, 777, 1500])
def test_pair_panels_match_lapack(gpu, n):
    """The schedule of the 4 096-16 384-row plans at small sizes (MA_LU_REG_PANEL=2, MA_LU_CU_SPLIT=64): a 64-column panel as two
    register half-panels, lu_lane_step_kernel between them, lu_lane_step2_kernel after them, ragged last panels and strips of 1, 2,
    4, 6 columns (where the step kernel once read L10 from rows another workgroup was permuting). LAPACK's pivots, LAPACK's solution."""
    import scipy.linalg as sla
    A, b = _rand(n, 4000 + n)
    with _with_env(MA_LU_REG_PANEL=2, MA_LU_CU_SPLIT=64):
        x, piv = ma.zgesv(A, b, return_pivots=True)
    _, piv_ref = sla.lu_factor(A)
    assert np.array_equal(piv, piv_ref)
    xr = np.linalg.solve(A, b)
    res = np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x))
    assert res <= 1e-14 * n
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) <= 1e-13 * (np.linalg.cond(A) if n <= 400 else 1e4)


@pytest.mark.parametrize("n", [66, 130, 450, 777, 1500, 2100])
def test_block_step_matches_lapack(gpu, n):
    """MA_LU_BLOCK_STEP=1 (round 4; built, measured neutral, off by default): the main lane's per-panel launches of a block of six
    64-column panels -- 12 gathers / scatters, 6 triangular solves, 6 zgemv, 5 in-block updates -- as lu_block_row_moves_kernel (every
    panel's interchanges on a strip of columns, panel after panel) + lu_block_trsm_kernel (U12 of the whole block row left-looking on
    the matrix cores, the right-hand side riding along) + one zgemv: LAPACK's pivots and solution; ragged last panels and blocks."""
    import scipy.linalg as sla
    A, b = _rand(n, 5000 + n)
    with _with_env(MA_LU_REG_PANEL=2, MA_LU_CU_SPLIT=64, MA_LU_BLOCK_STEP=1):
        x, piv = ma.zgesv(A, b, return_pivots=True)
        F = ma.LuFactorization(A)                            # the stored factors serve later right-hand sides (the row order of L is LAPACK's)
        x2 = F.solve(2.0 * b)
        F.close()
    _, piv_ref = sla.lu_factor(A)
    assert np.array_equal(piv, piv_ref)
    xr = np.linalg.solve(A, b)
    res = np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x))
    assert res <= 1e-14 * n
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) <= 1e-13 * (np.linalg.cond(A) if n <= 400 else 1e4)
    assert np.linalg.norm(x2 - 2.0 * xr) / np.linalg.norm(xr) <= 1e-12 * (np.linalg.cond(A) if n <= 400 else 1e4)


def test_split_plan_refuses_the_null_stream_for_its_staged_schedule(gpu):
    """a plan that splits the chip runs its big updates on a CU-masked stream, which the runtime makes as a BLOCKING stream;
    a driver on the NULL stream would serialise against every update and silently lose the lanes' overlap: stage_reset / stage_begin
    say so (MA_ERR_INVALID) instead."""
    import torch
    with _with_env(MA_LU_REG_PANEL=2, MA_LU_CU_SPLIT=64):
        lu = ma.LuPlan(900)
    assert lu.main_stream()
    with pytest.raises(ma.MaError) as e:
        lu.stage_reset(0)
    assert e.value.status == ma.MA_ERR_INVALID
    A = torch.zeros(900 * 900, dtype=torch.complex128, device="cuda"); b = torch.zeros(900, dtype=torch.complex128, device="cuda")
    with pytest.raises(ma.MaError) as e:
        lu.stage_begin(0, A.data_ptr(), b.data_ptr(), 1, 0)
    assert e.value.status == ma.MA_ERR_INVALID
    lu.stage_reset(lu.main_stream())                         # the plan's own stream is fine
    lu.close()


def test_default_schedule_of_a_4200_row_plan(gpu):
    """4 096-16 384 rows: the plan splits the chip by default (ma_lu_plan_main_stream is the masked update stream) and factors with
    the register pair panels and the LDS-DMA update kernel; pivots and solution against LAPACK at a size inside that range."""
    import scipy.linalg as sla
    n = 4200
    A, b = _rand(n, 4200)
    lu = ma.LuPlan(n)
    assert lu.main_stream()
    assert lu.stage_spacing(3) == (lu.num_blocks() + 1) // 3
    lu.close()
    x, piv = ma.zgesv(A, b, return_pivots=True)
    _, piv_ref = sla.lu_factor(A)
    assert np.array_equal(piv, piv_ref)
    res = np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x))
    assert res <= 1e-14 * n
    small = ma.LuPlan(900)
    assert not small.main_stream()                          # outside the range: the round-2 schedule on the whole chip
    small.close()


@pytest.mark.parametrize("n", [64, 300, 777])
def test_lu_pivots_are_lapacks(gpu, n):
    """Partial pivoting picks LAPACK's rows (izamax on |re| + |im|, first maximum): on generic data, where no two candidates
    of a column agree to 1e-6 (the device compares the top 32 bits of the magnitude), the interchanges are identical."""
    import scipy.linalg as sla
    A, b = _rand(n, 1000 + n)
    x, piv = ma.zgesv(A, b, return_pivots=True)
    _, piv_ref = sla.lu_factor(A)
    assert np.array_equal(piv, piv_ref)


def test_lu_needs_pivoting(gpu):
    """Zero leading diagonal forces interchanges across workgroups and across panels."""
    n = 400
    rng = np.random.default_rng(5)
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    A[np.arange(n), np.arange(n)] = 0.0
    A = np.roll(A, 7, axis=0)
    b = rng.standard_normal(n) + 0j
    x = ma.zgesv(A, b)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) <= 1e-10


def test_lu_on_bem_system_matches_oracle(gpu):
    """Config #2: S1 sphere system, x vs the CPU restatement <= 1e-8 relative L2."""
    from helpers import k_from_ka, RADIUS
    om = O.icosphere(RADIUS, 3)
    k = k_from_ka(1.0)
    beta, _ = O.beta_adaptive(k, RADIUS)
    A, rhs0 = O.build_tbem_system_with_beta(om, k, beta, nthreads=8)
    rhs = rhs0 + O.compute_rhs_with_beta(om.center, om.normal, k, beta)
    xo, _, rc = O.zgesv(A, rhs, nthreads=8)
    assert rc == 0
    x = ma.zgesv(A, rhs)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8


@pytest.mark.parametrize("batch_panel", [1, 0])
def test_batched_factor_solve_is_bitwise_the_single_one(gpu, batch_panel):
    """Frequencies in flight: factoring independent systems together must not change any of them. batch_panel = 1: one panel
    kernel walks the systems round-robin inside every column (lu_panel_batch_kernel, 32-column panels) -- each system's
    arithmetic is the single-system kernel's at that panel width, whatever the rows per workgroup; batch_panel = 0: a panel
    kernel per system, interleaved."""
    import torch
    n = 900
    dev = torch.device("cuda", 0)
    mats = [_rand(n, 100 + i) for i in range(3)]
    with _with_env(MA_LU_NB=32 if batch_panel else 64, MA_LU_BATCH_PANEL=batch_panel):
        lu = ma.LuPlan(n)                                  # the switches are read once per plan
    st = torch.cuda.current_stream().cuda_stream
    singles = []
    for A, b in mats:
        dA = torch.tensor(A, device=dev).reshape(-1); db = torch.tensor(b, device=dev)
        lu.factor_solve_dev(dA.data_ptr(), db.data_ptr(), 1, st)
        assert lu.status(st) == ma.MA_OK
        singles.append((dA.cpu().numpy().copy(), db.cpu().numpy().copy()))
    dAs = [torch.tensor(A, device=dev).reshape(-1) for A, _ in mats]; dbs = [torch.tensor(b, device=dev) for _, b in mats]
    lu.factor_solve_batch_dev([t.data_ptr() for t in dAs], [t.data_ptr() for t in dbs], 1, st)
    assert lu.status(st) == ma.MA_OK
    for (A, b), (fa, fb), dA, db in zip(mats, singles, dAs, dbs):
        assert np.array_equal(dA.cpu().numpy(), fa) and np.array_equal(db.cpu().numpy(), fb)
        assert np.linalg.norm(A @ fb - b) / np.linalg.norm(b) < 1e-11
    # a singular member of a batch is reported
    S = np.ones((n, n), dtype=complex)
    dS = torch.tensor(S, device=dev).reshape(-1); ds = torch.ones(n, dtype=torch.complex128, device=dev)
    dA0 = torch.tensor(mats[0][0], device=dev).reshape(-1); db0 = torch.tensor(mats[0][1], device=dev)
    lu.factor_solve_batch_dev([dA0.data_ptr(), dS.data_ptr()], [db0.data_ptr(), ds.data_ptr()], 1, st)
    assert lu.status(st) == ma.MA_ERR_SINGULAR
    lu.close()


@pytest.mark.parametrize("n,nrhs,nsys", [(333, 3, 1), (1100, 4, 2), (520, 2, 4)])
def test_several_right_hand_sides_and_systems(gpu, n, nrhs, nsys):
    """nrhs right-hand sides per system (d_B is [nrhs][n]) ride through interchanges, forward and backward substitution;
    up to 4 systems per batch."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n)
    lu = ma.LuPlan(n)
    st = torch.cuda.current_stream().cuda_stream
    As = [rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) for _ in range(nsys)]
    Bs = [rng.standard_normal((nrhs, n)) + 1j * rng.standard_normal((nrhs, n)) for _ in range(nsys)]
    dAs = [torch.tensor(A, device=dev).reshape(-1) for A in As]; dBs = [torch.tensor(B, device=dev).reshape(-1) for B in Bs]
    if nsys == 1:
        lu.factor_solve_dev(dAs[0].data_ptr(), dBs[0].data_ptr(), nrhs, st)
    else:
        lu.factor_solve_batch_dev([t.data_ptr() for t in dAs], [t.data_ptr() for t in dBs], nrhs, st)
    assert lu.status(st) == ma.MA_OK
    for A, B, dB in zip(As, Bs, dBs):
        X = dB.cpu().numpy().reshape(nrhs, n)
        for r in range(nrhs):
            assert np.linalg.norm(A @ X[r] - B[r]) / (np.linalg.norm(A) * np.linalg.norm(X[r])) <= 1e-14 * n
            assert np.linalg.norm(X[r] - np.linalg.solve(A, B[r])) / np.linalg.norm(X[r]) <= 1e-9
    lu.close()


@pytest.mark.parametrize("n", [5, 130, 1000])
def test_lu_factorize_then_solve_and_lu_solve(gpu, n):
    """lu_factorize + LuFactorization::solve (lu.rs:38-137) and lu_solve with untouched inputs (lu.rs:142-153)."""
    A, b = _rand(n, 3 * n)
    A0 = A.copy(); b0 = b.copy()
    x = ma.lu_solve(A, b)
    assert np.array_equal(A, A0) and np.array_equal(b, b0)
    assert np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x)) <= 1e-14 * max(n, 10)
    assert np.linalg.norm(x - ma.zgesv(A.copy(), b.copy())) <= 1e-12 * np.linalg.norm(x)
    F = ma.LuFactorization(A)
    rng = np.random.default_rng(n)
    for _ in range(3):
        c = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        y = F.solve(c)
        assert np.linalg.norm(A @ y - c) / (np.linalg.norm(A) * np.linalg.norm(y)) <= 1e-14 * max(n, 10)
    assert np.linalg.norm(F.solve(b) - x) <= 1e-12 * np.linalg.norm(x)
    F.close()
    with pytest.raises(ma.MaError) as e:
        ma.LuFactorization(np.ones((4, 4)))
    assert e.value.status == ma.MA_ERR_SINGULAR


_SCHEDULES = {"default": {}, "pair": {"MA_LU_REG_PANEL": 2, "MA_LU_CU_SPLIT": 64}, "pair_tail": {"MA_LU_REG_PANEL": 2, "MA_LU_CU_SPLIT": 64, "MA_LU_TAIL_ROWS": 400},
              "pair_block": {"MA_LU_REG_PANEL": 2, "MA_LU_CU_SPLIT": 64, "MA_LU_BLOCK_STEP": 1}}


@pytest.mark.parametrize("sched", ["default", "pair", "pair_tail", "pair_block"])
def test_staged_pipeline_is_bitwise_the_single_solve(gpu, sched):
    """The staged plan API (slots at their own block index, staggered by a fraction of a factorisation) runs the same kernels on
    the same data as a single factor+solve: seven systems through three slots, every factor and solution bit for bit. `pair`: the
    schedule plans of 4 096-16 384 rows get by default (64-column panels as two register half-panels, the fused step kernels, the
    big updates on a stream masked off 64 CUs, which the driver then uses as its own); `pair_tail`: with the last blocks' whole
    updates on the lanes."""
    import torch
    n = 900
    dev = torch.device("cuda", 0)
    mats = [_rand(n, 300 + i) for i in range(7)]
    with _with_env(**_SCHEDULES[sched]):
        lu = ma.LuPlan(n)
    st = torch.cuda.current_stream().cuda_stream
    if sched != "default":
        assert lu.main_stream()
        st = lu.main_stream()
    singles = []
    for A, b in mats:
        dA = torch.tensor(A, device=dev).reshape(-1); db = torch.tensor(b, device=dev)
        lu.factor_solve_dev(dA.data_ptr(), db.data_ptr(), 1, st)
        assert lu.status(st) == ma.MA_OK
        singles.append((dA.cpu().numpy().copy(), db.cpu().numpy().copy()))
    S = 3
    G = lu.num_blocks()
    assert G >= 3
    bufA = [torch.empty(n * n, dtype=torch.complex128, device=dev) for _ in range(S)]
    bufB = [torch.empty(n, dtype=torch.complex128, device=dev) for _ in range(S)]
    srcA = [torch.tensor(A, device=dev).reshape(-1) for A, _ in mats]; srcB = [torch.tensor(b, device=dev) for _, b in mats]
    outA = [None] * len(mats); outB = [None] * len(mats)
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
            if idx >= len(mats):
                continue
            live = True
            if g == 0:
                bufA[s].copy_(srcA[idx]); bufB[s].copy_(srcB[idx])
                lu.stage_begin(s, bufA[s].data_ptr(), bufB[s].data_ptr(), 1, st)
            sl.append(s); bl.append(g)
        if not live:
            break
        if sl:
            lu.stage_round(sl, bl, st)
        for s, g in zip(sl, bl):
            if g == G - 1:
                lu.stage_finish(s, st)
                idx = s + S * ((r - off[s]) // G)
                outA[idx] = bufA[s].clone(); outB[idx] = bufB[s].clone()
        r += 1
    assert lu.status(st) == ma.MA_OK
    for i, (Af, xf) in enumerate(singles):
        assert np.array_equal(outA[i].cpu().numpy(), Af), i
        assert np.array_equal(outB[i].cpu().numpy(), xf), i
    with pytest.raises(ma.MaError):
        lu.stage_round([0], [G], st)                       # block index out of range
    lu.close()


def test_staged_groups_are_bitwise_the_single_solve(gpu):
    """The staged API with GROUPS: two groups of three slots, each group in lock step with ONE panel kernel per panel for its
    three systems (lu_panel_wave_kernel, a wavefront per system, 32-column panels), the groups staggered by half a
    factorisation. Twelve systems; every factor and solution bit for bit what a single factor+solve with 32-column panels gives."""
    import torch
    n = 900
    dev = torch.device("cuda", 0)
    mats = [_rand(n, 500 + i) for i in range(12)]
    st = torch.cuda.current_stream().cuda_stream
    with _with_env(MA_LU_NB=32):
        lu1 = ma.LuPlan(n)
    singles = []
    for A, b in mats:
        dA = torch.tensor(A, device=dev).reshape(-1); db = torch.tensor(b, device=dev)
        lu1.factor_solve_dev(dA.data_ptr(), db.data_ptr(), 1, st)
        assert lu1.status(st) == ma.MA_OK
        singles.append((dA.cpu().numpy().copy(), db.cpu().numpy().copy()))
    lu1.close()
    lu = ma.LuPlan(n)
    gsz, U = 3, 2
    lu.stage_set_group(gsz)
    G = lu.num_blocks()
    S = gsz * U
    bufA = [torch.empty(n * n, dtype=torch.complex128, device=dev) for _ in range(S)]
    bufB = [torch.empty(n, dtype=torch.complex128, device=dev) for _ in range(S)]
    srcA = [torch.tensor(A, device=dev).reshape(-1) for A, _ in mats]; srcB = [torch.tensor(b, device=dev) for _, b in mats]
    outA = [None] * len(mats); outB = [None] * len(mats)
    off = [u * (G // U) for u in range(U)]
    lu.stage_reset(st)
    r = 0
    while True:
        sl, bl, live = [], [], False
        for u in range(U):
            lr = r - off[u]
            if lr < 0:
                live = True
                continue
            sysno, g = divmod(lr, G)
            base = (u + U * sysno) * gsz
            if base + gsz > len(mats):
                continue
            live = True
            if g == 0:
                for t in range(gsz):
                    s_ = u * gsz + t
                    bufA[s_].copy_(srcA[base + t]); bufB[s_].copy_(srcB[base + t])
                    lu.stage_begin(s_, bufA[s_].data_ptr(), bufB[s_].data_ptr(), 1, st)
                lu.stage_begin_group(u * gsz, st)
            for t in range(gsz):
                sl.append(u * gsz + t); bl.append(g)
        if not live:
            break
        if sl:
            lu.stage_round(sl, bl, st)
        for s_, g in zip(sl, bl):
            if g == G - 1:
                lu.stage_finish(s_, st)
                u, t = divmod(s_, gsz)
                idx = (u + U * ((r - off[u]) // G)) * gsz + t
                outA[idx] = bufA[s_].clone(); outB[idx] = bufB[s_].clone()
        r += 1
    assert lu.status(st) == ma.MA_OK
    for i, (Af, xf) in enumerate(singles):
        assert np.array_equal(outA[i].cpu().numpy(), Af), i
        assert np.array_equal(outB[i].cpu().numpy(), xf), i
    with pytest.raises(ma.MaError):
        lu.stage_round([0, 1], [0, 0], st)                 # a group must appear whole
    lu.close()


def test_lu_tall_system_switches_panel_width(gpu):
    """Above 36 352 rows a 64-column panel no longer fits the LDS of the co-resident workgroups: the factorisation starts with
    32-column panels (8 per trailing update) and widens to 64 once the remaining rows fit. 36 900 rows cross that boundary;
    the residual of the device solve is checked with a device matvec (21 GB matrix + copy: everything stays in HBM)."""
    import torch
    n = 36900
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7)
    A = torch.complex(torch.randn(n * n, dtype=torch.float64, device=dev, generator=g), torch.randn(n * n, dtype=torch.float64, device=dev, generator=g))
    b = torch.complex(torch.randn(n, dtype=torch.float64, device=dev, generator=g), torch.randn(n, dtype=torch.float64, device=dev, generator=g))
    A0 = A.clone(); x = b.clone()
    lu = ma.LuPlan(n)
    st = torch.cuda.current_stream().cuda_stream
    lu.factor_solve_dev(A.data_ptr(), x.data_ptr(), 1, st)
    assert lu.status(st) == ma.MA_OK
    r = torch.mv(A0.reshape(n, n), x) - b
    res = float(r.norm() / (A0.reshape(n, n)[:64].norm() * (n / 64) ** 0.5 * x.norm()))
    assert res <= 1e-14 * n
    lu.close()


def _with_env(**kv):
    """Context manager: the library reads its tuning / test switches once per plan (getenv at ma_lu_plan_create)."""
    import contextlib, os

    @contextlib.contextmanager
    def cm():
        old = {k: os.environ.get(k) for k in kv}
        os.environ.update({k: str(v) for k, v in kv.items()})
        try:
            yield
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return cm()


@pytest.mark.parametrize("batch_panel", [0, 1])
def test_abandoned_panel_poisons_the_plan_and_nothing_else(gpu, batch_panel):
    """The failure path behind the round-1 memory fault (DESIGN 4 "Residency", lu_kernels.hip): a panel kernel whose exchange
    does not complete must (1) make every workgroup of every panel kernel of the plan leave at once -- the poison word is
    read in every poll, no 4 s wait per workgroup --, (2) leave identity pivots behind, so that the interchange kernels,
    which are already enqueued, move no rows, (3) surface as MA_ERR_HIP, and (4) leave the plan usable after the next
    factorisation clears the word. The test hook makes the last workgroup of the panel that owns global column 200 give up
    in a 3-system batch of 11 panels each; every later kernel of all three systems still runs, on garbage but in bounds."""
    import time
    import torch
    n = 700
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    mats = [_rand(n, 300 + i) for i in range(3)]
    with _with_env(MA_LU_TEST_ABORT_COL=200, MA_LU_BATCH_PANEL=batch_panel):      # 1: one panel kernel for the three systems, a wavefront each
        lu = ma.LuPlan(n)
    dAs = [torch.tensor(A, device=dev).reshape(-1) for A, _ in mats]; dbs = [torch.tensor(b, device=dev) for _, b in mats]
    guard = torch.full((1 << 20,), 7.0, dtype=torch.float64, device=dev)       # a canary allocated right after the operands
    t0 = time.perf_counter()
    lu.factor_solve_batch_dev([t.data_ptr() for t in dAs], [t.data_ptr() for t in dbs], 1, st)
    rc = lu.status(st)
    dt = time.perf_counter() - t0
    assert rc == ma.MA_ERR_HIP
    assert b"abandoned" in ma.lib().ma_last_error_string()
    assert dt < 2.0, dt                                    # nobody sat out the 4 s limit
    assert float(guard.min()) == 7.0 and float(guard.max()) == 7.0
    lu.close()
    # the same three systems on a plan without the hook: bit for bit the single solves
    lu = ma.LuPlan(n)
    dAs = [torch.tensor(A, device=dev).reshape(-1) for A, _ in mats]; dbs = [torch.tensor(b, device=dev) for _, b in mats]
    lu.factor_solve_batch_dev([t.data_ptr() for t in dAs], [t.data_ptr() for t in dbs], 1, st)
    assert lu.status(st) == ma.MA_OK
    for (A, b), db in zip(mats, dbs):
        x = db.cpu().numpy()
        assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-11
    lu.close()


def test_nan_column_and_tiny_pivot_inside_a_batch(gpu):
    """A column of NaNs in the fourth panel of one system of a 3-system batch: no workgroup offers a candidate, the diagonal
    row stands in, the system is reported singular (MA_ERR_SINGULAR) -- and the two healthy systems of the batch still hold
    their exact solutions. Then lu.rs:106-110: a pivot column whose largest |z| is below 1e-30 is LuError::SingularMatrix
    (the LAPACK path only errors on an exact zero; the stricter of the two is reported)."""
    import torch
    n = 700
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    mats = [_rand(n, 400 + i) for i in range(3)]
    bad = mats[1][0].copy(); bad[:, 200] = np.nan
    lu = ma.LuPlan(n)
    dAs = [torch.tensor(A if i != 1 else bad, device=dev).reshape(-1) for i, (A, _) in enumerate(mats)]
    dbs = [torch.tensor(b, device=dev) for _, b in mats]
    lu.factor_solve_batch_dev([t.data_ptr() for t in dAs], [t.data_ptr() for t in dbs], 1, st)
    assert lu.status(st) == ma.MA_ERR_SINGULAR
    assert b"system 1" in ma.lib().ma_last_error_string()
    for i in (0, 2):
        x = dbs[i].cpu().numpy()
        assert np.linalg.norm(mats[i][0] @ x - mats[i][1]) / np.linalg.norm(mats[i][1]) < 1e-11
    lu.close()
    # a numerically singular system: column 3 scaled to 1e-40
    A, b = _rand(6, 9)
    A[:, 3] *= 1e-40
    with pytest.raises(ma.MaError) as e:
        ma.zgesv(A, b)
    assert e.value.status == ma.MA_ERR_SINGULAR
    A2, b2 = _rand(6, 9)
    A2[:, 3] *= 1e-20                                      # small but above the threshold: solved
    x = ma.zgesv(A2, b2)
    assert np.linalg.norm(A2 @ x - b2) / np.linalg.norm(b2) < 1e-6


def test_round1_fault_configuration_now_runs(gpu):
    """MA_LU_RPB=32 with 128-column panels and three systems in flight is the configuration of gpurun_out/bench_rpb32.log
    (70 KB of LDS per spinning workgroup). The old admission counted floor(160 KB / 70 KB) = 2 slots per CU and let two
    256-workgroup grids in; the fragmentation-safe count is 1 (lu_kernels.hip "Residency"), so the grids now run one at a
    time -- and complete."""
    import torch
    n = 6000
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    As = [torch.randn(n, n, dtype=torch.complex128, device=dev, generator=gen) for _ in range(3)]
    bs = [torch.randn(n, dtype=torch.complex128, device=dev, generator=gen) for _ in range(3)]
    with _with_env(MA_LU_RPB=32, MA_LU_NB=128):
        lu = ma.LuPlan(n)
    dAs = [A.clone().reshape(-1) for A in As]; dbs = [b.clone() for b in bs]
    lu.factor_solve_batch_dev([t.data_ptr() for t in dAs], [t.data_ptr() for t in dbs], 1, st)
    assert lu.status(st) == ma.MA_OK, ma.lib().ma_last_error_string()
    for A, b, x in zip(As, bs, dbs):
        assert float(torch.linalg.norm(A @ x - b) / torch.linalg.norm(b)) < 1e-10
    lu.close()
