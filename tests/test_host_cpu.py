"""Host-side helpers (mesh generators, physics scalars, sweep sharding) — CPU only."""
import os
import sys
import numpy as np
import pytest
import oracle_lib as O
from math_audio_amd import mesh as mm
from math_audio_amd import sweep


@pytest.mark.parametrize("sub", [0, 1, 2, 3])
def test_icosphere_matches_restatement_bit_for_bit(sub):
    a = mm.generate_icosphere_mesh(0.1, sub); b = O.icosphere(0.1, sub)
    assert np.array_equal(a.nodes, b.nodes) and np.array_equal(a.conn, b.conn)
    assert np.array_equal(a.center, b.center) and np.array_equal(a.normal, b.normal) and np.array_equal(a.area, b.area)


def test_uv_sphere_matches_restatement():
    """Same formula, same libm: identical up to the last bit of a few sin/cos values (a compiler may or may
    not fuse sin and cos of one angle into a sincos call); connectivity is identical."""
    a = mm.generate_sphere_mesh(0.1, 13, 20); b = O.uv_sphere(0.1, 13, 20)
    assert np.array_equal(a.conn, b.conn)
    assert np.abs(a.nodes - b.nodes).max() <= 2e-17
    assert np.abs(a.center - b.center).max() <= 2e-17 and np.abs(a.normal - b.normal).max() <= 1e-15
    assert np.abs(a.area - b.area).max() <= 1e-18
    s10 = mm.generate_sphere_mesh(0.1, 51, 100)
    assert s10.n_elem == 10000 and s10.nodes.shape[0] == 5002


def test_physics_scalars():
    k = mm.wave_number(545.9, 343.0)
    assert k == O.wave_number(545.9, 343.0)
    assert mm.burton_miller_beta_scaled(k, 4.0) == O.beta_scaled(k, 4.0)
    for kk in (2.0, 10.0, 15.0, 30.0):
        assert mm.burton_miller_beta_adaptive(kk, 0.1) == O.beta_adaptive(kk, 0.1)
    f = mm.log_space(100.0, 8000.0, 64)
    assert len(f) == 64 and abs(f[0] - 100.0) < 1e-9 and abs(f[-1] - 8000.0) < 1e-6
    assert all(f[i] < f[i + 1] for i in range(63)) and abs(f[1] / f[0] - f[33] / f[32]) < 1e-12


def test_shard_frequencies_partitions_the_sweep():
    for world in (1, 2, 3, 4, 8):
        seen = []
        for r in range(world):
            seen += sweep.shard_frequencies(64, r, world)
        assert sorted(seen) == list(range(64))
    assert sweep.shard_frequencies(64, 3, 8) == list(range(3, 64, 8))
    with pytest.raises(ValueError):
        sweep.shard_frequencies(64, 2, 2)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nf, width = 7, 5
    mine = sweep.shard_frequencies(nf, rank, world)
    vals = [np.full(width, f + 1) * (1 + 0.5j) + rank * 0 for f in mine]    # value depends on the frequency only
    table = sweep.gather_results(mine, vals, nf, width, dist=dist)
    tmax = sweep.max_over_ranks(0.25 + rank, dist=dist)
    dist.destroy_process_group()
    q.put((rank, table, tmax))


def test_two_process_gloo_sweep_gather():
    """world_size = 2 on CPU: the sharded sweep plus final gather reproduces the single-process table,
    and the timing reduce is the MAX over ranks (bench.py contract)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.array([np.full(5, f + 1) * (1 + 0.5j) for f in range(7)])
    for rank, table, tmax in res:
        assert np.array_equal(table, expect)
        assert tmax == 1.25


# ---------------------------------------------------------------- row-sharded operator + replicated GMRES (config #5, C2)
def _sharded_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from math_audio_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(17)                       # same matrix on every rank
    n = 91                                                # not a multiple of the world size: the last block is short
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) + 40.0 * np.eye(n)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    r0, r1 = sharded.row_block(n, rank, world)
    At = torch.tensor(A[r0:r1])

    def local_apply(x, y_block):                          # stands in for the HIP row-block operator on the CPU test
        y_block.copy_(At @ x)
    op = sharded.ShardedOperator(n, local_apply, dist=dist)
    y = op.apply(torch.tensor(b)).numpy()
    x, info = sharded.gmres(op, torch.tensor(b), restart=20, max_iterations=10, tol=1e-10)
    dist.destroy_process_group()
    q.put((rank, (r0, r1), y, x.numpy(), info))


def test_two_process_gloo_row_sharded_gmres():
    """world_size = 2 on CPU: the all-gathered product equals A x, and the replicated GMRES follows the restatement of
    gmres.rs iteration for iteration on both ranks."""
    import torch.multiprocessing as mp
    import oracle_lib as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(17)
    n = 91
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) + 40.0 * np.eye(n)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    x_ref, info_ref = O.gmres(b, dense=A, restart=20, max_iterations=10, tol=1e-10)
    blocks = sorted(r[1] for r in res)
    assert blocks == [(0, 46), (46, 91)]
    for rank, _, y, x, info in res:
        assert np.abs(y - A @ b).max() <= 1e-12 * np.abs(A @ b).max()
        assert info["converged"] and info_ref.converged == 1
        assert info["iterations"] == info_ref.iterations and info["restarts"] == info_ref.restarts
        assert np.abs(x - x_ref).max() <= 1e-10 * np.abs(x_ref).max()
    assert np.array_equal(res[0][3], res[1][3])           # replicated Krylov vectors: the ranks agree bit for bit


def test_row_block_partition():
    from math_audio_amd import sharded
    for n, w in ((50172, 8), (10, 4), (3, 8), (64, 1)):
        blocks = [sharded.row_block(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
        assert max(b1 - b0 for b0, b1 in blocks) == (n + w - 1) // w
