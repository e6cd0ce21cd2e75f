"""Rank form of BASELINE config #5 on the library's own device GMRES: two PROCESSES share one MI355X (the box has one), each owns a row
block of the matrix-free TBEM operator behind ma_op_create_gathered; the exchange callback all-gathers the blocks over gloo (host
staging; with the "nccl" backend = RCCL the same callback gathers device tensors). Every rank must run the single-GPU ma_gmres'
iteration: same counts, same solution."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import torch
        import torch.distributed as dist
        import oracle_lib as O
        import math_audio_amd as ma
        from math_audio_amd import sharded
        from helpers import to_ma_mesh, k_from_ka, RADIUS
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        om = O.icosphere(RADIUS, 2)
        k = k_from_ka(1.0); beta, _ = O.beta_adaptive(k, RADIUS)
        plan = ma.BemPlan(to_ma_mesh(om))
        so = sharded.tbem_sharded_operator(plan, k, beta, dist=dist, device=dev)
        assert so.lib_op is not None and (so.r0, so.r1) == sharded.row_block(plan.num_dofs, rank, world)
        b = ma.incident_rhs(om.center, om.normal, k, beta)
        xs, info = sharded.gmres(so, torch.tensor(b, device=dev), restart=30, max_iterations=10, tol=1e-8)
        # the same system on the whole operator in this process
        op = ma.LinearOperator.tbem(plan, k, beta)
        xr, info_r = ma.gmres(op, b, restart=30, max_iterations=10, tol=1e-8)
        ok = (info["converged"] and info_r.converged == 1 and info["iterations"] == info_r.iterations and info["restarts"] == info_r.restarts
              and np.linalg.norm(xs.cpu().numpy() - xr) <= 1e-9 * np.linalg.norm(xr))
        q.put((rank, bool(ok), info["iterations"], info_r.iterations, float(np.linalg.norm(xs.cpu().numpy() - xr) / np.linalg.norm(xr))))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # report instead of hanging the peer
        q.put((rank, False, -1, -1, repr(e)))


def test_two_ranks_one_gpu_library_gmres(gpu):
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(2):
            res.append(q.get(timeout=240))
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert len(res) == 2 and all(r[1] for r in res), res


def test_rccl_gather_inside_the_library_world_of_one(gpu):
    """ma_op_create_gathered_rccl: the row exchange of a rank-sharded operator as ncclAllGather inside the library,
    on a communicator made through ma_rccl_get_unique_id / ma_rccl_comm_create. One GPU gives a world of one (RCCL refuses two ranks
    on one device): the block is the whole operator, the collective runs on the apply's stream, and GMRES over the handle must be
    the plain operator's iteration. The row-block rule (per = ceil(n / nranks)) is checked against sharded.row_block.
    UNMEASURED on more than one GPU."""
    import oracle_lib as O
    import math_audio_amd as ma
    from math_audio_amd import sharded
    from helpers import to_ma_mesh, k_from_ka, RADIUS
    om = O.icosphere(RADIUS, 2)
    n = om.n_elem
    k = k_from_ka(1.0); beta, _ = O.beta_adaptive(k, RADIUS)
    plan = ma.BemPlan(to_ma_mesh(om))
    uid = ma.rccl_unique_id()
    assert len(uid) == 128
    comm = ma.RcclComm(1, 0, uid, device=0)
    inner = ma.LinearOperator.tbem(plan, k, beta, rows=sharded.row_block(n, 0, 1))
    gop = ma.LinearOperator.gathered_rccl(inner, comm, 1, 0)
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    assert np.array_equal(gop.apply(x), inner.apply(x))
    b = ma.incident_rhs(om.center, om.normal, k, beta)
    xg, ig = ma.gmres(gop, b, restart=30, max_iterations=10, tol=1e-8)
    xr, ir = ma.gmres(inner, b, restart=30, max_iterations=10, tol=1e-8)
    assert ig.converged == 1 and ig.iterations == ir.iterations and np.array_equal(xg, xr)
    # a row block that is not this rank's is refused
    wrong = ma.LinearOperator.tbem(plan, k, beta, rows=(0, n // 2))
    with pytest.raises(ma.MaError) as e:
        ma.LinearOperator.gathered_rccl(wrong, comm, 1, 0)
    assert e.value.status == ma.MA_ERR_INVALID
    for o in (gop, inner, wrong):
        o.close()
    comm.close(); plan.close()
