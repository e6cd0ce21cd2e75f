"""Handles give back what they took: every plan / operator / preconditioner kind is created and destroyed in a loop and the free
device memory returns to where it started (hipMemGetInfo through torch). A leak of one level buffer per create would show as a
steady drift."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
import oracle_lib as O
import math_audio_amd as ma
from math_audio_amd import fem
from helpers import to_ma_mesh, RADIUS
from fmm_clusters import grid_clusters

pytestmark = pytest.mark.gpu


def _free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _loop(make, rounds=100):
    make()                                       # first use may grow pools (module load, torch context)
    base = _free()
    for _ in range(rounds):
        make()
    return base - _free()


def test_handles_release_their_device_memory(gpu):
    om = O.icosphere(RADIUS, 2)
    k = 1.0 / RADIUS
    cl = grid_clusters(om.center, 0.07)
    mesh = to_ma_mesh(om)
    nodes, rp, ci, K, M = fem.helmholtz_box(8, 6, 5)
    n = len(rp) - 1
    r = np.ones(n, dtype=np.complex128)

    def slfmm():
        plan = ma.BemPlan(mesh)
        op = ma.LinearOperator.slfmm(plan, cl, k, 4, 8, 5)
        op.apply(np.ones(om.n_elem, dtype=np.complex128))
        op.close(); plan.close()

    def mlfmm():
        plan = ma.BemPlan(mesh)
        tree = ma.ClusterTree(mesh, 20, k)
        op = ma.LinearOperator.mlfmm(plan, tree, k)
        op.apply(np.ones(om.n_elem, dtype=np.complex128))
        op.close(); tree.close(); plan.close()

    def sparse_preconditioners():
        op = ma.CsrOperator(rp, ci, K=K, M=M); op.set_wavenumber(1.2 + 0.01j)
        for make in (lambda: ma.AmgFromCsr(op, ma.AmgConfig.preset("for_parallel", coarse_size=20)), lambda: ma.IluPreconditioner(op),
                     lambda: ma.IluFixedPointPreconditioner(op, 3), lambda: ma.AdditiveSchwarzPreconditioner(op, 4, 1),
                     lambda: ma.Preconditioner(op, "jacobi"), lambda: ma.Preconditioner(op, "sgs")):
            P = make(); P.apply(r); P.close()
        lin = ma.LinearOperator.csr(op); lin.close()
        op.close()

    def dense():
        A = np.eye(64, dtype=np.complex128) * 3.0 + 0.1
        lu = ma.LuPlan(64)
        op = ma.LinearOperator.dense(A)
        ma.gmres(op, np.ones(64, dtype=np.complex128), restart=10, max_iterations=20, tol=1e-10)
        op.close(); lu.close()

    slack = 4 << 20                              # allocator granularity: a few MiB either way; 100 rounds turn a 64 KiB leak per create into 6 MiB
    for name, fn in (("slfmm", slfmm), ("mlfmm", mlfmm), ("sparse preconditioners", sparse_preconditioners), ("dense", dense)):
        lost = _loop(fn)
        assert lost <= slack, "%s: %d bytes not returned after 100 create / destroy rounds" % (name, lost)
