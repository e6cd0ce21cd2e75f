"""Diagnostic: random sparse patterns (unsymmetric, empty rows, missing and zero diagonals, banded and scattered) through the three sweep
schedules; the default (the new value is its own flag) must equal a launch per level bit for bit, every time, and never abandon a wait.
usage: python tools/stress_sweeps.py [seed] [seconds]"""
import sys, os, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import math_audio_amd as ma
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t0 = time.time(); count = 0; persistent = 0
while time.time() - t0 < budget:
    n = int(rng.integers(40, 6000))
    kind = int(rng.integers(0, 3))
    if kind == 0:                                   # banded: long dependency chains
        bw = int(rng.integers(1, 6))
        diags = [rng.standard_normal(n - abs(o)) + 1j * rng.standard_normal(n - abs(o)) for o in range(-bw, bw + 1)]
        A = sp.diags(diags, list(range(-bw, bw + 1)), format="lil")
    else:                                           # scattered
        A = sp.random(n, n, density=min(1.0, float(rng.uniform(2.0, 9.0)) / n), random_state=int(rng.integers(1 << 30)), format="lil").astype(np.complex128)
    A = A.tocsr().astype(np.complex128)
    A.data = rng.standard_normal(A.nnz) + 1j * rng.standard_normal(A.nnz)
    A = (A + sp.diags(np.where(rng.random(n) < 0.9, 6.0 + rng.standard_normal(n), 0.0))).tocsr()      # some rows without a diagonal
    if kind == 2:                                   # a few empty rows
        A = A.tolil()
        for r in rng.integers(0, n, size=3):
            A.rows[r] = []; A.data[r] = []
        A = A.tocsr()
    A.sort_indices()
    if A.nnz == 0:
        continue
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n); x0 = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    res = {}
    for mode in ("flags", "launches"):
        os.environ["MA_CSR_GS_FLAGS"] = "1" if mode == "flags" else "0"
        h = ma.CsrOperator(A.indptr.astype(np.int64), A.indices.astype(np.int64), values=A.data)
        fwd, bwd = h.gauss_seidel_levels()
        res[mode] = (h.sym_gauss_seidel(x0, b, 2), h.fem_smooth(x0, b, kind=0, iterations=2) if hasattr(h, "fem_smooth") else None)
        ma.check(ma.lib().ma_csr_status(h.h))
        h.close()
    persistent += int(min(fwd, bwd) >= 8)
    assert np.array_equal(res["flags"][0], res["launches"][0], equal_nan=True), (n, kind, count)
    if res["flags"][1] is not None:
        assert np.array_equal(res["flags"][1], res["launches"][1], equal_nan=True), (n, kind, count, "fem_smooth")
    count += 1
print("patterns", count, "of which with >= 8 levels (persistent launch used)", persistent)
