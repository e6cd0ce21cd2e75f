"""TEST INFRASTRUCTURE (the oracle): CPU restatement of the reference's AMG setup, AmgPreconditioner::from_csr
(math-solvers/src/preconditioners/amg.rs:276-372) with the CSR algebra it calls (math-solvers/src/sparse/csr.rs). Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may import this; the product path never does.

Plain Python over lists (complex is Python's: a*b = (ar br - ai bi, ar bi + ai br), as num_complex'; norm = sqrt(re^2 + im^2),
traits.rs:94-96; inv = (re / |z|^2, -im / |z|^2), traits.rs:159-162). Every function names the lines it follows.
Pinned by the reference's own unit tests (amg.rs:1158-1266), restated in tests/test_oracle.py: parity of the setup is otherwise
unpinned at entry level (no reference run is possible here).
"""
import math


def cnorm(z):                                   # traits.rs:94-96
    return math.sqrt(z.real * z.real + z.imag * z.imag)


def cinv(z):                                    # traits.rs:159-162
    d = z.real * z.real + z.imag * z.imag
    return complex(z.real / d, -z.imag / d)


class Csr:
    """CsrMatrix<Complex64> (csr.rs:18-35): num_rows, num_cols, row_ptrs, col_indices, values."""

    def __init__(self, nr, nc, ptr=None, col=None, val=None):
        self.nr, self.nc = nr, nc
        self.ptr = ptr if ptr is not None else [0] * (nr + 1)
        self.col = col if col is not None else []
        self.val = val if val is not None else []

    def nnz(self):
        return len(self.val)

    def row(self, i):                           # row_entries, csr.rs:228-236
        return zip(self.col[self.ptr[i]:self.ptr[i + 1]], self.val[self.ptr[i]:self.ptr[i + 1]])

    def get(self, i, j):                        # csr.rs:340-347: the first stored (i, j), zero if none
        for q in range(self.ptr[i], self.ptr[i + 1]):
            if self.col[q] == j:
                return self.val[q]
        return 0j


def from_triplets(nr, nc, trip):
    """csr.rs:135-205: stable sort by (row, col), equal (row, col) accumulate in that order."""
    if not trip:
        return Csr(nr, nc)
    trip = sorted(trip, key=lambda t: (t[0], t[1]))
    ptr = [0] * (nr + 1); col = []; val = []
    prev = (-1, -1)
    counts = [0] * nr
    for r, c, v in trip:
        if (r, c) == prev:
            val[-1] += v
        else:
            val.append(v); col.append(c); counts[r] += 1; prev = (r, c)
    for i in range(nr):
        ptr[i + 1] = ptr[i] + counts[i]
    return Csr(nr, nc, ptr, col, val)


def matmul(a, b):
    """csr.rs:594-651: per row the products in (k, j) storage order, stable sort by column, summed in that order, entries with
    norm <= 1e-15 dropped."""
    if a.nr == 0 or b.nc == 0 or a.nnz() == 0 or b.nnz() == 0:
        return Csr(a.nr, b.nc)
    trip = []
    for i in range(a.nr):
        rd = []
        for k, aik in a.row(i):
            for j, bkj in b.row(k):
                rd.append((j, aik * bkj))
        if not rd:
            continue
        rd.sort(key=lambda t: t[0])
        cj, cv = rd[0]
        for j, v in rd[1:]:
            if j == cj:
                cv += v
            else:
                if cnorm(cv) > 1e-15:
                    trip.append((i, cj, cv))
                cj, cv = j, v
        if cnorm(cv) > 1e-15:
            trip.append((i, cj, cv))
    return from_triplets(a.nr, b.nc, trip)


def transpose(m):                               # amg.rs:810-822
    trip = []
    for i in range(m.nr):
        for j, v in m.row(i):
            trip.append((j, i, v))
    return from_triplets(m.nc, m.nr, trip)


def compute_diag_inv(m):                        # amg.rs:400-413
    out = [1 + 0j] * m.nr
    for i in range(m.nr):
        d = m.get(i, i)
        if cnorm(d) > 1e-15:
            out[i] = cinv(d)
    return out


def strength(m, theta):                         # amg.rs:418-474
    strong = []
    for i in range(m.nr):
        mx = 0.0
        for j, v in m.row(i):
            if i != j:
                nv = cnorm(v)
                if nv > mx:
                    mx = nv
        thr = theta * mx
        strong.append([j for j, v in m.row(i) if i != j and cnorm(v) >= thr])
    return strong


UNDECIDED, COARSE, FINE = 0, 1, 2


def coarsen_ruge_stuben(m, strong):
    """amg.rs:477-532. The reference sorts the points once by decreasing lambda (stable) and walks that order; the lambda updates
    inside the walk change nothing afterwards. Its scan `for j in 0..n if strong[j].contains(i)` is the transposed strength
    graph, used here directly (same result, not O(n^2))."""
    n = m.nr
    pt = [UNDECIDED] * n
    lam = [0] * n
    st = [[] for _ in range(n)]
    for i in range(n):
        for j in strong[i]:
            lam[j] += 1
            st[j].append(i)
    order = sorted(range(n), key=lambda a: -lam[a])
    for i in order:
        if pt[i] != UNDECIDED:
            continue
        pt[i] = COARSE
        for j in st[i]:
            if pt[j] == UNDECIDED:
                pt[j] = FINE
    for i in range(n):
        if pt[i] == UNDECIDED:
            pt[i] = FINE
    return pt, [i for i in range(n) if pt[i] == COARSE]


def coarsen_pmis(m, strong):                    # amg.rs:535-642 (both builds read the previous pass' state)
    n = m.nr
    pt = [UNDECIDED] * n
    w = [len(strong[i]) + (i * 0.0001) % 0.001 for i in range(n)]
    changed = True; it = 0
    while changed and it < 100:
        changed = False; it += 1
        old = list(pt)
        for i in range(n):
            if old[i] != UNDECIDED:
                continue
            is_max = True
            for j in strong[i]:
                if old[j] == UNDECIDED and w[j] > w[i]:
                    is_max = False
                    break
            has_c = any(old[j] == COARSE for j in strong[i])
            if has_c:
                pt[i] = FINE; changed = True
            elif is_max:
                pt[i] = COARSE; changed = True
    for i in range(n):
        if pt[i] == UNDECIDED:
            pt[i] = COARSE
    return pt, [i for i in range(n) if pt[i] == COARSE]


STANDARD, EXTENDED, DIRECT = 0, 1, 2          # the declaration order of AmgInterpolation (amg.rs:58-70)


def build_interpolation(m, strong, pt, c2f, interpolation, trunc_factor, max_interp_elements):   # amg.rs:645-807
    nf = m.nr
    f2c = [-1] * nf
    for ci, fi in enumerate(c2f):
        f2c[fi] = ci
    trip = []
    for i in range(nf):
        if pt[i] == COARSE:
            trip.append((i, f2c[i], 1 + 0j))
            continue
        if pt[i] != FINE:
            continue
        aii = m.get(i, i)
        cn = [j for j in strong[i] if pt[j] == COARSE]
        if not cn:
            continue
        if interpolation in (DIRECT, STANDARD):
            weights = []; sw = 0j
            for j in cn:
                aij = m.get(i, j)
                if cnorm(aii) > 1e-15:
                    wv = 0j - aij * cinv(aii)
                    weights.append([f2c[j], wv]); sw += wv
            if interpolation == STANDARD:
                weak = 0j
                for j, v in m.row(i):
                    if j != i and j not in cn:
                        weak += v
                if cnorm(sw) > 1e-15 and cnorm(weak) > 1e-15:
                    scale = (1 + 0j) + weak * cinv(aii * sw)
                    for e in weights:
                        e[1] *= scale
            if trunc_factor > 0.0:
                mw = 0.0
                for _, wv in weights:
                    if cnorm(wv) > mw:
                        mw = cnorm(wv)
                thr = trunc_factor * mw
                weights = [e for e in weights if cnorm(e[1]) >= thr]
                if len(weights) > max_interp_elements:
                    weights.sort(key=lambda e: -cnorm(e[1]))
                    weights = weights[:max_interp_elements]
            for ci, wv in weights:
                trip.append((i, ci, wv))
        else:
            weights = []
            for j in cn:
                aij = m.get(i, j)
                if cnorm(aii) > 1e-15:
                    weights.append([f2c[j], 0j - aij * cinv(aii)])
            for k in [j for j in strong[i] if pt[j] == FINE]:
                aik = m.get(i, k); akk = m.get(k, k)
                if cnorm(akk) < 1e-15:
                    continue
                for j in strong[k]:
                    if pt[j] == COARSE:
                        akj = m.get(k, j)
                        wv = 0j - aik * akj * cinv(aii * akk)
                        cj = f2c[j]
                        for e in weights:
                            if e[0] == cj:
                                e[1] += wv
                                break
                        else:
                            weights.append([cj, wv])
            if len(weights) > max_interp_elements:
                weights.sort(key=lambda e: -cnorm(e[1]))
                weights = weights[:max_interp_elements]
            for ci, wv in weights:
                trip.append((i, ci, wv))
    return from_triplets(nf, len(c2f), trip)


RUGE_STUBEN, PMIS, HMIS = 0, 1, 2


def default_config():                           # amg.rs:148-167
    return dict(coarsening=RUGE_STUBEN, interpolation=STANDARD, smoother=0, cycle=0, strong_threshold=0.25, max_levels=25, coarse_size=50,
                num_pre_smooth=1, num_post_smooth=1, jacobi_weight=0.6667, trunc_factor=0.0, max_interp_elements=4,
                aggressive_coarsening_levels=0)


def preset(name):                               # amg.rs:173-218
    c = default_config()
    if name == "bem":
        c.update(strong_threshold=0.5, coarsening=PMIS, smoother=1, max_interp_elements=6)
    elif name == "fem":
        c.update(strong_threshold=0.25, coarsening=RUGE_STUBEN, smoother=2)
    elif name == "parallel":
        c.update(coarsening=PMIS, smoother=0, jacobi_weight=0.8, num_pre_smooth=2, num_post_smooth=2)
    elif name == "difficult":
        c.update(coarsening=RUGE_STUBEN, interpolation=EXTENDED, smoother=2, strong_threshold=0.25, max_interp_elements=8, num_pre_smooth=2,
                 num_post_smooth=2)
    return c


def from_csr(matrix, config):
    """amg.rs:276-372: list of levels {A, P, R, diag_inv, coarse_to_fine, point_types} (the coarsest without P / R) and the two
    complexities (:837-853)."""
    levels = [dict(A=matrix, P=None, R=None, diag_inv=compute_diag_inv(matrix), coarse_to_fine=[], point_types=None)]
    cur = matrix
    for _ in range(config["max_levels"] - 1):
        n = cur.nr
        if n <= config["coarse_size"]:
            break
        strong = strength(cur, config["strong_threshold"])
        if config["coarsening"] == RUGE_STUBEN:
            pt, c2f = coarsen_ruge_stuben(cur, strong)
        else:
            pt, c2f = coarsen_pmis(cur, strong)
        ncoarse = len(c2f)
        if ncoarse == 0 or ncoarse >= n:
            break
        P = build_interpolation(cur, strong, pt, c2f, config["interpolation"], config["trunc_factor"], config["max_interp_elements"])
        R = transpose(P)
        Ac = matmul(R, matmul(cur, P))          # galerkin_product, amg.rs:825-828
        levels[-1].update(P=P, R=R, coarse_to_fine=c2f, point_types=pt)
        levels.append(dict(A=Ac, P=None, R=None, diag_inv=compute_diag_inv(Ac), coarse_to_fine=[], point_types=None))
        cur = Ac
    fd = float(levels[0]["A"].nr); fn = float(levels[0]["A"].nnz())
    gc = sum(float(l["A"].nr) for l in levels) / fd
    oc = sum(float(l["A"].nnz()) for l in levels) / fn if fn > 0 else 1.0
    return levels, gc, oc


def laplacian_1d(n):                            # the matrix of the reference's tests, amg.rs:1142-1156
    trip = []
    for i in range(n):
        trip.append((i, i, 2 + 0j))
        if i > 0:
            trip.append((i, i - 1, -1 + 0j))
        if i < n - 1:
            trip.append((i, i + 1, -1 + 0j))
    return from_triplets(n, n, trip)


def from_scipy(M):
    M = M.tocsr(); M.sort_indices()
    return Csr(M.shape[0], M.shape[1], [int(v) for v in M.indptr], [int(v) for v in M.indices], [complex(v) for v in M.data])


def to_scipy(m):
    import numpy as np, scipy.sparse as sp
    return sp.csr_matrix((np.array(m.val, dtype=np.complex128), np.array(m.col, dtype=np.int64), np.array(m.ptr, dtype=np.int64)), shape=(m.nr, m.nc))
