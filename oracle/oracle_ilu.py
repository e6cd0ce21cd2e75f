"""oracle_ilu.py -- CPU restatement (numpy / plain loops) of the reference's ILU(0) preconditioner. TEST INFRASTRUCTURE ONLY.

Follows math-solvers/src/preconditioners/ilu.rs: IluPreconditioner::from_csr (:36-140: in-place factorisation on the matrix' own
pattern, pivots below 1e-30 skipped, the row-k lookup first at the entry right of the diagonal and then by a scan of row k, split
into a strictly lower L with unit diagonal and an upper U with its diagonal kept aside) and apply (:143-175: forward substitution,
backward substitution, division by u_ii only when |u_ii| > 1e-30)."""
import numpy as np


class IluPreconditioner:
    def __init__(self, row_ptrs, col_indices, values):
        rp = [int(v) for v in row_ptrs]; ci = [int(v) for v in col_indices]
        val = [complex(v) for v in values]
        n = len(rp) - 1
        self.n = n
        none = -1
        diag = [none] * n
        for i in range(n):
            for idx in range(rp[i], rp[i + 1]):
                if ci[idx] == i:
                    diag[i] = idx
                    break
        for i in range(n):
            for idx in range(rp[i], rp[i + 1]):
                k = ci[idx]
                if k >= i:
                    break
                ukk = diag[k]
                if ukk == none:
                    continue
                u_kk = val[ukk]
                if abs(u_kk) < 1e-30:
                    continue
                l_ik = val[idx] * (u_kk.conjugate() / (u_kk.real * u_kk.real + u_kk.imag * u_kk.imag))      # * u_kk.inv()
                val[idx] = l_ik
                for jx in range(rp[i], rp[i + 1]):
                    j = ci[jx]
                    if j <= k:
                        continue
                    first = diag[k] + 1
                    if first < rp[k + 1] and ci[first] == j:
                        val[jx] = val[jx] - l_ik * val[first]
                    else:
                        for sx in range(rp[k] + 1, rp[k + 1]):
                            if ci[sx] == j:
                                val[jx] = val[jx] - l_ik * val[sx]
                                break
        self.l_ptr, self.l_col, self.l_val = [0], [], []
        self.u_ptr, self.u_col, self.u_val = [0], [], []
        self.u_diag = [1.0 + 0.0j] * n
        for i in range(n):
            for idx in range(rp[i], rp[i + 1]):
                j = ci[idx]
                if j < i:
                    self.l_col.append(j); self.l_val.append(val[idx])
                else:
                    self.u_col.append(j); self.u_val.append(val[idx])
                    if j == i:
                        self.u_diag[i] = val[idx]
            self.l_ptr.append(len(self.l_val)); self.u_ptr.append(len(self.u_val))

    def apply(self, r):
        y = [complex(v) for v in r]
        for i in range(self.n):
            for idx in range(self.l_ptr[i], self.l_ptr[i + 1]):
                y[i] = y[i] - self.l_val[idx] * y[self.l_col[idx]]
        x = y
        for i in range(self.n - 1, -1, -1):
            for idx in range(self.u_ptr[i], self.u_ptr[i + 1]):
                j = self.u_col[idx]
                if j > i:
                    x[i] = x[i] - self.u_val[idx] * x[j]
            u = self.u_diag[i]
            if abs(u) > 1e-30:
                x[i] = x[i] * (u.conjugate() / (u.real * u.real + u.imag * u.imag))
        return np.array(x, dtype=np.complex128)


class IluFixedPointPreconditioner:
    """math-solvers/src/preconditioners/ilu_parallel.rs:374-590: the ILU(0) factorisation again (:397-449, the row-k lookups by plain
    scans, pivots below 1e-30 skipped), split into L (strictly lower), U_off (strictly upper) and 1 / u_ii (1 where |u_ii| <= 1e-30);
    apply (:510-548): x = D^-1 r, then `iterations` times x <- D^-1 (r - (L + U_off) x). from_csr_default: 3 iterations (:492-494)."""

    def __init__(self, row_ptrs, col_indices, values, iterations=3):
        rp = [int(v) for v in row_ptrs]; ci = [int(v) for v in col_indices]
        val = [complex(v) for v in values]
        n = len(rp) - 1
        for i in range(n):
            for idx in range(rp[i], rp[i + 1]):
                k = ci[idx]
                if k >= i:
                    break
                u_kk = 0j
                for kx in range(rp[k], rp[k + 1]):
                    if ci[kx] == k:
                        u_kk = val[kx]
                        break
                if abs(u_kk) < 1e-30:
                    continue
                d = u_kk.real * u_kk.real + u_kk.imag * u_kk.imag
                l_ik = val[idx] * complex(u_kk.real / d, -u_kk.imag / d)
                val[idx] = l_ik
                for jx in range(rp[i], rp[i + 1]):
                    j = ci[jx]
                    if j <= k:
                        continue
                    for sx in range(rp[k], rp[k + 1]):
                        if ci[sx] == j:
                            val[jx] = val[jx] - l_ik * val[sx]
                            break
        self.n, self.iterations = n, int(iterations)
        self.rp, self.ci, self.val = rp, ci, val
        self.dinv = [1 + 0j] * n
        for i in range(n):
            for idx in range(rp[i], rp[i + 1]):
                if ci[idx] == i:
                    v = val[idx]
                    if abs(v) > 1e-30:
                        d = v.real * v.real + v.imag * v.imag
                        self.dinv[i] = complex(v.real / d, -v.imag / d)

    def apply(self, r):
        r = [complex(v) for v in r]
        x = [ri * di for ri, di in zip(r, self.dinv)]
        for _ in range(self.iterations):
            xn = [0j] * self.n
            for i in range(self.n):
                s = r[i]
                for idx in range(self.rp[i], self.rp[i + 1]):
                    j = self.ci[idx]
                    if j < i:
                        s -= self.val[idx] * x[j]
                for idx in range(self.rp[i], self.rp[i + 1]):
                    j = self.ci[idx]
                    if j > i:
                        s -= self.val[idx] * x[j]
                xn[i] = s * self.dinv[i]
            x = xn
        return np.array(x, dtype=np.complex128)


class AdditiveSchwarzPreconditioner:
    """math-solvers/src/preconditioners/schwarz.rs: from_csr (:84-145: contiguous blocks, extend_partition :196-229 `overlap` times along
    the matrix graph, weights 1 / count :112-128, build_subdomain + ilu_factorize :231-352) and apply_sequential (:394-408) with
    Subdomain::solve (:355-383: forward substitution, backward substitution, times 1 / u_ii when |u_ii| > 1e-30)."""

    def __init__(self, row_ptrs, col_indices, values, num_subdomains, overlap):
        rp = [int(v) for v in row_ptrs]; ci = [int(v) for v in col_indices]; val = [complex(v) for v in values]
        n = len(rp) - 1
        self.n = n
        ns = min(max(int(num_subdomains), 1), n)
        base, rem = n // ns, n % ns
        parts = []; start = 0
        for s in range(ns):
            size = base + (1 if s < rem else 0)
            parts.append(list(range(start, start + size))); start += size
        adj = [[ci[q] for q in range(rp[i], rp[i + 1]) if ci[q] != i] for i in range(n)]
        ext = []
        for part in parts:
            inp = [False] * n
            for i in part:
                inp[i] = True
            frontier = list(part)
            for _ in range(int(overlap)):
                new = []
                for i in frontier:
                    for nb in adj[i]:
                        if not inp[nb]:
                            inp[nb] = True; new.append(nb)
                frontier = new
            ext.append([i for i in range(n) if inp[i]])
        count = [0] * n
        for e in ext:
            for i in e:
                count[i] += 1
        self.weights = [1.0 / c if c > 0 else 1.0 for c in count]
        self.subdomains = []
        for gi in ext:
            g2l = {g: l for l, g in enumerate(gi)}
            lrp = [0]; lci = []; lv = []
            for g in gi:
                for q in range(rp[g], rp[g + 1]):
                    if ci[q] in g2l:
                        lci.append(g2l[ci[q]]); lv.append(val[q])
                lrp.append(len(lci))
            ln = len(gi)
            v = list(lv)
            for i in range(ln):                 # ilu_factorize, :253-352
                for idx in range(lrp[i], lrp[i + 1]):
                    k = lci[idx]
                    if k >= i:
                        break
                    u_kk = 0j
                    for kx in range(lrp[k], lrp[k + 1]):
                        if lci[kx] == k:
                            u_kk = v[kx]
                            break
                    if abs(u_kk) < 1e-30:
                        continue
                    d = u_kk.real * u_kk.real + u_kk.imag * u_kk.imag
                    l_ik = v[idx] * complex(u_kk.real / d, -u_kk.imag / d)
                    v[idx] = l_ik
                    for jx in range(lrp[i], lrp[i + 1]):
                        j = lci[jx]
                        if j <= k:
                            continue
                        for sx in range(lrp[k], lrp[k + 1]):
                            if lci[sx] == j:
                                v[jx] = v[jx] - l_ik * v[sx]
                                break
            udiag = [1 + 0j] * ln
            for i in range(ln):
                for idx in range(lrp[i], lrp[i + 1]):
                    if lci[idx] == i:
                        udiag[i] = v[idx]
            self.subdomains.append((gi, lrp, lci, v, udiag))

    def stats(self):
        sizes = [len(s[0]) for s in self.subdomains]
        return len(sizes), min(sizes), max(sizes), sum(sizes) / len(sizes)

    def apply(self, r):
        r = [complex(x) for x in r]
        out = [0j] * self.n
        for gi, lrp, lci, v, udiag in self.subdomains:
            ln = len(gi)
            y = [r[g] for g in gi]
            for i in range(ln):
                for idx in range(lrp[i], lrp[i + 1]):
                    j = lci[idx]
                    if j < i:
                        y[i] = y[i] - v[idx] * y[j]
            x = y
            for i in range(ln - 1, -1, -1):
                for idx in range(lrp[i], lrp[i + 1]):
                    j = lci[idx]
                    if j > i:
                        x[i] = x[i] - v[idx] * x[j]
                u = udiag[i]
                if abs(u) > 1e-30:
                    d = u.real * u.real + u.imag * u.imag
                    x[i] = x[i] * complex(u.real / d, -u.imag / d)
            for l, g in enumerate(gi):
                out[g] += x[l] * self.weights[g]
        return np.array(out, dtype=np.complex128)
