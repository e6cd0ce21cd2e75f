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
