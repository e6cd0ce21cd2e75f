"""oracle_krylov.py -- CPU restatement (numpy) of the reference's BiCGSTAB, CGS and CG. TEST INFRASTRUCTURE ONLY.

Follows math-solvers/src/iterative/bicgstab.rs:46-182, cgs.rs:46-139, cg.rs:49-138 statement by statement; inner_product and
vector_norm are blas_helpers.rs:21-50 (<x, y> = sum conj(x_i) y_i accumulated in index order). `apply` is the operator's
LinearOperator::apply. Each returns (x, iterations, residual, converged)."""
import numpy as np


def inner_product(x, y):
    s = 0.0 + 0.0j
    for a, b in zip(x, y):
        s += np.conj(a) * b
    return complex(s)


def vector_norm(x):
    s = 0.0
    for a in x:
        s += a.real * a.real + a.imag * a.imag
    return float(np.sqrt(s))


def bicgstab(apply, b, max_iterations=1000, tolerance=1e-6):
    b = np.asarray(b, dtype=np.complex128)
    n = len(b)
    x = np.zeros(n, dtype=np.complex128)
    b_norm = vector_norm(b)
    if b_norm < 1e-15:
        return x, 0, 0.0, True
    r = b.copy(); r0 = r.copy()
    rho = alpha = omega = 1.0 + 0.0j
    p = np.zeros(n, dtype=np.complex128); v = np.zeros(n, dtype=np.complex128)
    for it in range(max_iterations):
        rho_new = inner_product(r0, r)
        if abs(rho_new) < 1e-30:
            return x, it, vector_norm(r) / b_norm, False
        beta = (rho_new / rho) * (alpha / omega)
        rho = rho_new
        p = r + (p - v * omega) * beta
        v = apply(p)
        r0v = inner_product(r0, v)
        if abs(r0v) < 1e-30:
            return x, it, vector_norm(r) / b_norm, False
        alpha = rho / r0v
        s = r - v * alpha
        s_norm = vector_norm(s)
        if s_norm / b_norm < tolerance:
            return x + p * alpha, it + 1, s_norm / b_norm, True
        t = apply(s)
        tt = inner_product(t, t)
        if abs(tt) < 1e-30:
            return x, it, vector_norm(r) / b_norm, False
        omega = inner_product(t, s) / tt
        x = x + p * alpha + s * omega
        r = s - t * omega
        rel = vector_norm(r) / b_norm
        if rel < tolerance:
            return x, it + 1, rel, True
        if abs(omega) < 1e-30:
            return x, it + 1, rel, False
    return x, max_iterations, vector_norm(r) / b_norm, False


def cgs(apply, b, max_iterations=1000, tolerance=1e-6):
    b = np.asarray(b, dtype=np.complex128)
    n = len(b)
    x = np.zeros(n, dtype=np.complex128)
    b_norm = vector_norm(b)
    if b_norm < 1e-15:
        return x, 0, 0.0, True
    r = b.copy(); r0 = r.copy()
    rho = inner_product(r0, r)
    p = r.copy(); u = r.copy()
    for it in range(max_iterations):
        v = apply(p)
        sigma = inner_product(r0, v)
        if abs(sigma) < 1e-30:
            return x, it, vector_norm(r) / b_norm, False
        alpha = rho / sigma
        q = u - v * alpha
        upq = u + q
        w = apply(upq)
        x = x + upq * alpha
        r = r - w * alpha
        rel = vector_norm(r) / b_norm
        if rel < tolerance:
            return x, it + 1, rel, True
        rho_new = inner_product(r0, r)
        if abs(rho) < 1e-30:
            return x, it + 1, rel, False
        beta = rho_new / rho
        rho = rho_new
        u = r + q * beta
        p = u + (q + p * beta) * beta
    return x, max_iterations, vector_norm(r) / b_norm, False


def cg(apply, b, max_iterations=1000, tolerance=1e-6):
    b = np.asarray(b, dtype=np.complex128)
    n = len(b)
    x = np.zeros(n, dtype=np.complex128)
    b_norm = vector_norm(b)
    if b_norm < 1e-15:
        return x, 0, 0.0, True
    r = b.copy(); p = r.copy()
    rho = inner_product(r, r)
    for it in range(max_iterations):
        q = apply(p)
        pq = inner_product(p, q)
        if abs(pq) < 1e-30:
            return x, it, vector_norm(r) / b_norm, False
        alpha = rho / pq
        x = x + p * alpha
        r = r - q * alpha
        rel = vector_norm(r) / b_norm
        if rel < tolerance:
            return x, it + 1, rel, True
        rho_new = inner_product(r, r)
        if abs(rho) < 1e-30:
            return x, it + 1, rel, False
        beta = rho_new / rho
        rho = rho_new
        p = r + p * beta
    return x, max_iterations, vector_norm(r) / b_norm, False
