"""Checks against ANALYTIC solutions that no restatement by this repository's author stands behind: the sound-soft
sphere (pressure boundary condition) and the rigid sphere on an all-Quad4 mesh. A backend provides
  solve(mesh, k, beta) -> solution vector of the TBEM system with the plane-wave right-hand side (mesh.bc_type decides the condition),
  scattered(mesh, k, points, surface_pressure, surface_velocity) -> scattered field at the points."""
import numpy as np
import scipy.special as ss

RADIUS = 0.1


def soft_sphere_total_field(k, a, r, theta, terms=40):
    """p = sum (2n + 1) i^n [j_n(kr) - j_n(ka) / h_n(ka) h_n(kr)] P_n(cos theta): plane wave e^{ikz} on a sound-soft sphere."""
    out = np.zeros(len(theta), dtype=complex)
    for n in range(terms):
        jn = ss.spherical_jn(n, k * r); hn = jn + 1j * ss.spherical_yn(n, k * r)
        ja = ss.spherical_jn(n, k * a); ha = ja + 1j * ss.spherical_yn(n, k * a)
        out += (2 * n + 1) * (1j ** n) * (jn - ja / ha * hn) * ss.eval_legendre(n, np.cos(theta))
    return out


def soft_sphere_surface_dpdn(k, a, theta, terms=40):
    out = np.zeros(len(theta), dtype=complex)
    for n in range(terms):
        dj = ss.spherical_jn(n, k * a, derivative=True); ja = ss.spherical_jn(n, k * a)
        ha = ja + 1j * ss.spherical_yn(n, k * a); dh = dj + 1j * ss.spherical_yn(n, k * a, derivative=True)
        out += (2 * n + 1) * (1j ** n) * k * (dj - ja / ha * dh) * ss.eval_legendre(n, np.cos(theta))
    return out


def rigid_sphere_surface_pressure(k, a, theta, terms=40, reference_series=False):
    """p(a, theta) = sum (2n + 1) i^n [j_n(ka) - j_n'(ka) / h_n'(ka) h_n(ka)] P_n(cos theta) with SciPy's Bessel functions.
    reference_series=True: the series the reference's tests compare with (math-wave solutions_3d.rs:147-184), whose n = 0 term takes
    y_{-1}(x) = -sin(x) / x for the derivative of y_0 (the identity is +sin(x) / x): its a_0 is not the rigid sphere's."""
    out = np.zeros(len(theta), dtype=complex)
    ka = k * a
    for n in range(terms):
        ja = ss.spherical_jn(n, ka); ha = ja + 1j * ss.spherical_yn(n, ka)
        dj = ss.spherical_jn(n, ka, derivative=True); dy = ss.spherical_yn(n, ka, derivative=True)
        if reference_series and n == 0:
            dy = -np.sin(ka) / ka - (1.0 / ka) * ss.spherical_yn(0, ka)
        out += (2 * n + 1) * (1j ** n) * (ja - dj / (dj + 1j * dy) * ha) * ss.eval_legendre(n, np.cos(theta))
    return out


def soft_sphere_errors(backend, mesh, ka, incident_pressure):
    """(max relative error of |p| at r = 2a on 19 angles, mean of x / analytic dp/dn over the panels): the unknown of a pressure-type
    panel is dp/dn with the sign of compute_scattered_field's velocity term (pressure.rs:154-258)."""
    a = RADIUS
    k = ka / a
    x = backend.solve(mesh, k, complex(0.0, 4.0 / k))
    th = np.linspace(0.0, np.pi, 19); r = 2.0 * a
    pts = np.stack([r * np.sin(th), 0.0 * th, r * np.cos(th)], axis=1)
    tot = incident_pressure(pts, k) + backend.scattered(mesh, k, pts, np.zeros(mesh.n_elem, dtype=complex), x)
    ana = soft_sphere_total_field(k, a, r, th)
    field_err = float(np.max(np.abs(np.abs(tot) - np.abs(ana)) / np.abs(ana)))
    c = np.asarray(mesh.center); tha = np.arccos(np.clip(c[:, 2] / np.linalg.norm(c, axis=1), -1.0, 1.0))
    ratio = x / soft_sphere_surface_dpdn(k, a, tha)
    return field_err, complex(np.mean(ratio))


def rigid_surface_error(backend, mesh, ka):
    """(relative L2 distance of the surface solution from the series the reference's tests use, the same from the rigid sphere's true series)"""
    a = RADIUS
    k = ka / a
    x = backend.solve(mesh, k, complex(0.0, 4.0 / k))
    c = np.asarray(mesh.center); th = np.arccos(np.clip(c[:, 2] / np.linalg.norm(c, axis=1), -1.0, 1.0))
    ref = rigid_sphere_surface_pressure(k, a, th, reference_series=True)
    true = rigid_sphere_surface_pressure(k, a, th)
    return float(np.linalg.norm(x - ref) / np.linalg.norm(ref)), float(np.linalg.norm(x - true) / np.linalg.norm(true))
