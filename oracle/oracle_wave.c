/* oracle_wave.c — CPU ORACLE (test infrastructure, NOT product code).
 * Rigid-sphere Mie series of math-wave (analytical/solutions_3d.rs:56-273), restated in C.
 */
#include "ma_oracle.h"
#include <math.h>
#include <stdlib.h>

#define PI 3.14159265358979323846264338327950288
static inline mao_c64 C(double re, double im) { mao_c64 z = {re, im}; return z; }
static inline mao_c64 cmul(mao_c64 a, mao_c64 b) { return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline mao_c64 cdiv(mao_c64 a, mao_c64 b) {
  double ns = b.re * b.re + b.im * b.im;
  return C((a.re * b.re + a.im * b.im) / ns, (a.im * b.re - a.re * b.im) / ns);
}

double mao_spherical_bessel_j(int n, double x) {        /* solutions_3d.rs:191-224 */
  if (fabs(x) < 1e-10) return n == 0 ? 1.0 : 0.0;
  if (n == 0) return sin(x) / x;
  if (n == 1) return sin(x) / (x * x) - cos(x) / x;
  int start = n + (int)fabs(x) + 20;
  double* v = (double*)calloc((size_t)start + 1, sizeof(double));
  double jn = 0.0, jc = 1e-30;
  v[start] = jc;
  for (int k = start - 1; k >= 0; --k) {
    double jp = (double)(2 * k + 3) / x * jc - jn;
    v[k] = jp; jn = jc; jc = jp;
  }
  double scale = (sin(x) / x) / v[0];
  double r = v[n] * scale;
  free(v);
  return r;
}

double mao_spherical_bessel_y(int n, double x) {        /* solutions_3d.rs:229-251 */
  if (fabs(x) < 1e-10) return -INFINITY;
  if (n == 0) return -cos(x) / x;
  if (n == 1) return -cos(x) / (x * x) - sin(x) / x;
  double y2 = -cos(x) / x, y1 = -cos(x) / (x * x) - sin(x) / x;
  for (int k = 2; k <= n; ++k) { double yn = (double)(2 * k - 1) / x * y1 - y2; y2 = y1; y1 = yn; }
  return y1;
}

double mao_legendre_p(int n, double x) {                /* solutions_3d.rs:256-273 */
  if (n == 0) return 1.0;
  if (n == 1) return x;
  double p2 = 1.0, p1 = x;
  for (int k = 2; k <= n; ++k) { double pn = ((double)(2 * k - 1) * x * p1 - (double)(k - 1) * p2) / (double)k; p2 = p1; p1 = pn; }
  return p1;
}

/* sphere_rcs_3d (solutions_3d.rs:278-288): 4 pi / k^2 sum (2n + 1) |a_n|^2 with compute_rigid_sphere_coefficients (:147-184) */
double mao_sphere_rcs_3d(double k, double radius, int T) {
  const double ka = k * radius;
  double rcs = 0.0;
  for (int n = 0; n < T; ++n) {
    double nf = (double)n;
    double jn = mao_spherical_bessel_j(n, ka), yn = mao_spherical_bessel_y(n, ka);
    double jm = n > 0 ? mao_spherical_bessel_j(n - 1, ka) : cos(ka) / ka;
    double jp = jm - (nf + 1.0) / ka * jn;
    double ym = n > 0 ? mao_spherical_bessel_y(n - 1, ka) : -sin(ka) / ka;
    double yp = ym - (nf + 1.0) / ka * yn;
    mao_c64 a = cdiv(C(jp, 0.0), C(jp, yp));
    rcs += (double)(2 * n + 1) * (a.re * a.re + a.im * a.im);
  }
  return 4.0 * PI * rcs / (k * k);
}

void mao_sphere_scattering_3d(double k, double radius, int T, int nr, const double* r, int nt, const double* th, mao_c64* out) {
  double ka = k * radius;
  mao_c64* a = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)(T > 0 ? T : 1));
  for (int n = 0; n < T; ++n) {                         /* solutions_3d.rs:147-184 */
    double nf = (double)n;
    double jn = mao_spherical_bessel_j(n, ka), yn = mao_spherical_bessel_y(n, ka);
    double jm = n > 0 ? mao_spherical_bessel_j(n - 1, ka) : cos(ka) / ka;
    double jp = jm - (nf + 1.0) / ka * jn;
    double ym = n > 0 ? mao_spherical_bessel_y(n - 1, ka) : -sin(ka) / ka;
    double yp = ym - (nf + 1.0) / ka * yn;
    a[n] = cdiv(C(jp, 0.0), C(jp, yp));
  }
  int o = 0;
  for (int ir = 0; ir < nr; ++ir)
    for (int it = 0; it < nt; ++it) {                   /* solutions_3d.rs:72-106 */
      double kr = k * r[ir], ct = cos(th[it]);
      mao_c64 tot = C(0.0, 0.0);
      for (int n = 0; n < T; ++n) {
        double nf = (double)n;
        double pre = 2.0 * nf + 1.0;
        mao_c64 ipn = C(cos(nf * PI / 2.0), sin(nf * PI / 2.0));
        double jn = mao_spherical_bessel_j(n, kr), yn = mao_spherical_bessel_y(n, kr);
        mao_c64 hn = C(jn, yn);
        double pn = mao_legendre_p(n, ct);
        mao_c64 ah = cmul(a[n], hn);
        mao_c64 inner = C(jn - ah.re, -ah.im);          /* f64 - Complex */
        mao_c64 t1 = C(pre * ipn.re, pre * ipn.im);     /* f64 * Complex */
        mao_c64 term = cmul(t1, inner);
        tot.re += term.re * pn; tot.im += term.im * pn;
      }
      out[o++] = tot;
    }
  free(a);
}
