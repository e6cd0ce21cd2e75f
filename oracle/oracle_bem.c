/* oracle_bem.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the math-bem TBEM assembly path of the reference
 * (Rust). Arithmetic is written in the reference's operation order and built
 * with -ffp-contract=off. See ma_oracle.h for the pinning statement.
 */
#include "ma_oracle.h"
#include "ma_oracle_tables.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define PI 3.14159265358979323846264338327950288

/* ---- complex helpers with num_complex's formulas (no fma, no special cases) ---- */
static inline mao_c64 C(double re, double im) { mao_c64 z = {re, im}; return z; }
static inline mao_c64 cadd(mao_c64 a, mao_c64 b) { return C(a.re + b.re, a.im + b.im); }
static inline mao_c64 csub(mao_c64 a, mao_c64 b) { return C(a.re - b.re, a.im - b.im); }
static inline mao_c64 cneg(mao_c64 a) { return C(-a.re, -a.im); }
static inline mao_c64 cmul(mao_c64 a, mao_c64 b) {
  return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
static inline mao_c64 cscale(mao_c64 a, double s) { return C(a.re * s, a.im * s); }
static inline mao_c64 cdivr(mao_c64 a, double s) { return C(a.re / s, a.im / s); }
static inline double cnorm(mao_c64 a) { return hypot(a.re, a.im); }

/* ======================================================================
 * Quadrature tables — gauss.rs:15-105
 * ====================================================================== */
int mao_gauss_legendre(int order, double* x, double* w) {
  /* gauss.rs:15-60: exact table or the next table up; >20 -> 20 */
  if (order < 1) return 0;
  int idx = order > 20 ? 20 : order;
  int off = mao_gl_index[idx][0], n = mao_gl_index[idx][1];
  for (int i = 0; i < n; ++i) { x[i] = mao_gl_x[off + i]; w[i] = mao_gl_w[off + i]; }
  return n;
}

int mao_triangle_quadrature(int order, double* q) {
  /* gauss.rs:67-89: order 1->1pt, 2->4pt, 3->7pt, anything else ->13pt; w *= 0.5 */
  const double (*t)[3]; int n;
  switch (order) {
    case 1: t = mao_tri1; n = 1; break;
    case 2: t = mao_tri4; n = 4; break;
    case 3: t = mao_tri7; n = 7; break;
    default: t = mao_tri13; n = 13; break;
  }
  for (int i = 0; i < n; ++i) { q[3*i] = t[i][0]; q[3*i+1] = t[i][1]; q[3*i+2] = t[i][2] * 0.5; }
  return n;
}

int mao_quad_quadrature(int order, double* q) {
  /* gauss.rs:94-105: tensor rule, i outer / j inner */
  double x[20], w[20];
  int n = mao_gauss_legendre(order, x, w), c = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) { q[3*c] = x[i]; q[3*c+1] = x[j]; q[3*c+2] = w[i] * w[j]; ++c; }
  return c;
}

/* ======================================================================
 * Physics — types.rs:39-219
 * ====================================================================== */
double mao_wave_number(double f, double c) { double omega = 2.0 * PI * f; return omega / c; } /* types.rs:41-42 */
mao_c64 mao_burton_miller_beta(double k, double h, double tau) {            /* types.rs:64-70 */
  return tau > 0.0 ? C(0.0, h / k) : C(0.0, 0.0);
}
mao_c64 mao_burton_miller_beta_scaled(double k, double h, double tau, double s) { /* types.rs:144-150 */
  return tau > 0.0 ? C(0.0, h * s / k) : C(0.0, 0.0);
}
mao_c64 mao_burton_miller_beta_adaptive(double k, double h, double tau, double radius, double* scale_out) {
  /* types.rs:173-195 */
  if (tau <= 0.0) { if (scale_out) *scale_out = 1.0; return C(0.0, 0.0); }
  double ka = k * radius;
  double scale = ka < 0.5 ? 1.0 : (ka < 1.2 ? 4.0 : (ka < 1.8 ? 8.0 : 16.0));
  if (scale_out) *scale_out = scale;
  return C(0.0, h * scale / k);
}

/* ======================================================================
 * Mesh generators — mesh/generators.rs
 * ====================================================================== */
void mao_uv_sphere_counts(int nt, int np, int* nn, int* ne) {
  *nn = 2 + (nt - 1) * np;
  *ne = 2 * np + 2 * np * (nt - 2);
}

void mao_uv_sphere(double radius, int nt, int np, double* nodes, int* conn) {
  /* generators.rs:29-98 */
  int c = 0;
  nodes[0] = 0.0; nodes[1] = 0.0; nodes[2] = radius; c = 1;
  for (int i = 1; i < nt; ++i) {
    double theta = PI * (double)i / (double)nt;
    double st = sin(theta), ct = cos(theta);
    for (int j = 0; j < np; ++j) {
      double phi = 2.0 * PI * (double)j / (double)np;
      nodes[3*c] = radius * st * cos(phi);
      nodes[3*c+1] = radius * st * sin(phi);
      nodes[3*c+2] = radius * ct;
      ++c;
    }
  }
  nodes[3*c] = 0.0; nodes[3*c+1] = 0.0; nodes[3*c+2] = -radius;
  int south = c;
  int e = 0;
  for (int j = 0; j < np; ++j) {
    int jn = (j + 1) % np;
    conn[4*e] = 0; conn[4*e+1] = 1 + j; conn[4*e+2] = 1 + jn; conn[4*e+3] = -1; ++e;
  }
  for (int i = 0; i < nt - 2; ++i) {
    int rs = 1 + i * np, nrs = 1 + (i + 1) * np;
    for (int j = 0; j < np; ++j) {
      int jn = (j + 1) % np;
      int n0 = rs + j, n1 = rs + jn, n2 = nrs + j, n3 = nrs + jn;
      conn[4*e] = n0; conn[4*e+1] = n2; conn[4*e+2] = n1; conn[4*e+3] = -1; ++e;
      conn[4*e] = n1; conn[4*e+1] = n2; conn[4*e+2] = n3; conn[4*e+3] = -1; ++e;
    }
  }
  int lrs = 1 + (nt - 2) * np;
  for (int j = 0; j < np; ++j) {
    int jn = (j + 1) % np;
    conn[4*e] = lrs + j; conn[4*e+1] = south; conn[4*e+2] = lrs + jn; conn[4*e+3] = -1; ++e;
  }
}

void mao_icosphere_counts(int sub, int* nn, int* ne) {
  int f = 20, v = 12;
  for (int s = 0; s < sub; ++s) { v += f * 3 / 2; f *= 4; }
  *nn = v; *ne = f;
}

/* midpoint cache keyed by the sorted vertex pair (generators.rs:201-228); a hash map
 * whose iteration order is never used, so an open-addressing table is equivalent */
typedef struct { long long key; int val; } mp_slot;
static int get_midpoint(double* verts, int* nverts, mp_slot* tab, int cap, int v0, int v1) {
  int a = v0 < v1 ? v0 : v1, b = v0 < v1 ? v1 : v0;
  long long key = ((long long)a << 32) | (unsigned)b;
  unsigned long long h = (unsigned long long)key * 0x9E3779B97F4A7C15ull;
  int p = (int)(h % (unsigned long long)cap);
  while (tab[p].key != -1) {
    if (tab[p].key == key) return tab[p].val;
    p = (p + 1) % cap;
  }
  double mid[3];
  for (int d = 0; d < 3; ++d) mid[d] = (verts[3*v0+d] + verts[3*v1+d]) / 2.0;
  double len = sqrt(mid[0]*mid[0] + mid[1]*mid[1] + mid[2]*mid[2]);
  int idx = *nverts;
  for (int d = 0; d < 3; ++d) verts[3*idx+d] = mid[d] / len;
  *nverts = idx + 1;
  tab[p].key = key; tab[p].val = idx;
  return idx;
}

void mao_icosphere(double radius, int sub, double* nodes, int* conn) {
  /* generators.rs:110-198 */
  int nn, ne; mao_icosphere_counts(sub, &nn, &ne);
  double phi = (1.0 + sqrt(5.0)) / 2.0;
  double* v = (double*)malloc(sizeof(double) * 3 * (size_t)nn);
  const double v0[12][3] = {
    {-1.0, phi, 0.0}, {1.0, phi, 0.0}, {-1.0, -phi, 0.0}, {1.0, -phi, 0.0},
    {0.0, -1.0, phi}, {0.0, 1.0, phi}, {0.0, -1.0, -phi}, {0.0, 1.0, -phi},
    {phi, 0.0, -1.0}, {phi, 0.0, 1.0}, {-phi, 0.0, -1.0}, {-phi, 0.0, 1.0}};
  for (int i = 0; i < 12; ++i) {
    double len = sqrt(v0[i][0]*v0[i][0] + v0[i][1]*v0[i][1] + v0[i][2]*v0[i][2]);
    for (int d = 0; d < 3; ++d) v[3*i+d] = v0[i][d] / len;
  }
  int nv = 12;
  static const int f0[20][3] = {
    {0,11,5},{0,5,1},{0,1,7},{0,7,10},{0,10,11},{1,5,9},{5,11,4},{11,10,2},{10,7,6},{7,1,8},
    {3,9,4},{3,4,2},{3,2,6},{3,6,8},{3,8,9},{4,9,5},{2,4,11},{6,2,10},{8,6,7},{9,8,1}};
  int nf = 20;
  int* faces = (int*)malloc(sizeof(int) * 3 * (size_t)ne);
  int* nfaces = (int*)malloc(sizeof(int) * 3 * (size_t)ne);
  for (int i = 0; i < 20; ++i) for (int d = 0; d < 3; ++d) faces[3*i+d] = f0[i][d];
  for (int s = 0; s < sub; ++s) {
    int cap = nf * 4 + 17;
    mp_slot* tab = (mp_slot*)malloc(sizeof(mp_slot) * (size_t)cap);
    for (int i = 0; i < cap; ++i) tab[i].key = -1;
    int o = 0;
    for (int f = 0; f < nf; ++f) {
      int a = faces[3*f], b = faces[3*f+1], c = faces[3*f+2];
      int m01 = get_midpoint(v, &nv, tab, cap, a, b);
      int m12 = get_midpoint(v, &nv, tab, cap, b, c);
      int m20 = get_midpoint(v, &nv, tab, cap, c, a);
      int t[4][3] = {{a, m01, m20}, {b, m12, m01}, {c, m20, m12}, {m01, m12, m20}};
      for (int q = 0; q < 4; ++q) { for (int d = 0; d < 3; ++d) nfaces[3*o+d] = t[q][d]; ++o; }
    }
    free(tab);
    nf = o;
    int* tmp = faces; faces = nfaces; nfaces = tmp;
  }
  for (int i = 0; i < nv; ++i) for (int d = 0; d < 3; ++d) nodes[3*i+d] = v[3*i+d] * radius;
  for (int f = 0; f < nf; ++f) {
    conn[4*f] = faces[3*f]; conn[4*f+1] = faces[3*f+1]; conn[4*f+2] = faces[3*f+2]; conn[4*f+3] = -1;
  }
  free(v); free(faces); free(nfaces);
}

void mao_element_geometry(int ne, const double* nodes, const int* conn, double* center, double* normal, double* area) {
  /* generators.rs:513-602 */
  for (int e = 0; e < ne; ++e) {
    const int* cn = conn + 4*e;
    int n = cn[3] < 0 ? 3 : 4;
    double c[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < n; ++i) for (int j = 0; j < 3; ++j) c[j] += nodes[3*cn[i]+j];
    for (int j = 0; j < 3; ++j) c[j] /= (double)n;
    double a[3], b[3];
    if (n == 3) {
      for (int j = 0; j < 3; ++j) { a[j] = nodes[3*cn[1]+j] - nodes[3*cn[0]+j]; b[j] = nodes[3*cn[2]+j] - nodes[3*cn[0]+j]; }
    } else {
      for (int j = 0; j < 3; ++j) { a[j] = nodes[3*cn[2]+j] - nodes[3*cn[0]+j]; b[j] = nodes[3*cn[3]+j] - nodes[3*cn[1]+j]; }
    }
    double cr[3] = {a[1]*b[2] - a[2]*b[1], a[2]*b[0] - a[0]*b[2], a[0]*b[1] - a[1]*b[0]};
    double len = sqrt(cr[0]*cr[0] + cr[1]*cr[1] + cr[2]*cr[2]);
    area[e] = len / 2.0;
    double nrm[3] = {0.0, 0.0, 0.0};
    if (len > 1e-15) { nrm[0] = cr[0] / len; nrm[1] = cr[1] / len; nrm[2] = cr[2] / len; }
    double ndc = nrm[0]*c[0] + nrm[1]*c[1] + nrm[2]*c[2];
    if (ndc < 0.0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
    for (int j = 0; j < 3; ++j) { center[3*e+j] = c[j]; normal[3*e+j] = nrm[j]; }
  }
}

/* ======================================================================
 * Shape functions / geometry at a local point — regular.rs:193-260,
 * singular.rs:398-465 (identical bodies)
 * ====================================================================== */
static void shape_functions(int nn, double s, double t, double* N, double* ds, double* dt) {
  if (nn == 3) {
    N[0] = 1.0 - s - t; N[1] = s; N[2] = t;
    ds[0] = -1.0; ds[1] = 1.0; ds[2] = 0.0;
    dt[0] = -1.0; dt[1] = 0.0; dt[2] = 1.0;
  } else {
    double s1 = 0.25 * (s + 1.0), s2 = 0.25 * (s - 1.0), t1 = t + 1.0, t2 = t - 1.0;
    N[0] = s1 * t1; N[1] = -s2 * t1; N[2] = s2 * t2; N[3] = -s1 * t2;
    ds[0] = 0.25 * (t + 1.0); ds[1] = -0.25 * (t + 1.0); ds[2] = 0.25 * (t - 1.0); ds[3] = -0.25 * (t - 1.0);
    dt[0] = 0.25 * (s + 1.0); dt[1] = 0.25 * (1.0 - s); dt[2] = 0.25 * (s - 1.0); dt[3] = -0.25 * (s + 1.0);
  }
}

static void compute_parameters(const double* coords, int nn, double s, double t,
                               double* N, double* jac, double* el_norm, double* crd) {
  double ds[4], dt[4];
  shape_functions(nn, s, t, N, ds, dt);
  double dxs[3] = {0, 0, 0}, dxt[3] = {0, 0, 0};
  crd[0] = crd[1] = crd[2] = 0.0;
  for (int i = 0; i < nn; ++i)
    for (int j = 0; j < 3; ++j) {
      crd[j] += N[i] * coords[3*i+j];
      dxs[j] += ds[i] * coords[3*i+j];
      dxt[j] += dt[i] * coords[3*i+j];
    }
  double n[3] = {dxs[1]*dxt[2] - dxs[2]*dxt[1], dxs[2]*dxt[0] - dxs[0]*dxt[2], dxs[0]*dxt[1] - dxs[1]*dxt[0]};
  double j2 = ((n[0]*n[0]) + n[1]*n[1]) + n[2]*n[2];   /* normal.dot(&normal): sequential for len 3 */
  *jac = sqrt(j2);
  if (*jac > 1e-15) { el_norm[0] = n[0] / *jac; el_norm[1] = n[1] / *jac; el_norm[2] = n[2] / *jac; }
  else { el_norm[0] = el_norm[1] = el_norm[2] = 0.0; }
}

static inline double dot3(const double* a, const double* b) { return ((a[0]*b[0]) + a[1]*b[1]) + a[2]*b[2]; }

/* element.rs:124-131 normalize */
static double normalize3(const double* v, double* u) {
  double len = sqrt(dot3(v, v));
  if (len > 1e-15) { u[0] = v[0] / len; u[1] = v[1] / len; u[2] = v[2] / len; return len; }
  u[0] = u[1] = u[2] = 0.0; return 0.0;
}

/* ======================================================================
 * generate_subelements — singular.rs:497-693
 * ====================================================================== */
static const double CSI6[6] = {0.0, 1.0, 0.0, 0.5, 0.5, 0.0};
static const double ETA6[6] = {0.0, 0.0, 1.0, 0.0, 0.5, 0.5};
static const double CSI8[8] = {1.0, -1.0, -1.0, 1.0, 0.0, -1.0, 0.0, 1.0};
static const double ETA8[8] = {1.0, 1.0, -1.0, -1.0, 1.0, 0.0, -1.0, 0.0};

static double powi(double b, int e) { double r = 1.0; for (int i = 0; i < e; ++i) r *= b; return r; }
/* NOTE: Rust's f64::powi lowers to llvm.powi (repeated squaring); the exact rounding of the error
 * model only matters within 1 ulp of the 5e-4 threshold and the chosen order does not change Tri3
 * results (every order >= 4 maps to the 13-point rule, gauss.rs:84). */

static int compute_gauss_order(double disfac, int gmin, int gmax, double acc) {
  /* singular.rs:663-693 */
  for (int order = gmin; order <= gmax; ++order) {
    double n = (double)order;
    double base = disfac / (2.0 * n + 1.0);
    double eg = powi(base, 2 * order + 1), eh = powi(base, 2 * order + 2), ee = powi(base, 2 * order + 3);
    if (eg < acc && eh < acc && ee < acc) return order;
  }
  return gmax;
}

static void local_to_global(const double* coords, int nn, double s, double t, double* out) {
  /* singular.rs:696-721 */
  double N[4];
  if (nn == 3) { N[0] = 1.0 - s - t; N[1] = s; N[2] = t; }
  else {
    double s1 = 0.25 * (s + 1.0), s2 = 0.25 * (s - 1.0), t1 = t + 1.0, t2 = t - 1.0;
    N[0] = s1 * t1; N[1] = -s2 * t1; N[2] = s2 * t2; N[3] = -s1 * t2;
  }
  out[0] = out[1] = out[2] = 0.0;
  for (int i = 0; i < nn; ++i) for (int j = 0; j < 3; ++j) out[j] += N[i] * coords[3*i+j];
}

int mao_generate_subelements(const double* x, const double* coords, int nv, double area, mao_subelement* out) {
  enum { MAX_NSE = 60, NSE = 4 };
  const double TOL_F = 3.0; const int GAU_MAX = 7, GAU_MIN = 4; const double GAU_ACCU = 0.0005;
  int nres = 0;
  double xi_sfp[MAX_NSE][4], et_sfp[MAX_NSE][4];
  memset(xi_sfp, 0, sizeof xi_sfp); memset(et_sfp, 0, sizeof et_sfp);
  for (int i = 0; i < nv; ++i) {
    xi_sfp[0][i] = nv == 3 ? CSI6[i] : CSI8[i];
    et_sfp[0][i] = nv == 3 ? ETA6[i] : ETA8[i];
  }
  int nsfl = 1; double faclin = 2.0;
  for (;;) {
    int ndie = 0;
    faclin *= 0.5;
    double arels = area * faclin * faclin;
    int nsel = nsfl;
    double xi_sep[MAX_NSE][4], et_sep[MAX_NSE][4];
    memcpy(xi_sep, xi_sfp, sizeof(double) * 4 * (size_t)nsel);
    memcpy(et_sep, et_sfp, sizeof(double) * 4 * (size_t)nsel);
    for (int idi = 0; idi < nsel; ++idi) {
      double ssum = 0.0, tsum = 0.0;
      for (int i = 0; i < nv; ++i) { ssum += xi_sep[idi][i]; tsum += et_sep[idi][i]; }
      double scent = ssum / (double)nv, tcent = tsum / (double)nv;
      double P[3]; local_to_global(coords, nv, scent, tcent, P);
      double d[3] = {P[0] - x[0], P[1] - x[1], P[2] - x[2]};
      double dist = sqrt(dot3(d, d));
      double ratdis = dist / sqrt(arels);
      if (ratdis < TOL_F) {
        ndie += 1;
        if (ndie > 15) break;
        nsfl = ndie * NSE;
        int nsf0 = nsfl - NSE;
        double xisp[8], etsp[8];
        for (int j = 0; j < nv; ++j) {
          int j1 = (j + 1) % nv;
          xisp[j] = xi_sep[idi][j];
          xisp[j + nv] = (xi_sep[idi][j] + xi_sep[idi][j1]) / 2.0;
          etsp[j] = et_sep[idi][j];
          etsp[j + nv] = (et_sep[idi][j] + et_sep[idi][j1]) / 2.0;
        }
        for (int j = 0; j < nv; ++j) {
          int nsu = nsf0 + j;
          int j1 = j + nv;
          int j2 = j1 > nv ? j1 - 1 : j1 + nv - 1;
          if (nv == 4) {
            xi_sfp[nsu][0] = xisp[j]; xi_sfp[nsu][1] = xisp[j1]; xi_sfp[nsu][2] = scent; xi_sfp[nsu][3] = xisp[j2];
            et_sfp[nsu][0] = etsp[j]; et_sfp[nsu][1] = etsp[j1]; et_sfp[nsu][2] = tcent; et_sfp[nsu][3] = etsp[j2];
          } else {
            xi_sfp[nsu][0] = xisp[j]; xi_sfp[nsu][1] = xisp[j1]; xi_sfp[nsu][2] = xisp[j2];
            et_sfp[nsu][0] = etsp[j]; et_sfp[nsu][1] = etsp[j1]; et_sfp[nsu][2] = etsp[j2];
            if (j == nv - 1) {
              int nc = nsf0 + NSE - 1;
              xi_sfp[nc][0] = xisp[nv]; xi_sfp[nc][1] = xisp[nv+1]; xi_sfp[nc][2] = xisp[nv+2];
              et_sfp[nc][0] = etsp[nv]; et_sfp[nc][1] = etsp[nv+1]; et_sfp[nc][2] = etsp[nv+2];
            }
          }
        }
      } else {
        mao_subelement* se = &out[nres];
        if (nv == 4) {
          se->xi_center = (((xi_sep[idi][0] + xi_sep[idi][1]) + xi_sep[idi][2]) + xi_sep[idi][3]) / 4.0;
          se->eta_center = (((et_sep[idi][0] + et_sep[idi][1]) + et_sep[idi][2]) + et_sep[idi][3]) / 4.0;
          se->has_tri = 0; memset(se->tri, 0, sizeof se->tri);
        } else {
          se->xi_center = (xi_sep[idi][0] + xi_sep[idi][1] + xi_sep[idi][2]) / 3.0;
          se->eta_center = (et_sep[idi][0] + et_sep[idi][1] + et_sep[idi][2]) / 3.0;
          se->has_tri = 1;
          for (int i = 0; i < 3; ++i) { se->tri[2*i] = xi_sep[idi][i]; se->tri[2*i+1] = et_sep[idi][i]; }
        }
        se->factor = faclin;
        double disfac = 0.5 / ratdis;
        se->gauss_order = compute_gauss_order(disfac, GAU_MIN, GAU_MAX, GAU_ACCU);
        ++nres;
        if (nres >= MAO_MAX_SUBELEMENTS) return nres;
      }
    }
    if (ndie == 0) break;
  }
  return nres;
}

/* ======================================================================
 * regular_integration — regular.rs:33-182
 * ====================================================================== */
void mao_regular_integration(const double* x, const double* nx, const double* coords, int nn,
                             double area, double kwave, double harmonic, double tau,
                             const mao_c64* bc, int bc_len, int bc_type, int compute_rhs,
                             mao_integration_result* res) {
  double wavruim = harmonic * kwave;
  double k2 = kwave * kwave;
  memset(res, 0, sizeof *res);
  mao_subelement sub[MAO_MAX_SUBELEMENTS];
  int nsub = mao_generate_subelements(x, coords, nn, area, sub);
  double q[3 * 400];
  for (int is = 0; is < nsub; ++is) {
    const mao_subelement* se = &sub[is];
    double xice = se->xi_center, etce = se->eta_center, fase = se->factor, fase2 = fase * fase;
    int iforie = fabs(fabs(fase) - 1.0) < 1e-10;
    int nq = nn == 3 ? mao_triangle_quadrature(se->gauss_order, q) : mao_quad_quadrature(se->gauss_order, q);
    for (int iq = 0; iq < nq; ++iq) {
      double csi = q[3*iq], eta = q[3*iq+1], wei = q[3*iq+2];
      double xio, eto, weih2;
      if (iforie) { xio = csi; eto = eta; weih2 = wei; }
      else if (se->has_tri) {
        const double* tv = se->tri;
        double l0 = 1.0 - csi - eta;
        xio = tv[0] * l0 + tv[2] * csi + tv[4] * eta;
        eto = tv[1] * l0 + tv[3] * csi + tv[5] * eta;
        double dx1 = tv[2] - tv[0], dy1 = tv[3] - tv[1], dx2 = tv[4] - tv[0], dy2 = tv[5] - tv[1];
        double det = fabs(dx1 * dy2 - dx2 * dy1);
        weih2 = wei * det;
      } else { xio = xice + csi * fase; eto = etce + eta * fase; weih2 = wei * fase2; }

      double N[4], jac, ny[3], crd[3];
      compute_parameters(coords, nn, xio, eto, N, &jac, ny, crd);
      double wga = weih2 * jac;
      double diff[3] = {crd[0] - x[0], crd[1] - x[1], crd[2] - x[2]};
      double ur[3]; double r = normalize3(diff, ur);
      if (r < 1e-15) continue;

      double re1 = wavruim * r;
      double re2 = wga / (4.0 * PI * r);
      mao_c64 zg = C(cos(re1) * re2, sin(re1) * re2);
      mao_c64 z1 = C(-1.0 / r, wavruim);
      mao_c64 zhb = cmul(zg, z1);
      double re1h = dot3(ur, ny);
      mao_c64 zhh = cscale(zhb, re1h);
      double re2h = -dot3(ur, nx);
      mao_c64 zht = cscale(zhb, re2h);
      double rq = re1h * re2h;
      double nxny = dot3(nx, ny);
      double dq = r * r;
      mao_c64 zef = C((3.0 / dq - k2) * rq + nxny / dq, -wavruim / r * (3.0 * rq + nxny));
      mao_c64 ze = cmul(zg, zef);
      res->g = cadd(res->g, zg);
      res->dg_dn = cadd(res->dg_dn, zhh);
      res->dg_dnx = cadd(res->dg_dnx, zht);
      res->d2g = cadd(res->d2g, ze);

      if (compute_rhs && bc) {
        mao_c64 zb = C(0.0, 0.0);
        for (int i = 0; i < nn; ++i) if (i < bc_len) zb = cadd(zb, cscale(bc[i], N[i]));
        double gamma = 1.0;
        mao_c64 beta = mao_burton_miller_beta(kwave, harmonic, tau);   /* regular.rs:168 (unscaled beta: Appendix C) */
        if (bc_type == 0) {
          mao_c64 t = cadd(cscale(cscale(zg, gamma), tau), cmul(zht, beta));
          res->rhs = cadd(res->rhs, cmul(t, zb));
        } else if (bc_type == 1) {
          mao_c64 t = cadd(cscale(cscale(zhh, gamma), tau), cmul(ze, beta));
          res->rhs = csub(res->rhs, cmul(t, zb));
        }
      }
    }
  }
}

/* ======================================================================
 * singular_integration — singular.rs:48-82,123-394,730-745
 * ====================================================================== */
static double estimate_element_size(const double* coords, int nn) {
  double total = 0.0;
  for (int i = 0; i < nn; ++i) {
    int j = (i + 1) % nn;
    double e2 = 0.0;
    for (int k = 0; k < 3; ++k) { double d = coords[3*j+k] - coords[3*i+k]; e2 += d * d; }
    total += sqrt(e2);
  }
  return total / (double)nn;
}

void mao_singular_integration_with_params(const double* x, const double* nx, const double* coords, int nn,
        double kwave, double harmonic, double tau, const mao_c64* bc, int bc_len, int bc_type, int compute_rhs,
        int ngpo1, int ngausin, int nsec1, int nsec2, mao_integration_result* res) {
  double wavruim = harmonic * kwave;
  double k2 = kwave * kwave;
  memset(res, 0, sizeof *res);
  double cg[20], wg[20];
  int ne = mao_gauss_legendre(ngpo1, cg, wg);
  double gc[20], gw[20];
  int ns = mao_gauss_legendre(ngausin, gc, gw);
  const double* CS = nn == 3 ? CSI6 : CSI8;
  const double* ET = nn == 3 ? ETA6 : ETA8;

  for (int ieg = 0; ieg < nn; ++ieg) {
    int ig1 = (ieg + 1) % nn, ig2 = ieg + nn;
    double dpoi[3], leneg = 0.0;
    for (int i = 0; i < 3; ++i) { dpoi[i] = coords[3*ig1+i] - coords[3*ieg+i]; leneg += dpoi[i] * dpoi[i]; }
    leneg = sqrt(leneg);
    double dpoo[3] = {dpoi[0] / leneg, dpoi[1] / leneg, dpoi[2] / leneg};
    double lens = leneg / (2.0 * (double)nsec1);

    mao_c64 zre = C(0.0, 0.0);
    double delsec = 2.0 / (double)nsec1;
    double secmid = -1.0 - delsec / 2.0;
    for (int isec = 0; isec < nsec1; ++isec) {
      secmid += delsec;
      for (int ig = 0; ig < ne; ++ig) {
        double sga = secmid + cg[ig] / (double)nsec1;
        double wga = wg[ig] * lens;
        double gp[3], df[3];
        for (int i = 0; i < 3; ++i) { gp[i] = coords[3*ieg+i] + dpoi[i] * (sga + 1.0) / 2.0; df[i] = gp[i] - x[i]; }
        double ur[3]; double r = normalize3(df, ur);
        if (r < 1e-15) continue;
        double re1 = wavruim * r, re2 = 4.0 * PI * r;
        mao_c64 zg = C(cos(re1) / re2, sin(re1) / re2);
        mao_c64 z1 = C(-1.0 / r, wavruim);
        mao_c64 zgf = cmul(zg, z1);
        mao_c64 zd[3] = {cscale(zgf, ur[0]), cscale(zgf, ur[1]), cscale(zgf, ur[2])};
        mao_c64 zw[3] = {
          csub(cscale(zd[1], dpoo[2]), cscale(zd[2], dpoo[1])),
          csub(cscale(zd[2], dpoo[0]), cscale(zd[0], dpoo[2])),
          csub(cscale(zd[0], dpoo[1]), cscale(zd[1], dpoo[0]))};
        mao_c64 dotn = cadd(cadd(cscale(zw[0], nx[0]), cscale(zw[1], nx[1])), cscale(zw[2], nx[2]));
        zre = cadd(zre, cscale(dotn, wga));
      }
    }
    res->d2g = cadd(res->d2g, zre);

    for (int isec = 0; isec < nsec2; ++isec) {
      double ssub[3], tsub[3], aresub;
      if (nn == 3) { aresub = 1.0 / 24.0 / (double)nsec2; ssub[0] = 1.0 / 3.0; tsub[0] = 1.0 / 3.0; }
      else { aresub = 0.25 / (double)nsec2; ssub[0] = 0.0; tsub[0] = 0.0; }
      if (isec == 0) { ssub[1] = CS[ieg]; ssub[2] = CS[ig2]; tsub[1] = ET[ieg]; tsub[2] = ET[ig2]; }
      else { ssub[1] = CS[ig2]; ssub[2] = CS[ig1]; tsub[1] = ET[ig2]; tsub[2] = ET[ig1]; }

      for (int i = 0; i < ns; ++i) {
        double sga = gc[i];
        for (int j = 0; j < ns; ++j) {
          double tga = gc[j];
          double wei = gw[i] * gw[j];
          double sgg = 0.5 * (1.0 - sga) * ssub[0] + 0.25 * (1.0 + sga) * ((1.0 - tga) * ssub[1] + (1.0 + tga) * ssub[2]);
          double tgg = 0.5 * (1.0 - sga) * tsub[0] + 0.25 * (1.0 + sga) * ((1.0 - tga) * tsub[1] + (1.0 + tga) * tsub[2]);
          double N[4], jac, ny[3], crd[3];
          compute_parameters(coords, nn, sgg, tgg, N, &jac, ny, crd);
          double wga = wei * (1.0 + sga) * aresub * jac;
          double df[3] = {crd[0] - x[0], crd[1] - x[1], crd[2] - x[2]};
          double ur[3]; double r = normalize3(df, ur);
          if (r < 1e-15) continue;
          double re1 = wavruim * r;
          double re2 = wga / (4.0 * PI * r);
          mao_c64 zg = C(cos(re1) * re2, sin(re1) * re2);
          mao_c64 z1 = C(-1.0 / r, wavruim);
          mao_c64 zhb = cmul(zg, z1);
          double re1h = dot3(ur, ny);
          double re2h = -dot3(ur, nx);
          mao_c64 zhh = cscale(zhb, re1h), zht = cscale(zhb, re2h);
          res->g = cadd(res->g, zg);
          res->dg_dn = cadd(res->dg_dn, zhh);
          res->dg_dnx = cadd(res->dg_dnx, zht);
          res->d2g = cadd(res->d2g, cscale(cscale(zg, k2), dot3(nx, ny)));
          if (compute_rhs && bc_type == 0 && bc) {
            mao_c64 zb = C(0.0, 0.0);
            for (int ii = 0; ii < nn; ++ii) if (ii < bc_len) zb = cadd(zb, cscale(bc[ii], N[ii]));
            mao_c64 beta = mao_burton_miller_beta(kwave, harmonic, tau);
            mao_c64 t = cadd(cscale(cscale(zg, 1.0), tau), cmul(zht, beta));
            res->rhs = cadd(res->rhs, cmul(t, zb));
          }
        }
      }
    }
  }
  if (compute_rhs && bc_type == 1 && bc) {
    mao_c64 s = C(0.0, 0.0);
    for (int i = 0; i < bc_len; ++i) s = cadd(s, bc[i]);
    mao_c64 zb = cdivr(s, (double)bc_len);
    mao_c64 beta = mao_burton_miller_beta(kwave, harmonic, tau);
    mao_c64 t = cadd(cscale(cscale(res->dg_dn, 1.0), tau), cmul(res->d2g, beta));
    res->rhs = cmul(cneg(t), zb);
  }
}

void mao_singular_integration(const double* x, const double* nx, const double* coords, int nn,
        double kwave, double harmonic, double tau, const mao_c64* bc, int bc_len, int bc_type, int compute_rhs,
        mao_integration_result* res) {
  /* singular.rs:133-136 + QuadratureParams::for_ka :48-82 */
  double ka = kwave * estimate_element_size(coords, nn);
  int p[4];
  if (ka < 0.3)      { p[0] = 3; p[1] = 4; p[2] = 4;  p[3] = 2; }
  else if (ka < 1.0) { p[0] = 4; p[1] = 5; p[2] = 6;  p[3] = 2; }
  else if (ka < 2.0) { p[0] = 5; p[1] = 6; p[2] = 8;  p[3] = 3; }
  else               { p[0] = 6; p[1] = 7; p[2] = 10; p[3] = 4; }
  mao_singular_integration_with_params(x, nx, coords, nn, kwave, harmonic, tau, bc, bc_len, bc_type, compute_rhs,
                                       p[0], p[1], p[2], p[3], res);
}

/* ======================================================================
 * build_tbem_system_with_beta — tbem.rs:96-345
 * ====================================================================== */
typedef struct {
  int n_elem; const double* nodes; const int* conn; const double* center; const double* normal; const double* area;
  const int* dof; const unsigned char* bc_type; const mao_c64* bc_values; const int* bc_len; const unsigned char* is_eval;
  double k, harmonic, tau; mao_c64 beta; double sign;
  mao_c64* A; mao_c64* rhs; int num_dofs;
  int row_begin, row_end, stride, offset;
  int packed;     /* test convenience: rows [row_begin, row_end) stored as a (row_end - row_begin) x num_dofs strip */
} tbem_job;

static int has_nonzero_bc(const mao_c64* v, int n) {
  for (int i = 0; i < n; ++i) if (cnorm(v[i]) > 1e-15) return 1;
  return 0;
}

static void* tbem_rows(void* arg) {
  tbem_job* J = (tbem_job*)arg;
  const double gamma = 1.0;
  mao_c64 cg = C(gamma, 0.0), ct = C(J->tau, 0.0);
  for (int iel = J->row_begin + J->offset; iel < J->row_end; iel += J->stride) {
    if (J->is_eval && J->is_eval[iel]) continue;
    const double* x = J->center + 3*iel;
    const double* nx = J->normal + 3*iel;
    int sdof = J->dof[iel];
    int bct = J->bc_type[iel] > 1 ? 2 : J->bc_type[iel];
    const mao_c64* bcv = J->bc_values + 4*iel; int bcl = J->bc_len[iel];
    static const mao_c64 zero1 = {0.0, 0.0};
    if (bct == 2) { bcv = &zero1; bcl = 1; }                        /* tbem.rs:239-242 */
    /* add_free_terms tbem.rs:273-304 */
    mao_c64 s = C(0.0, 0.0);
    for (int i = 0; i < bcl; ++i) s = cadd(s, bcv[i]);
    mao_c64 avg = cdivr(s, (double)bcl);
    int srow = J->packed ? iel - J->row_begin : sdof;                /* where this source row is stored */
    mao_c64* Arow = J->A + (size_t)srow * (size_t)J->num_dofs;
    if (bct == 0) {
      Arow[sdof] = csub(Arow[sdof], cscale(cg, 0.5));
      J->rhs[srow] = cadd(J->rhs[srow], cscale(cmul(cmul(avg, J->beta), ct), 0.5));
    } else if (bct == 1) {
      Arow[sdof] = csub(Arow[sdof], cscale(cmul(J->beta, ct), 0.5));
      J->rhs[srow] = cadd(J->rhs[srow], cscale(cmul(avg, ct), 0.5));
    }
    for (int jel = 0; jel < J->n_elem; ++jel) {
      if (J->is_eval && J->is_eval[jel]) continue;
      const int* cn = J->conn + 4*jel;
      int nn = cn[3] < 0 ? 3 : 4;
      double coords[12];
      for (int i = 0; i < nn; ++i) for (int d = 0; d < 3; ++d) coords[3*i+d] = J->nodes[3*cn[i]+d];
      int fdof = J->dof[jel];
      int fbt = J->bc_type[jel] > 1 ? 2 : J->bc_type[jel];
      const mao_c64* fbv = J->bc_values + 4*jel; int fbl = J->bc_len[jel];
      if (fbt == 2) { fbv = &zero1; fbl = 1; }
      int crhs = has_nonzero_bc(fbv, fbl);
      mao_integration_result r;
      if (jel == iel)
        mao_singular_integration(x, nx, coords, nn, J->k, J->harmonic, J->tau, crhs ? fbv : NULL, fbl, fbt, crhs, &r);
      else
        mao_regular_integration(x, nx, coords, nn, J->area[jel], J->k, J->harmonic, J->tau, crhs ? fbv : NULL, fbl, fbt, crhs, &r);
      r.dg_dn = cscale(r.dg_dn, J->sign);                                  /* tbem.rs:203 */
      mao_c64 coeff;                                                        /* assemble_tbem tbem.rs:311-345 */
      if (fbt == 0) coeff = cadd(cmul(cmul(r.dg_dn, cg), ct), cmul(r.d2g, J->beta));
      else if (fbt == 1) coeff = cneg(cadd(cmul(cmul(r.g, cg), ct), cmul(r.dg_dnx, J->beta)));
      else coeff = C(0.0, 0.0);
      Arow[fdof] = cadd(Arow[fdof], coeff);
      if (crhs) J->rhs[srow] = cadd(J->rhs[srow], r.rhs);
    }
  }
  return NULL;
}

int mao_build_tbem_system_with_beta(int n_elem, const double* nodes, const int* conn,
        const double* center, const double* normal, const double* area,
        const int* dof, const unsigned char* bc_type, const mao_c64* bc_values, const int* bc_len,
        const unsigned char* is_eval, double k, double harmonic, double tau, double bre, double bim,
        mao_c64* A, mao_c64* rhs, int num_dofs, int row_begin, int row_end, int nthreads) {
  return mao_build_tbem_rows(n_elem, nodes, conn, center, normal, area, dof, bc_type, bc_values, bc_len, is_eval, k, harmonic, tau,
                             bre, bim, A, rhs, num_dofs, row_begin, row_end, nthreads, 0);
}

/* The same rows; packed != 0 stores source rows [row_begin, row_end) as a (row_end - row_begin) x num_dofs strip (A) and
 * row_end - row_begin right-hand-side entries, so that sampled rows of a 50k-panel system need no N x N buffer. */
int mao_build_tbem_rows(int n_elem, const double* nodes, const int* conn,
        const double* center, const double* normal, const double* area,
        const int* dof, const unsigned char* bc_type, const mao_c64* bc_values, const int* bc_len,
        const unsigned char* is_eval, double k, double harmonic, double tau, double bre, double bim,
        mao_c64* A, mao_c64* rhs, int num_dofs, int row_begin, int row_end, int nthreads, int packed) {
  /* tbem.rs:108-123 sign switch from the first <=100 element centres */
  double avg = 0.0; int nc = n_elem < 100 ? n_elem : 100;
  for (int e = 0; e < nc; ++e) avg += sqrt(dot3(center + 3*e, center + 3*e));
  if (nc > 0) avg /= (double)nc;
  double ka = k * avg;
  double sign = ka < 0.5 ? 1.0 : -1.0;
  /* TbemSystem::new zeroes (tbem.rs:24-30): only the rows in range */
  for (int e = row_begin; e < row_end; ++e) {
    if (is_eval && is_eval[e]) continue;
    int srow = packed ? e - row_begin : dof[e];
    memset(A + (size_t)srow * (size_t)num_dofs, 0, sizeof(mao_c64) * (size_t)num_dofs);
    rhs[srow] = C(0.0, 0.0);
  }
  if (nthreads < 1) nthreads = 1;
  tbem_job* jobs = (tbem_job*)malloc(sizeof(tbem_job) * (size_t)nthreads);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  for (int t = 0; t < nthreads; ++t) {
    tbem_job j = {n_elem, nodes, conn, center, normal, area, dof, bc_type, bc_values, bc_len, is_eval,
                  k, harmonic, tau, {bre, bim}, sign, A, rhs, num_dofs, row_begin, row_end, nthreads, t, packed};
    jobs[t] = j;
    if (nthreads > 1) pthread_create(&th[t], NULL, tbem_rows, &jobs[t]);
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  else tbem_rows(&jobs[0]);
  free(jobs); free(th);
  return 0;
}

/* ======================================================================
 * Incident field — incident.rs:93-342
 * ====================================================================== */
void mao_incident_pressure(int kind, const double* v, mao_c64 amp, int n, const double* pts, double k, mao_c64* out) {
  for (int i = 0; i < n; ++i) {
    const double* p = pts + 3*i;
    if (kind == 0) {
      double kdx = k * (v[0] * p[0] + v[1] * p[1] + v[2] * p[2]);
      out[i] = cmul(amp, C(cos(kdx), sin(kdx)));
    } else {
      double dx = p[0] - v[0], dy = p[1] - v[1], dz = p[2] - v[2];
      double r = sqrt(dx*dx + dy*dy + dz*dz);
      out[i] = C(0.0, 0.0);
      if (r > 1e-10) {
        double kr = k * r;
        mao_c64 g = cdivr(C(cos(kr), sin(kr)), 4.0 * PI * r);
        out[i] = cmul(amp, g);
      }
    }
  }
}

void mao_incident_normal_derivative(int kind, const double* v, mao_c64 amp, int n, const double* pts,
                                    const double* nrm, double k, mao_c64* out) {
  for (int i = 0; i < n; ++i) {
    const double* p = pts + 3*i; const double* nn = nrm + 3*i;
    if (kind == 0) {
      double kdx = k * (v[0] * p[0] + v[1] * p[1] + v[2] * p[2]);
      double kdn = k * (v[0] * nn[0] + v[1] * nn[1] + v[2] * nn[2]);
      mao_c64 pw = cmul(amp, C(cos(kdx), sin(kdx)));
      out[i] = cmul(C(0.0, kdn), pw);
    } else {
      double dx = p[0] - v[0], dy = p[1] - v[1], dz = p[2] - v[2];
      double r = sqrt(dx*dx + dy*dy + dz*dz);
      out[i] = C(0.0, 0.0);
      if (r > 1e-10) {
        double kr = k * r;
        mao_c64 g = cdivr(C(cos(kr), sin(kr)), 4.0 * PI * r);
        mao_c64 dgdr = cmul(csub(C(0.0, k), C(1.0 / r, 0.0)), g);
        double drdn = (dx * nn[0] + dy * nn[1] + dz * nn[2]) / r;
        out[i] = cscale(cmul(amp, dgdr), drdn);
      }
    }
  }
}

void mao_compute_rhs_with_beta(int kind, const double* v, mao_c64 amp, int n, const double* centers,
                               const double* normals, double k, double tau, mao_c64 beta, mao_c64* rhs) {
  /* incident.rs:317-342 */
  mao_c64* p = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  mao_c64* d = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  mao_incident_pressure(kind, v, amp, n, centers, k, p);
  mao_incident_normal_derivative(kind, v, amp, n, centers, normals, k, d);
  mao_c64 g = C(1.0, 0.0), t = C(tau, 0.0);
  for (int i = 0; i < n; ++i) rhs[i] = cneg(cadd(cmul(g, p[i]), cmul(cmul(beta, t), d[i])));
  free(p); free(d);
}

/* ======================================================================
 * Field post-processing — postprocess/pressure.rs:81-258
 * ====================================================================== */
void mao_compute_scattered_field(int n_eval, const double* ep, int n_elem, const double* nodes,
        const int* conn, const unsigned char* is_eval, const mao_c64* ps, const mao_c64* vs,
        double k, double harmonic, mao_c64* out) {
  double wavruim = k * harmonic;
  double q[39]; int nq = mao_triangle_quadrature(3, q);
  for (int i = 0; i < n_eval; ++i) {
    const double* x = ep + 3*i;
    mao_c64 acc = C(0.0, 0.0);
    int j = 0;                                   /* index into the boundary-only list (pressure.rs:97-113) */
    for (int e = 0; e < n_elem; ++e) {
      if (is_eval && is_eval[e]) continue;
      mao_c64 p_surf = ps[j], v_surf = vs ? vs[j] : C(0.0, 0.0);
      ++j;
      const int* cn = conn + 4*e;
      double co[9];
      for (int a = 0; a < 3; ++a) for (int d = 0; d < 3; ++d) co[3*a+d] = nodes[3*cn[a]+d];  /* first 3 nodes only (:198) */
      mao_c64 r_e = C(0.0, 0.0);
      for (int iq = 0; iq < nq; ++iq) {
        double xi = q[3*iq], eta = q[3*iq+1], w = q[3*iq+2];
        double N[3] = {1.0 - xi - eta, xi, eta};
        static const double ds[3] = {-1.0, 1.0, 0.0}, dt[3] = {-1.0, 0.0, 1.0};
        double crd[3] = {0,0,0}, dxs[3] = {0,0,0}, dxt[3] = {0,0,0};
        for (int a = 0; a < 3; ++a) for (int d = 0; d < 3; ++d) {
          crd[d] += N[a] * co[3*a+d]; dxs[d] += ds[a] * co[3*a+d]; dxt[d] += dt[a] * co[3*a+d];
        }
        double nr[3] = {dxs[1]*dxt[2] - dxs[2]*dxt[1], dxs[2]*dxt[0] - dxs[0]*dxt[2], dxs[0]*dxt[1] - dxs[1]*dxt[0]};
        double jac = sqrt(dot3(nr, nr));
        if (jac < 1e-15) continue;
        double en[3] = {nr[0] / jac, nr[1] / jac, nr[2] / jac};
        double rv[3] = {crd[0] - x[0], crd[1] - x[1], crd[2] - x[2]};
        double r = sqrt(dot3(rv, rv));
        if (r < 1e-15) continue;
        double vj = jac * w;
        double kr = wavruim * r, re1 = 4.0 * PI * r;
        mao_c64 zg = C(cos(kr) / re1, sin(kr) / re1);
        mao_c64 z1 = C(-1.0 / r, wavruim);
        mao_c64 zgikr = cmul(zg, z1);
        double drdn = dot3(rv, en) / r;
        mao_c64 zd = cscale(zgikr, drdn);
        r_e = cadd(r_e, cscale(cmul(p_surf, zd), vj));
        if (cnorm(v_surf) > 1e-15) r_e = csub(r_e, cscale(cmul(v_surf, zg), vj));
      }
      acc = cadd(acc, r_e);
    }
    out[i] = acc;
  }
}

/* ======================================================================
 * Room-acoustics point-collocation assembly — room_acoustics/solver.rs:28-35,448-493
 * (element centre/normal/area are supplied by the caller, computed as :38-122)
 * ====================================================================== */
typedef struct { int n; const double* c; const double* nr; const double* ar; double k; mao_c64* A; int stride, offset; } room_job;
static void* room_rows(void* arg) {
  room_job* J = (room_job*)arg;
  int n = J->n; double k = J->k;
  for (int i = J->offset; i < n; i += J->stride) {
    const double* ci = J->c + 3*i; const double* ni = J->nr + 3*i;
    for (int j = 0; j < n; ++j) {
      const double* cj = J->c + 3*j;
      double ddx = ci[0] - cj[0], ddy = ci[1] - cj[1], ddz = ci[2] - cj[2];
      double r = sqrt(ddx*ddx + ddy*ddy + ddz*ddz);
      mao_c64 v;
      if (i == j) v = cscale(C(0.0, -k / (2.0 * PI)), J->ar[j]);
      else {
        double cosang = (ddx * ni[0] + ddy * ni[1] + ddz * ni[2]) / r;
        if (r < 1e-10) v = C(0.0, 0.0);
        else {
          mao_c64 ikr = C(0.0, k * r);
          mao_c64 e = C(cos(ikr.im), sin(ikr.im));            /* exp(0 + i kr) */
          mao_c64 f = cdivr(cmul(C(ikr.re - 1.0, ikr.im), e), 4.0 * PI * r * r);
          v = cscale(cscale(f, cosang), J->ar[j]);
        }
      }
      J->A[(size_t)i * (size_t)n + (size_t)j] = v;
    }
  }
  return NULL;
}
void mao_room_build_matrix(int n, const double* center, const double* normal, const double* area, double k, mao_c64* A, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  room_job* jobs = (room_job*)malloc(sizeof(room_job) * (size_t)nthreads);
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  for (int t = 0; t < nthreads; ++t) {
    room_job j = {n, center, normal, area, k, A, nthreads, t}; jobs[t] = j;
    if (nthreads > 1) pthread_create(&th[t], NULL, room_rows, &jobs[t]);
  }
  if (nthreads > 1) for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  else room_rows(&jobs[0]);
  free(jobs); free(th);
}

/* ======================================================================
 * Rest of the room-acoustics path — room_acoustics/solver.rs:38-122 (element data), 500-597 (adaptive assembly),
 * 600-611 (characteristic length), 638-678 (incident derivative), 687-748 (field pressure)
 * ====================================================================== */
/* element_center_and_normal :38-67, element_area :70-122, element_characteristic_length :600-611; conn rows hold 3 or 4
 * node ids (-1 in the fourth slot for triangles) */
void mao_room_element_data(int n_elem, const double* nodes, const int* conn, double* center, double* normal, double* area, double* charlen) {
  for (int e = 0; e < n_elem; ++e) {
    const int* cn = conn + 4*e;
    int nn = cn[3] < 0 ? 3 : 4;
    const double* p[4];
    for (int a = 0; a < nn; ++a) p[a] = nodes + 3*cn[a];
    for (int d = 0; d < 3; ++d) {
      double s = 0.0;
      for (int a = 0; a < nn; ++a) s += p[a][d];
      center[3*e+d] = s / (double)nn;
    }
    double v1[3], v2[3];
    for (int d = 0; d < 3; ++d) { v1[d] = p[1][d] - p[0][d]; v2[d] = p[2][d] - p[0][d]; }
    double nx = v1[1]*v2[2] - v1[2]*v2[1], ny = v1[2]*v2[0] - v1[0]*v2[2], nz = v1[0]*v2[1] - v1[1]*v2[0];
    double nrm = sqrt(nx*nx + ny*ny + nz*nz);
    normal[3*e] = nx / nrm; normal[3*e+1] = ny / nrm; normal[3*e+2] = nz / nrm;
    double a1 = 0.5 * sqrt(nx*nx + ny*ny + nz*nz);
    if (nn == 4) {
      double v3[3];
      for (int d = 0; d < 3; ++d) v3[d] = p[3][d] - p[0][d];
      double cx = v2[1]*v3[2] - v2[2]*v3[1], cy = v2[2]*v3[0] - v2[0]*v3[2], cz = v2[0]*v3[1] - v2[1]*v3[0];
      a1 += 0.5 * sqrt(cx*cx + cy*cy + cz*cz);
    }
    area[e] = a1;
    double d01 = 0.0, d12 = 0.0, d20 = 0.0;
    for (int d = 0; d < 3; ++d) {
      d01 += (p[0][d] - p[1][d]) * (p[0][d] - p[1][d]); d12 += (p[1][d] - p[2][d]) * (p[1][d] - p[2][d]); d20 += (p[2][d] - p[0][d]) * (p[2][d] - p[0][d]);
    }
    charlen[e] = (sqrt(d01) + sqrt(d12) + sqrt(d20)) / 3.0;
  }
}

static mao_c64 room_dgdn(double r, double k, double cosang) {      /* greens_function_derivative :28-35 */
  if (r < 1e-10) return C(0.0, 0.0);
  mao_c64 e = C(cos(k * r), sin(k * r));
  mao_c64 f = cdivr(cmul(C(-1.0, k * r), e), 4.0 * PI * r * r);
  return cscale(f, cosang);
}

/* build_bem_matrix_adaptive :500-597. Near pairs (r < 2 (l_i + l_j) or i == j) take dg_dn_integral of
 * singular_integration_with_params on the FIRST THREE nodes of element j (ElementType::Tri3 whatever the element is, :556),
 * with QuadratureParams::for_ka(k l_j); the rest is the point collocation of build_bem_matrix_parallel. */
void mao_room_build_matrix_adaptive(int n_elem, const double* nodes, const int* conn, double k, int use_adaptive, mao_c64* A) {
  double* c = (double*)malloc(sizeof(double) * 3 * (size_t)n_elem); double* nr = (double*)malloc(sizeof(double) * 3 * (size_t)n_elem);
  double* ar = (double*)malloc(sizeof(double) * (size_t)n_elem); double* cl = (double*)malloc(sizeof(double) * (size_t)n_elem);
  mao_room_element_data(n_elem, nodes, conn, c, nr, ar, cl);
  for (int i = 0; i < n_elem; ++i)
    for (int j = 0; j < n_elem; ++j) {
      double dx = c[3*i] - c[3*j], dy = c[3*i+1] - c[3*j+1], dz = c[3*i+2] - c[3*j+2];
      double r = sqrt(dx*dx + dy*dy + dz*dz);
      int is_near = r < 2.0 * (cl[i] + cl[j]) || i == j;
      mao_c64 v;
      if (use_adaptive && is_near) {
        double coords[9];
        for (int a = 0; a < 3; ++a) for (int d = 0; d < 3; ++d) coords[3*a+d] = nodes[3*conn[4*j+a]+d];
        double ka = k * cl[j];
        int p[4];
        if (ka < 0.3)      { p[0] = 3; p[1] = 4; p[2] = 4;  p[3] = 2; }
        else if (ka < 1.0) { p[0] = 4; p[1] = 5; p[2] = 6;  p[3] = 2; }
        else if (ka < 2.0) { p[0] = 5; p[1] = 6; p[2] = 8;  p[3] = 3; }
        else               { p[0] = 6; p[1] = 7; p[2] = 10; p[3] = 4; }
        mao_integration_result res;
        mao_singular_integration_with_params(c + 3*i, nr + 3*i, coords, 3, k, 1.0, -1.0, NULL, 0, 0, 0, p[0], p[1], p[2], p[3], &res);
        v = res.dg_dn;
      } else if (i == j) v = cscale(C(0.0, -k / (2.0 * PI)), ar[j]);
      else {
        double cosang = (dx * nr[3*i] + dy * nr[3*i+1] + dz * nr[3*i+2]) / r;
        v = cscale(room_dgdn(r, k, cosang), ar[j]);
      }
      A[(size_t)i * (size_t)n_elem + (size_t)j] = v;
    }
  free(c); free(nr); free(ar); free(cl);
}

/* calculate_incident_field_derivative_parallel :638-678: rhs_i = - sum_s dG/dn(r_is) amp_is; amp is [nsrc] (the same
 * towards every element: an omnidirectional source) or [nsrc][n] when per_point != 0 (Source::amplitude_towards evaluated
 * by the caller, math-xem-common/src/source.rs:203-219) */
void mao_room_incident_derivative(int n, const double* center, const double* normal, int nsrc, const double* src_pos, const double* amp,
                                  int per_point, double k, mao_c64* out) {
  for (int i = 0; i < n; ++i) {
    mao_c64 s = C(0.0, 0.0);
    for (int q = 0; q < nsrc; ++q) {
      double dx = center[3*i] - src_pos[3*q], dy = center[3*i+1] - src_pos[3*q+1], dz = center[3*i+2] - src_pos[3*q+2];
      double r = sqrt(dx*dx + dy*dy + dz*dz);
      if (r < 1e-10) continue;
      double cosang = (dx * normal[3*i] + dy * normal[3*i+1] + dz * normal[3*i+2]) / r;
      double a = per_point ? amp[(size_t)q * (size_t)n + (size_t)i] : amp[q];
      s = cadd(s, cscale(room_dgdn(r, k, cosang), a));
    }
    out[i] = cneg(s);
  }
}

/* calculate_field_pressure_bem_parallel :687-748: p(x) = sum_s G(|x - s|) amp + sum_j dG/dn_j(x - c_j) p_j A_j */
void mao_room_field_pressure(int n, const double* center, const double* normal, const double* area, const mao_c64* surface_pressure,
                             int nsrc, const double* src_pos, const double* amp, int per_point, int npts, const double* pts, double k, mao_c64* out) {
  for (int m = 0; m < npts; ++m) {
    const double* x = pts + 3*m;
    mao_c64 p = C(0.0, 0.0);
    for (int q = 0; q < nsrc; ++q) {
      double dx = x[0] - src_pos[3*q], dy = x[1] - src_pos[3*q+1], dz = x[2] - src_pos[3*q+2];
      double r = sqrt(dx*dx + dy*dy + dz*dz);
      if (r < 1e-10) continue;
      double a = per_point ? amp[(size_t)q * (size_t)npts + (size_t)m] : amp[q];
      mao_c64 g = cdivr(C(cos(k * r), sin(k * r)), 4.0 * PI * r);        /* greens_function_3d :18-24 */
      p = cadd(p, cscale(g, a));
    }
    for (int j = 0; j < n; ++j) {
      double dx = x[0] - center[3*j], dy = x[1] - center[3*j+1], dz = x[2] - center[3*j+2];
      double r = sqrt(dx*dx + dy*dy + dz*dz);
      if (r < 1e-10) continue;
      double cosang = (dx * normal[3*j] + dy * normal[3*j+1] + dz * normal[3*j+2]) / r;
      p = cadd(p, cscale(cmul(room_dgdn(r, k, cosang), surface_pressure[j]), area[j]));
    }
    out[m] = p;
  }
}
