/* oracle_fmm.c — CPU restatement of the reference's single-level fast multipole operator. TEST INFRASTRUCTURE ONLY.
 *
 * Follows math-bem/src/core/assembly/slfmm.rs: build_slfmm_system (:417-470), build_near_field / compute_near_block (:473-612),
 * build_t_matrices (:615-656), build_d_matrices (:663-721), build_s_matrices (:724-765), SlfmmSystem::matvec (:150-257) and
 * matvec_transpose (:262-376); unit_sphere_quadrature (integration/gauss.rs:110-130); spherical_hankel_first_kind
 * (math-wave/src/special/spherical.rs:165-246). The reference's far field is the simplified model its source states
 * ("D[p,p] = h_0(kr) * ik ... all entries are the same in this simplified model", slfmm.rs:707-710): it is restated as it is.
 */
#include "ma_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static mao_c64 C(double re, double im) { mao_c64 z = {re, im}; return z; }
static mao_c64 cadd(mao_c64 a, mao_c64 b) { return C(a.re + b.re, a.im + b.im); }
static mao_c64 cmul(mao_c64 a, mao_c64 b) { return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static mao_c64 cscale(mao_c64 a, double s) { return C(a.re * s, a.im * s); }

/* spherical.rs:165-246 (order >= 2, x > 0) */
void mao_spherical_hankel_first_kind(int order, double x, double harmonic, mao_c64* result) {
  const double cos_x = cos(x), sin_x = sin(x);
  double* y_n = (double*)malloc(sizeof(double) * (size_t)order);
  double* gg = (double*)calloc((size_t)order, sizeof(double));
  double* dg = (double*)calloc((size_t)order, sizeof(double));
  y_n[0] = -cos_x / x;
  y_n[1] = -(cos_x / x + sin_x) / x;
  for (int n = 2; n < order; ++n) y_n[n] = (double)(2 * n - 1) / x * y_n[n - 1] - y_n[n - 2];
  const double nu = (double)(order - 1);
  double di = (2.0 * (nu + 1.0) + 1.0) / x, cj = di, dj = 0.0, err = 1.0;
  int j = 1;
  while (err > 1e-9) {
    const double aj = -1.0, bj = (2.0 * (nu + (double)j + 1.0) + 1.0) / x;
    dj = bj + aj * dj; if (dj == 0.0) dj = 1e-30; dj = 1.0 / dj;
    cj = bj + aj / cj; if (cj == 0.0) cj = 1e-30;
    di = di * cj * dj;
    err = fabs(cj * dj - 1.0);
    j += 1;
    if (j > 1000) break;
  }
  const double gnu = nu / x - 1.0 / di;
  gg[order - 1] = 1.0; dg[order - 1] = gnu;
  for (int i = order - 2; i >= 0; --i) {
    const double d = (double)i;
    gg[i] = (d + 2.0) / x * gg[i + 1] + dg[i + 1];
    dg[i] = d / x * gg[i] - gg[i + 1];
  }
  const double dp = fabs(gg[0]) > 1e-5 ? sin_x / x / gg[0] : (cos_x - sin_x / x) / x / dg[0];
  for (int n = 0; n < order; ++n) result[n] = C(dp * gg[n], harmonic * y_n[n]);
  free(y_n); free(gg); free(dg);
}

/* gauss.rs:110-130; returns the number of points actually produced (n_theta must be a tabulated Gauss-Legendre order) */
int mao_unit_sphere_quadrature(int n_theta, int n_phi, double* coords /* [n][3] */, double* weights) {
  double ct[32], wt[32];
  const int nt = mao_gauss_legendre(n_theta, ct, wt);
  const double pi = 3.14159265358979323846;
  const double dphi = 2.0 * pi / (double)n_phi;
  int q = 0;
  for (int i = 0; i < nt; ++i) {
    const double st = sqrt(1.0 - ct[i] * ct[i]);
    for (int j = 0; j < n_phi; ++j, ++q) {
      const double phi = dphi * (double)j;
      coords[3 * q] = st * cos(phi); coords[3 * q + 1] = st * sin(phi); coords[3 * q + 2] = ct[i];
      weights[q] = wt[i] * dphi / (4.0 * pi);
    }
  }
  return q;
}

/* compute_near_block (mlfmm.rs:647-710; the same as slfmm.rs:536-612): coefficient dg_dn gamma tau + d2g beta with
 * beta = physics.burton_miller_beta(), singular integration only for the same element of a self block, NO free term. out: ns x nf */
void mao_fmm_near_block(const double* nodes, const int* conn, const double* center, const double* normal, const double* area,
                        int ns, const int* src_idx, int nf, const int* fld_idx, int is_self, double k, double harmonic, double tau, mao_c64* out) {
  const double gamma = 1.0;
  const mao_c64 beta = mao_burton_miller_beta(k, harmonic, tau);
  for (int i = 0; i < ns; ++i) {
    const int se = src_idx[i];
    for (int j = 0; j < nf; ++j) {
      const int fe = fld_idx[j];
      const int* cn = conn + 4 * fe; const int nn = cn[3] < 0 ? 3 : 4;
      double coords[12];
      for (int a = 0; a < nn; ++a) for (int d = 0; d < 3; ++d) coords[3 * a + d] = nodes[3 * cn[a] + d];
      mao_integration_result r;
      if (is_self && se == fe) mao_singular_integration(center + 3 * se, normal + 3 * se, coords, nn, k, harmonic, tau, NULL, 0, 0, 0, &r);
      else mao_regular_integration(center + 3 * se, normal + 3 * se, coords, nn, area[fe], k, harmonic, tau, NULL, 0, 0, 0, &r);
      out[(size_t)i * nf + j] = cadd(cscale(cscale(r.dg_dn, gamma), tau), cmul(r.d2g, beta));
    }
  }
}

struct mao_slfmm {
  int num_dofs, nc, P;
  int* eptr; int* eidx;            /* cluster -> element indices (cluster_dof_indices: dof = dof_addresses[0]) */
  int* dofs;                       /* dof of every listed element, same layout as eidx */
  int nblocks; int* bsrc; int* bfld; size_t* boff; mao_c64* bval;     /* near blocks, row-major n_src x n_fld */
  mao_c64** T; mao_c64** S;        /* per cluster: T [P][n_c], S [n_c][P] */
  int nfar; int* dsrc; int* dfld; mao_c64* dval;                       /* D entries: diagonal = dval (constant over p) */
};

mao_slfmm* mao_slfmm_build(int n_elem, const double* nodes, const int* conn, const double* center, const double* normal, const double* area,
                           const int* dof, const unsigned char* bc_type, int n_clusters, const double* ccenter, const int* elem_ptr, const int* elem_idx,
                           const int* near_ptr, const int* near_idx, const int* far_ptr, const int* far_idx,
                           double k, double harmonic, double tau, int n_theta, int n_phi, int n_terms) {
  mao_slfmm* S = (mao_slfmm*)calloc(1, sizeof(mao_slfmm));
  S->num_dofs = n_elem; S->nc = n_clusters; S->P = n_theta * n_phi;
  const int ne_listed = elem_ptr[n_clusters];
  S->eptr = (int*)malloc(sizeof(int) * (size_t)(n_clusters + 1)); memcpy(S->eptr, elem_ptr, sizeof(int) * (size_t)(n_clusters + 1));
  S->eidx = (int*)malloc(sizeof(int) * (size_t)(ne_listed > 0 ? ne_listed : 1)); memcpy(S->eidx, elem_idx, sizeof(int) * (size_t)ne_listed);
  S->dofs = (int*)malloc(sizeof(int) * (size_t)(ne_listed > 0 ? ne_listed : 1));
  for (int q = 0; q < ne_listed; ++q) S->dofs[q] = dof[elem_idx[q]];
  double* sc = (double*)malloc(sizeof(double) * 3 * (size_t)S->P * 4);
  double* sw = (double*)malloc(sizeof(double) * (size_t)S->P * 4);
  const int P = mao_unit_sphere_quadrature(n_theta, n_phi, sc, sw);
  S->P = P;
  /* ---- near field: (i, i, self) and (i, j > i) over near_clusters (slfmm.rs:484-497) */
  const double gamma = 1.0;
  const mao_c64 beta = mao_burton_miller_beta(k, harmonic, tau);                  /* :482 physics.burton_miller_beta() */
  int nb = 0;
  for (int i = 0; i < n_clusters; ++i) { nb += 1; for (int q = near_ptr[i]; q < near_ptr[i + 1]; ++q) if (near_idx[q] > i) nb += 1; }
  S->nblocks = nb; S->bsrc = (int*)malloc(sizeof(int) * (size_t)nb); S->bfld = (int*)malloc(sizeof(int) * (size_t)nb); S->boff = (size_t*)malloc(sizeof(size_t) * (size_t)(nb + 1));
  size_t tot = 0; nb = 0;
  for (int i = 0; i < n_clusters; ++i) {
    S->bsrc[nb] = i; S->bfld[nb] = i; S->boff[nb] = tot; tot += (size_t)(elem_ptr[i + 1] - elem_ptr[i]) * (size_t)(elem_ptr[i + 1] - elem_ptr[i]); nb += 1;
    for (int q = near_ptr[i]; q < near_ptr[i + 1]; ++q) {
      const int j = near_idx[q];
      if (j > i) { S->bsrc[nb] = i; S->bfld[nb] = j; S->boff[nb] = tot; tot += (size_t)(elem_ptr[i + 1] - elem_ptr[i]) * (size_t)(elem_ptr[j + 1] - elem_ptr[j]); nb += 1; }
    }
  }
  S->boff[nb] = tot;
  S->bval = (mao_c64*)calloc(tot > 0 ? tot : 1, sizeof(mao_c64));
  for (int b = 0; b < S->nblocks; ++b) {
    const int ci = S->bsrc[b], cj = S->bfld[b];
    const int ns = elem_ptr[ci + 1] - elem_ptr[ci], nf = elem_ptr[cj + 1] - elem_ptr[cj];
    const int is_self = ci == cj;
    mao_c64* B = S->bval + S->boff[b];
    for (int i = 0; i < ns; ++i) {
      const int se = elem_idx[elem_ptr[ci] + i];
      for (int j = 0; j < nf; ++j) {
        const int fe = elem_idx[elem_ptr[cj] + j];
        const int* cn = conn + 4 * fe; const int nn = cn[3] < 0 ? 3 : 4;
        double coords[12];
        for (int a = 0; a < nn; ++a) for (int d = 0; d < 3; ++d) coords[3 * a + d] = nodes[3 * cn[a] + d];
        mao_integration_result r;
        if (is_self && se == fe) mao_singular_integration(center + 3 * se, normal + 3 * se, coords, nn, k, harmonic, tau, NULL, 0, 0, 0, &r);   /* :555-566 */
        else mao_regular_integration(center + 3 * se, normal + 3 * se, coords, nn, area[fe], k, harmonic, tau, NULL, 0, 0, 0, &r);              /* :567-580 */
        B[(size_t)i * nf + j] = cadd(cscale(cscale(r.dg_dn, gamma), tau), cmul(r.d2g, beta));                                                  /* :583-584 */
      }
    }
    if (is_self)                                                                   /* free terms on the diagonal, :514-533 */
      for (int i = 0; i < ns; ++i) {
        const int e = elem_idx[elem_ptr[ci] + i];
        if (bc_type[e] == 0) B[(size_t)i * nf + i] = cadd(B[(size_t)i * nf + i], C(gamma * 0.5, 0.0));
        else if (bc_type[e] == 1) B[(size_t)i * nf + i] = cadd(B[(size_t)i * nf + i], cscale(cscale(beta, tau), 0.5));
      }
  }
  /* ---- T and S matrices (:615-656, :724-765) */
  S->T = (mao_c64**)calloc((size_t)n_clusters, sizeof(mao_c64*)); S->S = (mao_c64**)calloc((size_t)n_clusters, sizeof(mao_c64*));
  for (int c = 0; c < n_clusters; ++c) {
    const int n = elem_ptr[c + 1] - elem_ptr[c];
    S->T[c] = (mao_c64*)calloc((size_t)P * (size_t)(n > 0 ? n : 1), sizeof(mao_c64));
    S->S[c] = (mao_c64*)calloc((size_t)P * (size_t)(n > 0 ? n : 1), sizeof(mao_c64));
    for (int j = 0; j < n; ++j) {
      const int e = elem_idx[elem_ptr[c] + j];
      const double diff[3] = {center[3 * e] - ccenter[3 * c], center[3 * e + 1] - ccenter[3 * c + 1], center[3 * e + 2] - ccenter[3 * c + 2]};
      for (int p = 0; p < P; ++p) {
        const double sd = sc[3 * p] * diff[0] + sc[3 * p + 1] * diff[1] + sc[3 * p + 2] * diff[2];
        S->T[c][(size_t)p * n + j] = cscale(C(cos(k * sd), -sin(k * sd)), sw[p]);
        S->S[c][(size_t)j * P + p] = cscale(C(cos(k * sd), sin(k * sd)), sw[p]);
      }
    }
  }
  /* ---- D entries (:663-721): one per (i, j in far_clusters[i]); diagonal value h_0(k r) * i k */
  S->nfar = far_ptr[n_clusters];
  S->dsrc = (int*)malloc(sizeof(int) * (size_t)(S->nfar > 0 ? S->nfar : 1)); S->dfld = (int*)malloc(sizeof(int) * (size_t)(S->nfar > 0 ? S->nfar : 1));
  S->dval = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)(S->nfar > 0 ? S->nfar : 1));
  const int order = n_terms > 2 ? n_terms : 2;
  mao_c64* h = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)order);
  int q = 0;
  for (int i = 0; i < n_clusters; ++i)
    for (int t = far_ptr[i]; t < far_ptr[i + 1]; ++t, ++q) {
      const int j = far_idx[t];
      const double d[3] = {ccenter[3 * i] - ccenter[3 * j], ccenter[3 * i + 1] - ccenter[3 * j + 1], ccenter[3 * i + 2] - ccenter[3 * j + 2]};
      const double r = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      mao_spherical_hankel_first_kind(order, k * r, 1.0, h);
      S->dsrc[q] = i; S->dfld[q] = j; S->dval[q] = cmul(h[0], C(0.0, k));
    }
  free(h); free(sc); free(sw);
  return S;
}

void mao_slfmm_free(mao_slfmm* S) {
  if (!S) return;
  for (int c = 0; c < S->nc; ++c) { free(S->T[c]); free(S->S[c]); }
  free(S->T); free(S->S); free(S->eptr); free(S->eidx); free(S->dofs); free(S->bsrc); free(S->bfld); free(S->boff); free(S->bval);
  free(S->dsrc); free(S->dfld); free(S->dval); free(S);
}

/* matvec (:150-257) / matvec_transpose (:262-376) */
void mao_slfmm_matvec(const mao_slfmm* S, int transpose, const mao_c64* x, mao_c64* y) {
  const int P = S->P, nc = S->nc;
  memset(y, 0, sizeof(mao_c64) * (size_t)S->num_dofs);
  for (int b = 0; b < S->nblocks; ++b) {
    const int ci = S->bsrc[b], cj = S->bfld[b];
    const int ns = S->eptr[ci + 1] - S->eptr[ci], nf = S->eptr[cj + 1] - S->eptr[cj];
    const int* sd = S->dofs + S->eptr[ci]; const int* fd = S->dofs + S->eptr[cj];
    const mao_c64* B = S->bval + S->boff[b];
    if (!transpose || ci != cj) {                       /* y_src += B x_fld (forward: every block; transpose: the symmetric half) */
      for (int i = 0; i < ns; ++i) { mao_c64 s = C(0, 0); for (int j = 0; j < nf; ++j) s = cadd(s, cmul(B[(size_t)i * nf + j], x[fd[j]])); y[sd[i]] = cadd(y[sd[i]], s); }
    }
    if (transpose || ci != cj) {                        /* y_fld += B^T x_src */
      for (int j = 0; j < nf; ++j) { mao_c64 s = C(0, 0); for (int i = 0; i < ns; ++i) s = cadd(s, cmul(B[(size_t)i * nf + j], x[sd[i]])); y[fd[j]] = cadd(y[fd[j]], s); }
    }
  }
  mao_c64* up = (mao_c64*)calloc((size_t)nc * (size_t)P, sizeof(mao_c64));
  mao_c64* tr = (mao_c64*)calloc((size_t)nc * (size_t)P, sizeof(mao_c64));
  for (int c = 0; c < nc; ++c) {                        /* multipoles = T x (forward) / locals = S^T x (transpose) */
    const int n = S->eptr[c + 1] - S->eptr[c]; const int* d = S->dofs + S->eptr[c];
    for (int p = 0; p < P; ++p) {
      mao_c64 s = C(0, 0);
      for (int j = 0; j < n; ++j) s = cadd(s, cmul(transpose ? S->S[c][(size_t)j * P + p] : S->T[c][(size_t)p * n + j], x[d[j]]));
      up[(size_t)c * P + p] = s;
    }
  }
  for (int q = 0; q < S->nfar; ++q) {                   /* forward: source -> field; transpose: field -> source */
    const int from = transpose ? S->dfld[q] : S->dsrc[q], to = transpose ? S->dsrc[q] : S->dfld[q];
    for (int p = 0; p < P; ++p) tr[(size_t)to * P + p] = cadd(tr[(size_t)to * P + p], cmul(S->dval[q], up[(size_t)from * P + p]));
  }
  for (int c = 0; c < nc; ++c) {                        /* y += S locals (forward) / T^T multipoles (transpose) */
    const int n = S->eptr[c + 1] - S->eptr[c]; const int* d = S->dofs + S->eptr[c];
    for (int j = 0; j < n; ++j) {
      mao_c64 s = C(0, 0);
      for (int p = 0; p < P; ++p) s = cadd(s, cmul(transpose ? S->T[c][(size_t)p * n + j] : S->S[c][(size_t)j * P + p], tr[(size_t)c * P + p]));
      y[d[j]] = cadd(y[d[j]], s);
    }
  }
  free(up); free(tr);
}

/* extract_near_field_matrix (:104-132) */
void mao_slfmm_near_matrix(const mao_slfmm* S, mao_c64* A) {
  const size_t n = (size_t)S->num_dofs;
  memset(A, 0, sizeof(mao_c64) * n * n);
  for (int b = 0; b < S->nblocks; ++b) {
    const int ci = S->bsrc[b], cj = S->bfld[b];
    const int ns = S->eptr[ci + 1] - S->eptr[ci], nf = S->eptr[cj + 1] - S->eptr[cj];
    const int* sd = S->dofs + S->eptr[ci]; const int* fd = S->dofs + S->eptr[cj];
    const mao_c64* B = S->bval + S->boff[b];
    for (int i = 0; i < ns; ++i) for (int j = 0; j < nf; ++j) {
      A[(size_t)sd[i] * n + fd[j]] = cadd(A[(size_t)sd[i] * n + fd[j]], B[(size_t)i * nf + j]);
      if (ci != cj) A[(size_t)fd[j] * n + sd[i]] = cadd(A[(size_t)fd[j] * n + sd[i]], B[(size_t)i * nf + j]);
    }
  }
}
