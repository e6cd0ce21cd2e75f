/* oracle_solvers.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restatement of the math-solvers / math-fem pieces on the hot path:
 * dense solve (lu.rs), CSR SpMV (csr.rs), AMG smoothers (amg.rs), the
 * geometric-MG smoothers on COO triplets (math-fem smoother.rs), GMRES(m)
 * (gmres.rs) and the per-frequency value update K - k^2 M (assembler.rs).
 */
#include "ma_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

static inline mao_c64 C(double re, double im) { mao_c64 z = {re, im}; return z; }
static inline mao_c64 cadd(mao_c64 a, mao_c64 b) { return C(a.re + b.re, a.im + b.im); }
static inline mao_c64 csub(mao_c64 a, mao_c64 b) { return C(a.re - b.re, a.im - b.im); }
static inline mao_c64 cmul(mao_c64 a, mao_c64 b) { return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline mao_c64 cscale(mao_c64 a, double s) { return C(a.re * s, a.im * s); }
static inline mao_c64 cconj(mao_c64 a) { return C(a.re, -a.im); }
static inline double cnorm_sqr(mao_c64 a) { return a.re * a.re + a.im * a.im; }
static inline double cnorm(mao_c64 a) { return hypot(a.re, a.im); }
/* num_complex Complex::inv(): conj / norm_sqr */
static inline mao_c64 cinv(mao_c64 a) { double ns = cnorm_sqr(a); return C(a.re / ns, -a.im / ns); }
/* num_complex Complex / Complex */
static inline mao_c64 cdiv(mao_c64 a, mao_c64 b) {
  double ns = cnorm_sqr(b);
  return C((a.re * b.re + a.im * b.im) / ns, (a.im * b.re - a.re * b.im) / ns);
}
static inline double cabs1(mao_c64 a) { return fabs(a.re) + fabs(a.im); }

/* ======================================================================
 * Dense solve. Reference: math-solvers/src/direct/lu.rs:142-146 hands the
 * C-order Array2 to ndarray-linalg -> LAPACK zgetrf on the column-major view
 * (= A^T), then zgetrs with trans='T'. Restated here in terms of the
 * row-major buffer: at step k the pivot is the entry of ROW k (columns >= k)
 * with the largest |re|+|im| (izamax), columns are interchanged, the row tail
 * is scaled by the reciprocal pivot and the trailing block gets a rank-1
 * update. Solve = U^T forward, L^T backward, then the interchanges in
 * reverse (zgetrs 'T'). Published algorithm of LAPACK 3.x zgetf2/zgetrs;
 * third-party pins: ndarray-linalg 0.18.0 / lax 0.18.0 / openblas-src 0.10.13.
 * ====================================================================== */
typedef struct { int n, k; mao_c64* A; int r0, r1; } rank1_job;
static void* rank1_rows(void* arg) {
  rank1_job* J = (rank1_job*)arg;
  int n = J->n, k = J->k; mao_c64* A = J->A;
  const mao_c64* rk = A + (size_t)k * n;
  for (int r = J->r0; r < J->r1; ++r) {
    mao_c64* rr = A + (size_t)r * n;
    mao_c64 m = rr[k];
    if (m.re == 0.0 && m.im == 0.0) continue;
    for (int i = k + 1; i < n; ++i) {
      rr[i].re -= m.re * rk[i].re - m.im * rk[i].im;
      rr[i].im -= m.re * rk[i].im + m.im * rk[i].re;
    }
  }
  return NULL;
}

int mao_zgesv(int n, mao_c64* A, mao_c64* b, int* ipiv_out, int nthreads) {
  if (n < 0) return 2;
  int* ipiv = ipiv_out ? ipiv_out : (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  int info = 0;
  if (nthreads < 1) nthreads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  rank1_job* jobs = (rank1_job*)malloc(sizeof(rank1_job) * (size_t)nthreads);
  for (int k = 0; k < n; ++k) {
    mao_c64* rk = A + (size_t)k * n;
    int p = k; double best = cabs1(rk[k]);
    for (int i = k + 1; i < n; ++i) { double v = cabs1(rk[i]); if (v > best) { best = v; p = i; } }
    ipiv[k] = p;
    if (rk[p].re == 0.0 && rk[p].im == 0.0) { if (!info) info = k + 1; continue; }
    if (p != k) for (int r = 0; r < n; ++r) { mao_c64 t = A[(size_t)r * n + k]; A[(size_t)r * n + k] = A[(size_t)r * n + p]; A[(size_t)r * n + p] = t; }
    /* reciprocal scaling (zgetf2: |pivot| >= sfmin branch) */
    double pr = rk[k].re, pi = rk[k].im, rre, rim;
    /* LAPACK computes ONE / A(j,j) with the robust complex division (zladiv) */
    if (fabs(pi) < fabs(pr)) { double e = pi / pr, f = pr + pi * e; rre = 1.0 / f; rim = -e / f; }
    else { double e = pr / pi, f = pi + pr * e; rre = e / f; rim = -1.0 / f; }
    for (int i = k + 1; i < n; ++i) {
      double xr = rk[i].re, xi = rk[i].im;
      rk[i].re = xr * rre - xi * rim; rk[i].im = xr * rim + xi * rre;
    }
    int rows = n - (k + 1);
    int nt = rows < 64 ? 1 : nthreads;
    for (int t = 0; t < nt; ++t) {
      rank1_job j = {n, k, A, k + 1 + (int)((long long)rows * t / nt), k + 1 + (int)((long long)rows * (t + 1) / nt)};
      jobs[t] = j;
      if (nt > 1) pthread_create(&th[t], NULL, rank1_rows, &jobs[t]); else rank1_rows(&jobs[0]);
    }
    if (nt > 1) for (int t = 0; t < nt; ++t) pthread_join(th[t], NULL);
  }
  free(th); free(jobs);
  if (info) { if (!ipiv_out) free(ipiv); return 1; }   /* -> LuError::SingularMatrix (lu.rs:145) */
  /* zgetrs 'T' */
  for (int r = 0; r < n; ++r) {                          /* U^T y = b: lower triangle of the buffer, non-unit */
    const mao_c64* rr = A + (size_t)r * n;
    mao_c64 s = b[r];
    for (int c = 0; c < r; ++c) s = csub(s, cmul(rr[c], b[c]));
    b[r] = cdiv(s, rr[r]);
  }
  for (int r = n - 1; r >= 0; --r) {                     /* L^T z = y: strict upper triangle, unit diagonal */
    const mao_c64* rr = A + (size_t)r * n;
    mao_c64 s = b[r];
    for (int c = r + 1; c < n; ++c) s = csub(s, cmul(rr[c], b[c]));
    b[r] = s;
  }
  for (int k = n - 1; k >= 0; --k) if (ipiv[k] != k) { mao_c64 t = b[k]; b[k] = b[ipiv[k]]; b[ipiv[k]] = t; }
  if (!ipiv_out) free(ipiv);
  return 0;
}

/* Literal restatement of the pure-Rust fallback (lu.rs:83-137 factorize, :38-78 solve), including the
 * way solve() replays `pivots` (a permutation vector) as a swap sequence. Not the native path. */
int mao_lu_solve_fallback(int n, const mao_c64* A, const mao_c64* b, mao_c64* x) {
  mao_c64* lu = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n * (size_t)n);
  int* piv = (int*)malloc(sizeof(int) * (size_t)n);
  memcpy(lu, A, sizeof(mao_c64) * (size_t)n * (size_t)n);
  for (int i = 0; i < n; ++i) piv[i] = i;
  int rc = 0;
  for (int k = 0; k < n && !rc; ++k) {
    double mv = cnorm(lu[(size_t)k * n + k]); int mr = k;
    for (int i = k + 1; i < n; ++i) { double v = cnorm(lu[(size_t)i * n + k]); if (v > mv) { mv = v; mr = i; } }
    if (mv < 1e-30) { rc = 1; break; }
    if (mr != k) {
      for (int j = 0; j < n; ++j) { mao_c64 t = lu[(size_t)k * n + j]; lu[(size_t)k * n + j] = lu[(size_t)mr * n + j]; lu[(size_t)mr * n + j] = t; }
      int t = piv[k]; piv[k] = piv[mr]; piv[mr] = t;
    }
    mao_c64 pinv = cinv(lu[(size_t)k * n + k]);
    for (int i = k + 1; i < n; ++i) {
      mao_c64 m = cmul(lu[(size_t)i * n + k], pinv);
      lu[(size_t)i * n + k] = m;
      for (int j = k + 1; j < n; ++j) lu[(size_t)i * n + j] = csub(lu[(size_t)i * n + j], cmul(m, lu[(size_t)k * n + j]));
    }
  }
  if (!rc) {
    memcpy(x, b, sizeof(mao_c64) * (size_t)n);
    for (int i = 0; i < n; ++i) if (piv[i] != i) { mao_c64 t = x[i]; x[i] = x[piv[i]]; x[piv[i]] = t; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) x[i] = csub(x[i], cmul(lu[(size_t)i * n + j], x[j]));
    for (int i = n - 1; i >= 0; --i) {
      for (int j = i + 1; j < n; ++j) x[i] = csub(x[i], cmul(lu[(size_t)i * n + j], x[j]));
      mao_c64 u = lu[(size_t)i * n + i];
      if (cnorm(u) < 1e-30) { rc = 1; break; }
      x[i] = cmul(x[i], cinv(u));
    }
  }
  free(lu); free(piv);
  return rc;
}

/* ======================================================================
 * CSR SpMV — csr.rs:240-292 (row-parallel == sequential arithmetic per row)
 * ====================================================================== */
typedef struct { int r0, r1; const long long* rp; const long long* col; const mao_c64* val; const mao_c64* x; mao_c64* y; } spmv_job;
static void* spmv_rows(void* arg) {
  spmv_job* J = (spmv_job*)arg;
  for (int i = J->r0; i < J->r1; ++i) {
    mao_c64 s = C(0.0, 0.0);
    for (long long idx = J->rp[i]; idx < J->rp[i + 1]; ++idx) s = cadd(s, cmul(J->val[idx], J->x[J->col[idx]]));
    J->y[i] = s;
  }
  return NULL;
}
void mao_csr_matvec(int n, const long long* rp, const long long* col, const mao_c64* val, const mao_c64* x, mao_c64* y, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads == 1) { spmv_job j = {0, n, rp, col, val, x, y}; spmv_rows(&j); return; }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  spmv_job* jobs = (spmv_job*)malloc(sizeof(spmv_job) * (size_t)nthreads);
  for (int t = 0; t < nthreads; ++t) {
    spmv_job j = {(int)((long long)n * t / nthreads), (int)((long long)n * (t + 1) / nthreads), rp, col, val, x, y};
    jobs[t] = j; pthread_create(&th[t], NULL, spmv_rows, &jobs[t]);
  }
  for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
  free(th); free(jobs);
}

/* HelmholtzAssembler::assemble — math-fem/src/assembly/assembler.rs:216-257: A = K - k^2 M, k complex */
void mao_helmholtz_values(long long nnz, const double* K, const double* M, double k_re, double k_im, mao_c64* val) {
  mao_c64 k = C(k_re, k_im), k2 = cmul(k, k);
  for (long long i = 0; i < nnz; ++i) {
    mao_c64 kv = C(K[i], 0.0);
    mao_c64 mv = cscale(k2, M[i]);
    val[i] = csub(kv, mv);
  }
}

/* ======================================================================
 * AMG smoothers — amg.rs:400-413 (diag_inv), 855-884, 887-929, 932-978
 * ====================================================================== */
static mao_c64 csr_get(const long long* rp, const long long* col, const mao_c64* val, int i, int j) {
  for (long long idx = rp[i]; idx < rp[i + 1]; ++idx) if (col[idx] == j) return val[idx];
  return C(0.0, 0.0);
}
void mao_amg_jacobi(int n, const long long* rp, const long long* col, const mao_c64* val,
                    mao_c64* x, const mao_c64* b, double omega, int sweeps, int nthreads) {
  mao_c64* dinv = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  mao_c64* ax = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    mao_c64 d = csr_get(rp, col, val, i, i);
    dinv[i] = cnorm(d) > 1e-15 ? cinv(d) : C(1.0, 0.0);
  }
  mao_c64 om = C(omega, 0.0);
  for (int s = 0; s < sweeps; ++s) {
    mao_csr_matvec(n, rp, col, val, x, ax, nthreads);
    for (int i = 0; i < n; ++i) {
      mao_c64 r = csub(b[i], ax[i]);
      x[i] = cadd(x[i], cmul(cmul(om, dinv[i]), r));
    }
  }
  free(dinv); free(ax);
}
void mao_amg_l1_jacobi(int n, const long long* rp, const long long* col, const mao_c64* val,
                       mao_c64* x, const mao_c64* b, int sweeps, int nthreads) {
  double* l1 = (double*)malloc(sizeof(double) * (size_t)n);
  mao_c64* ax = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (long long idx = rp[i]; idx < rp[i + 1]; ++idx) s += cnorm(val[idx]);
    l1[i] = s > 1e-15 ? s : 1.0;
  }
  for (int s = 0; s < sweeps; ++s) {
    mao_csr_matvec(n, rp, col, val, x, ax, nthreads);
    for (int i = 0; i < n; ++i) {
      mao_c64 r = csub(b[i], ax[i]);
      x[i] = cadd(x[i], cmul(r, cinv(C(l1[i], 0.0))));
    }
  }
  free(l1); free(ax);
}
void mao_amg_sym_gauss_seidel(int n, const long long* rp, const long long* col, const mao_c64* val,
                              mao_c64* x, const mao_c64* b, int sweeps) {
  for (int s = 0; s < sweeps; ++s) {
    for (int pass = 0; pass < 2; ++pass) {
      for (int ii = 0; ii < n; ++ii) {
        int i = pass == 0 ? ii : n - 1 - ii;
        mao_c64 sum = b[i], diag = C(1.0, 0.0);
        for (long long idx = rp[i]; idx < rp[i + 1]; ++idx) {
          if (col[idx] == i) diag = val[idx];
          else sum = csub(sum, cmul(val[idx], x[col[idx]]));
        }
        if (cnorm(diag) > 1e-15) x[i] = cmul(sum, cinv(diag));
      }
    }
  }
}

/* ======================================================================
 * math-fem geometric-MG smoothers on COO triplets — multigrid/smoother.rs:44-176
 * (the HashMap<row, Vec<(col,val)>> keeps per-row entries in COO order; equivalent
 *  to a stable counting sort by row)
 * ====================================================================== */
typedef struct { long long* start; long long* ecol; mao_c64* eval; mao_c64* diag; } coo_rows;
static void build_rows(int n, long long nnz, const long long* rows, const long long* cols, const mao_c64* vals, coo_rows* R) {
  R->start = (long long*)calloc((size_t)n + 1, sizeof(long long));
  R->diag = (mao_c64*)calloc((size_t)n, sizeof(mao_c64));
  for (long long k = 0; k < nnz; ++k) { if (rows[k] == cols[k]) R->diag[rows[k]] = cadd(R->diag[rows[k]], vals[k]); else R->start[rows[k] + 1]++; }
  for (int i = 0; i < n; ++i) R->start[i + 1] += R->start[i];
  long long off = R->start[n];
  R->ecol = (long long*)malloc(sizeof(long long) * (size_t)(off > 0 ? off : 1));
  R->eval = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)(off > 0 ? off : 1));
  long long* cur = (long long*)malloc(sizeof(long long) * (size_t)n);
  memcpy(cur, R->start, sizeof(long long) * (size_t)n);
  for (long long k = 0; k < nnz; ++k) if (rows[k] != cols[k]) { long long p = cur[rows[k]]++; R->ecol[p] = cols[k]; R->eval[p] = vals[k]; }
  free(cur);
}
static void free_rows(coo_rows* R) { free(R->start); free(R->ecol); free(R->eval); free(R->diag); }

static void gs_sweep(int n, const coo_rows* R, mao_c64* x, const mao_c64* b, int backward) {
  for (int ii = 0; ii < n; ++ii) {
    int i = backward ? n - 1 - ii : ii;
    if (cnorm(R->diag[i]) < 1e-15) continue;
    mao_c64 sigma = C(0.0, 0.0);
    for (long long p = R->start[i]; p < R->start[i + 1]; ++p) sigma = cadd(sigma, cmul(R->eval[p], x[R->ecol[p]]));
    x[i] = cdiv(csub(b[i], sigma), R->diag[i]);
  }
}
static void jac_sweep(int n, const coo_rows* R, mao_c64* x, const mao_c64* b, double omega, mao_c64* xn) {
  for (int i = 0; i < n; ++i) {
    if (cnorm(R->diag[i]) < 1e-15) { xn[i] = x[i]; continue; }
    mao_c64 sigma = C(0.0, 0.0);
    for (long long p = R->start[i]; p < R->start[i + 1]; ++p) sigma = cadd(sigma, cmul(R->eval[p], x[R->ecol[p]]));
    mao_c64 xgs = cdiv(csub(b[i], sigma), R->diag[i]);
    xn[i] = cadd(cmul(C(omega, 0.0), xgs), cmul(C(1.0 - omega, 0.0), x[i]));
  }
  memcpy(x, xn, sizeof(mao_c64) * (size_t)n);
}
void mao_fem_smooth(int n, long long nnz, const long long* rows, const long long* cols, const mao_c64* vals,
                    mao_c64* x, const mao_c64* b, int kind, int iterations, double omega) {
  coo_rows R; build_rows(n, nnz, rows, cols, vals, &R);
  mao_c64* xn = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)(n > 0 ? n : 1));
  for (int it = 0; it < iterations; ++it) {
    if (kind == 0) gs_sweep(n, &R, x, b, 0);
    else if (kind == 1) jac_sweep(n, &R, x, b, omega, xn);
    else { gs_sweep(n, &R, x, b, 0); gs_sweep(n, &R, x, b, 1); }
  }
  free(xn); free_rows(&R);
}
void mao_fem_residual(int n, long long nnz, const long long* rows, const long long* cols, const mao_c64* vals,
                      const mao_c64* x, const mao_c64* b, mao_c64* r) {
  memcpy(r, b, sizeof(mao_c64) * (size_t)n);
  for (long long k = 0; k < nnz; ++k) r[rows[k]] = csub(r[rows[k]], cmul(vals[k], x[cols[k]]));
}

/* ======================================================================
 * GMRES(m) — iterative/gmres.rs:105-277, 589-621; blas_helpers.rs:21-73
 * ====================================================================== */
static void op_apply(int n, int kind, const mao_c64* dense, const long long* rp, const long long* col, const mao_c64* val,
                     const mao_c64* x, mao_c64* y) {
  if (kind == 1) { mao_csr_matvec(n, rp, col, val, x, y, 1); return; }
  for (int i = 0; i < n; ++i) {
    mao_c64 s = C(0.0, 0.0);
    const mao_c64* r = dense + (size_t)i * n;
    for (int j = 0; j < n; ++j) s = cadd(s, cmul(r[j], x[j]));
    y[i] = s;
  }
}
static mao_c64 inner(int n, const mao_c64* x, const mao_c64* y) {
  mao_c64 s = C(0.0, 0.0);
  for (int i = 0; i < n; ++i) s = cadd(s, cmul(cconj(x[i]), y[i]));
  return s;
}
static double vnorm(int n, const mao_c64* x) { double s = 0.0; for (int i = 0; i < n; ++i) s += cnorm_sqr(x[i]); return sqrt(s); }
static void axpy(int n, mao_c64 a, const mao_c64* x, mao_c64* y) { for (int i = 0; i < n; ++i) y[i] = cadd(y[i], cmul(a, x[i])); }

static void givens(mao_c64 a, mao_c64 b, mao_c64* c, mao_c64* s) {
  if (cnorm(b) < 1e-30) { *c = C(1.0, 0.0); *s = C(0.0, 0.0); return; }
  if (cnorm(a) < 1e-30) { *c = C(0.0, 0.0); *s = C(1.0, 0.0); return; }
  double r = sqrt(cnorm_sqr(a) + cnorm_sqr(b));
  *c = cmul(a, C(1.0 / r, 0.0)); *s = cmul(b, C(1.0 / r, 0.0));
}

void mao_gmres(int n, int kind, const mao_c64* dense, const long long* rp, const long long* col, const mao_c64* val,
               const mao_c64* b, const mao_c64* x0, int m, int max_it, double tol, mao_c64* x, mao_gmres_info* info) {
  if (x0) memcpy(x, x0, sizeof(mao_c64) * (size_t)n); else memset(x, 0, sizeof(mao_c64) * (size_t)n);
  double bnorm = vnorm(n, b);
  info->iterations = 0; info->restarts = 0; info->converged = 1; info->residual = 0.0;
  if (bnorm < 1e-15) return;
  mao_c64* V = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n * (size_t)(m + 1));
  mao_c64* H = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)(m + 1) * (size_t)m);
  mao_c64* cs = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)m);
  mao_c64* sn = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)m);
  mao_c64* g = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)(m + 1));
  mao_c64* w = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  mao_c64* y = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)m);
  int total = 0, restarts = 0, done = 0;
#define HH(i, j) H[(size_t)(i) * (size_t)m + (size_t)(j)]
  for (int outer = 0; outer < max_it && !done; ++outer) {
    op_apply(n, kind, dense, rp, col, val, x, w);
    mao_c64* r = V;                                   /* v[0] slot */
    for (int i = 0; i < n; ++i) r[i] = csub(b[i], w[i]);
    double beta = vnorm(n, r);
    double rel = beta / bnorm;
    if (rel < tol) { info->iterations = total; info->restarts = restarts; info->residual = rel; info->converged = 1; done = 1; break; }
    mao_c64 ib = C(1.0 / beta, 0.0);
    for (int i = 0; i < n; ++i) r[i] = cmul(r[i], ib);
    memset(H, 0, sizeof(mao_c64) * (size_t)(m + 1) * (size_t)m);
    memset(g, 0, sizeof(mao_c64) * (size_t)(m + 1));
    g[0] = C(beta, 0.0);
    int nv = 1, inner_conv = 0, finished = 0;
    for (int j = 0; j < m; ++j) {
      total += 1;
      op_apply(n, kind, dense, rp, col, val, V + (size_t)j * n, w);
      for (int i = 0; i <= j; ++i) {
        HH(i, j) = inner(n, V + (size_t)i * n, w);
        mao_c64 h = HH(i, j);
        axpy(n, C(-h.re, -h.im), V + (size_t)i * n, w);
      }
      double wn = vnorm(n, w);
      HH(j + 1, j) = C(wn, 0.0);
      if (wn < 1e-14) inner_conv = 1;
      else {
        /* new_v = w; axpy(inv_norm - 1, w, new_v)  (gmres.rs:198-201) */
        mao_c64 a = csub(C(1.0 / wn, 0.0), C(1.0, 0.0));
        mao_c64* nvp = V + (size_t)nv * n;
        memcpy(nvp, w, sizeof(mao_c64) * (size_t)n);
        axpy(n, a, w, nvp);
        nv += 1;
      }
      for (int i = 0; i < j; ++i) {
        mao_c64 t = cadd(cmul(cconj(cs[i]), HH(i, j)), cmul(cconj(sn[i]), HH(i + 1, j)));
        HH(i + 1, j) = cadd(csub(C(0.0, 0.0), cmul(sn[i], HH(i, j))), cmul(cs[i], HH(i + 1, j)));
        HH(i, j) = t;
      }
      mao_c64 c, s; givens(HH(j, j), HH(j + 1, j), &c, &s);
      cs[j] = c; sn[j] = s;
      HH(j, j) = cadd(cmul(cconj(c), HH(j, j)), cmul(cconj(s), HH(j + 1, j)));
      HH(j + 1, j) = C(0.0, 0.0);
      mao_c64 t = cadd(cmul(cconj(c), g[j]), cmul(cconj(s), g[j + 1]));
      g[j + 1] = cadd(csub(C(0.0, 0.0), cmul(s, g[j])), cmul(c, g[j + 1]));
      g[j] = t;
      rel = cnorm(g[j + 1]) / bnorm;
      if (rel < tol || inner_conv) {
        int kk = j + 1;
        for (int i = kk - 1; i >= 0; --i) {
          mao_c64 sum = g[i];
          for (int q = i + 1; q < kk; ++q) sum = csub(sum, cmul(HH(i, q), y[q]));
          y[i] = cnorm(HH(i, i)) > 1e-30 ? cmul(sum, cinv(HH(i, i))) : C(0.0, 0.0);
        }
        for (int i = 0; i < kk; ++i) axpy(n, y[i], V + (size_t)i * n, x);
        info->iterations = total; info->restarts = restarts; info->residual = rel; info->converged = 1;
        finished = 1; done = 1; break;
      }
    }
    if (finished) break;
    for (int i = m - 1; i >= 0; --i) {
      mao_c64 sum = g[i];
      for (int q = i + 1; q < m; ++q) sum = csub(sum, cmul(HH(i, q), y[q]));
      y[i] = cnorm(HH(i, i)) > 1e-30 ? cmul(sum, cinv(HH(i, i))) : C(0.0, 0.0);
    }
    for (int i = 0; i < m; ++i) axpy(n, y[i], V + (size_t)i * n, x);
    restarts += 1;
  }
  if (!done) {
    op_apply(n, kind, dense, rp, col, val, x, w);
    for (int i = 0; i < n; ++i) w[i] = csub(b[i], w[i]);
    info->iterations = total; info->restarts = restarts; info->residual = vnorm(n, w) / bnorm; info->converged = 0;
  }
#undef HH
  free(V); free(H); free(cs); free(sn); free(g); free(w); free(y);
}

/* ======================================================================
 * gmres_preconditioned(_with_guess) — iterative/gmres.rs:282-585, left preconditioning, with the
 * Preconditioner::apply (traits.rs:370-375) of a one-level smoother: z = 0, then `sweeps` Jacobi
 * (kind 1, amg.rs:855-884) or l1-Jacobi (kind 2, amg.rs:887-929) sweeps on A z = r, i.e. what
 * AmgPreconditioner::apply does on its coarsest level (amg.rs:981-1005, 1068-1087).
 * ====================================================================== */
/* ---- AmgPreconditioner::v_cycle / apply (preconditioners/amg.rs:981-1065, 1068-1103) over a given hierarchy ---- */
static void amg_smooth(const mao_amg_hierarchy* H, int l, mao_c64* x, const mao_c64* b, int sweeps) {
  if (H->smoother == 1) mao_amg_l1_jacobi(H->n[l], H->a_rp[l], H->a_col[l], H->a_val[l], x, b, sweeps, 1);
  else if (H->smoother == 2) mao_amg_sym_gauss_seidel(H->n[l], H->a_rp[l], H->a_col[l], H->a_val[l], x, b, sweeps);
  else mao_amg_jacobi(H->n[l], H->a_rp[l], H->a_col[l], H->a_val[l], x, b, H->jacobi_weight, sweeps, 1);
}
static void amg_v_cycle(const mao_amg_hierarchy* H, int level, mao_c64* x, const mao_c64* b) {
  const int n = H->n[level];
  /* amg.rs:985-1003: coarsest level (or no prolongation): 20 Jacobi / 20 l1-Jacobi / 10 symmetric Gauss-Seidel sweeps */
  if (level == H->nlevels - 1 || !H->p_rp[level]) { amg_smooth(H, level, x, b, H->smoother == 2 ? 10 : 20); return; }
  amg_smooth(H, level, x, b, H->num_pre_smooth);                                  /* :1006-1024 */
  mao_c64* r = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  mao_csr_matvec(n, H->a_rp[level], H->a_col[level], H->a_val[level], x, r, 1);    /* :1027 r = b - A x */
  for (int i = 0; i < n; ++i) r[i] = csub(b[i], r[i]);
  const int nc = H->n[level + 1];
  mao_c64* rc = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)nc);
  mao_c64* ec = (mao_c64*)calloc((size_t)nc, sizeof(mao_c64));                     /* :1033-1034 e_c = 0 */
  mao_csr_matvec(nc, H->r_rp[level], H->r_col[level], H->r_val[level], r, rc, 1);  /* :1030 r_c = R r */
  amg_v_cycle(H, level + 1, ec, rc);                                              /* :1037 */
  mao_csr_matvec(n, H->p_rp[level], H->p_col[level], H->p_val[level], ec, r, 1);   /* :1040 e = P e_c */
  for (int i = 0; i < n; ++i) x[i] = cadd(x[i], r[i]);                             /* :1043 */
  amg_smooth(H, level, x, b, H->num_post_smooth);                                 /* :1046-1064 */
  free(r); free(rc); free(ec);
}
void mao_amg_apply(const mao_amg_hierarchy* H, const mao_c64* r, mao_c64* z) {
  const int n = H->n[0];
  memset(z, 0, sizeof(mao_c64) * (size_t)n);                                       /* :1079 */
  amg_v_cycle(H, 0, z, r);
  if (H->cycle == 1) amg_v_cycle(H, 0, z, r);                                      /* W: :1084-1087 */
  if (H->cycle == 2) {                                                             /* F: :1088-1094 */
    mao_c64* res = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
    mao_c64* corr = (mao_c64*)calloc((size_t)n, sizeof(mao_c64));
    mao_csr_matvec(n, H->a_rp[0], H->a_col[0], H->a_val[0], z, res, 1);
    for (int i = 0; i < n; ++i) res[i] = csub(r[i], res[i]);
    amg_v_cycle(H, 0, corr, res);
    for (int i = 0; i < n; ++i) z[i] = cadd(z[i], corr[i]);
    free(res); free(corr);
  }
}
static const mao_amg_hierarchy* g_amg = NULL;     /* the hierarchy behind pkind 3 (test infrastructure: one caller at a time) */

static void precond_apply(int n, const long long* rp, const long long* col, const mao_c64* val, int kind, double omega, int sweeps,
                          const mao_c64* r, mao_c64* z) {
  if (kind == 0) { memcpy(z, r, sizeof(mao_c64) * (size_t)n); return; }
  if (kind == 3) { mao_amg_apply(g_amg, r, z); return; }
  memset(z, 0, sizeof(mao_c64) * (size_t)n);
  if (kind == 1) mao_amg_jacobi(n, rp, col, val, z, r, omega, sweeps, 1);
  else mao_amg_l1_jacobi(n, rp, col, val, z, r, sweeps, 1);
}

void mao_gmres_preconditioned(int n, const long long* rp, const long long* col, const mao_c64* val, int pkind, double omega, int sweeps,
                              const mao_c64* b, const mao_c64* x0, int m, int max_it, double tol, mao_c64* x, mao_gmres_info* info) {
  if (x0) memcpy(x, x0, sizeof(mao_c64) * (size_t)n); else memset(x, 0, sizeof(mao_c64) * (size_t)n);
  mao_c64* V = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n * (size_t)(m + 1));
  mao_c64* H = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)(m + 1) * (size_t)m);
  mao_c64* cs = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)m);
  mao_c64* sn = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)m);
  mao_c64* g = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)(m + 1));
  mao_c64* w = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  mao_c64* t = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  mao_c64* y = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)m);
  precond_apply(n, rp, col, val, pkind, omega, sweeps, b, w);
  double bnorm = vnorm(n, w);
  info->iterations = 0; info->restarts = 0; info->converged = 1; info->residual = 0.0;
  int total = 0, restarts = 0, done = 0;
  if (bnorm < 1e-15) done = 2;
#define HH(i, j) H[(size_t)(i) * (size_t)m + (size_t)(j)]
  for (int outer = 0; outer < max_it && !done; ++outer) {
    mao_csr_matvec(n, rp, col, val, x, t, 1);
    for (int i = 0; i < n; ++i) t[i] = csub(b[i], t[i]);
    mao_c64* r = V;
    precond_apply(n, rp, col, val, pkind, omega, sweeps, t, r);
    double beta = vnorm(n, r);
    double rel = beta / bnorm;
    if (rel < tol) { info->iterations = total; info->restarts = restarts; info->residual = rel; done = 1; break; }
    mao_c64 ib = C(1.0 / beta, 0.0);
    for (int i = 0; i < n; ++i) r[i] = cmul(r[i], ib);
    memset(H, 0, sizeof(mao_c64) * (size_t)(m + 1) * (size_t)m);
    memset(g, 0, sizeof(mao_c64) * (size_t)(m + 1));
    g[0] = C(beta, 0.0);
    int nv = 1, inner_conv = 0, finished = 0;
    for (int j = 0; j < m; ++j) {
      total += 1;
      mao_csr_matvec(n, rp, col, val, V + (size_t)j * n, t, 1);
      precond_apply(n, rp, col, val, pkind, omega, sweeps, t, w);
      for (int i = 0; i <= j; ++i) {
        HH(i, j) = inner(n, V + (size_t)i * n, w);
        mao_c64 h = HH(i, j);
        const mao_c64* vi = V + (size_t)i * n;
        for (int q = 0; q < n; ++q) w[q] = csub(w[q], cmul(vi[q], h));
      }
      double wn = vnorm(n, w);
      HH(j + 1, j) = C(wn, 0.0);
      if (wn < 1e-14) inner_conv = 1;
      else { mao_c64 iw = C(1.0 / wn, 0.0); mao_c64* nvp = V + (size_t)nv * n; for (int q = 0; q < n; ++q) nvp[q] = cmul(w[q], iw); nv += 1; }
      for (int i = 0; i < j; ++i) {
        mao_c64 tt = cadd(cmul(cconj(cs[i]), HH(i, j)), cmul(cconj(sn[i]), HH(i + 1, j)));
        HH(i + 1, j) = cadd(csub(C(0.0, 0.0), cmul(sn[i], HH(i, j))), cmul(cs[i], HH(i + 1, j)));
        HH(i, j) = tt;
      }
      mao_c64 c, s2; givens(HH(j, j), HH(j + 1, j), &c, &s2);
      cs[j] = c; sn[j] = s2;
      HH(j, j) = cadd(cmul(cconj(c), HH(j, j)), cmul(cconj(s2), HH(j + 1, j)));
      HH(j + 1, j) = C(0.0, 0.0);
      mao_c64 tt = cadd(cmul(cconj(c), g[j]), cmul(cconj(s2), g[j + 1]));
      g[j + 1] = cadd(csub(C(0.0, 0.0), cmul(s2, g[j])), cmul(c, g[j + 1]));
      g[j] = tt;
      rel = cnorm(g[j + 1]) / bnorm;
      if (rel < tol || inner_conv) {
        int kk = j + 1;
        for (int i = kk - 1; i >= 0; --i) {
          mao_c64 sum = g[i];
          for (int q = i + 1; q < kk; ++q) sum = csub(sum, cmul(HH(i, q), y[q]));
          y[i] = cnorm(HH(i, i)) > 1e-30 ? cmul(sum, cinv(HH(i, i))) : C(0.0, 0.0);
        }
        for (int i = 0; i < kk; ++i) { const mao_c64* vi = V + (size_t)i * n; for (int q = 0; q < n; ++q) x[q] = cadd(x[q], cmul(vi[q], y[i])); }
        info->iterations = total; info->restarts = restarts; info->residual = rel; finished = 1; done = 1; break;
      }
    }
    if (finished) break;
    for (int i = m - 1; i >= 0; --i) {
      mao_c64 sum = g[i];
      for (int q = i + 1; q < m; ++q) sum = csub(sum, cmul(HH(i, q), y[q]));
      y[i] = cnorm(HH(i, i)) > 1e-30 ? cmul(sum, cinv(HH(i, i))) : C(0.0, 0.0);
    }
    for (int i = 0; i < m; ++i) { const mao_c64* vi = V + (size_t)i * n; for (int q = 0; q < n; ++q) x[q] = cadd(x[q], cmul(vi[q], y[i])); }
    restarts += 1;
  }
  if (!done) {
    mao_csr_matvec(n, rp, col, val, x, t, 1);
    for (int i = 0; i < n; ++i) t[i] = csub(b[i], t[i]);
    precond_apply(n, rp, col, val, pkind, omega, sweeps, t, w);
    info->iterations = total; info->restarts = restarts; info->residual = vnorm(n, w) / bnorm; info->converged = 0;
  }
#undef HH
  free(V); free(H); free(cs); free(sn); free(g); free(w); free(t); free(y);
}

/* gmres_preconditioned with an AmgPreconditioner (SolverType::GmresAmg, math-fem/src/solver/mod.rs:667): the fine operator is
 * the hierarchy's level 0 */
void mao_gmres_amg(const mao_amg_hierarchy* H, const mao_c64* b, const mao_c64* x0, int restart, int max_iterations, double tol,
                   mao_c64* x, mao_gmres_info* info) {
  g_amg = H;
  mao_gmres_preconditioned(H->n[0], H->a_rp[0], H->a_col[0], H->a_val[0], 3, 0.0, 0, b, x0, restart, max_iterations, tol, x, info);
  g_amg = NULL;
}

/* ======================================================================
 * gmres_pipelined — iterative/gmres_pipelined.rs:18-250 (p-GMRES, Ghysels et al.): auxiliary basis Z = M^-1 A V, the dot
 * products of step j (classical Gram-Schmidt of z_j against v_0..v_j) are independent of q = M^-1 A z_j, which the reference
 * computes concurrently (rayon::join, :106-119). op_kind 0 dense / 1 CSR; pkind 0 identity, 1 Jacobi, 2 l1-Jacobi (on a CSR
 * operator), 3 the AMG hierarchy of mao_gmres_pipelined_amg.
 * ====================================================================== */
void mao_gmres_pipelined(int n, int op_kind, const mao_c64* dense, const long long* rp, const long long* col, const mao_c64* val,
                         int pkind, double omega, int sweeps, const mao_c64* b, const mao_c64* x0, int m, int max_it, double tol,
                         mao_c64* x, mao_gmres_info* info) {
  if (x0) memcpy(x, x0, sizeof(mao_c64) * (size_t)n); else memset(x, 0, sizeof(mao_c64) * (size_t)n);
  mao_c64* V = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n * (size_t)(m + 1));
  mao_c64* Z = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n * (size_t)(m + 1));
  mao_c64* H = (mao_c64*)calloc((size_t)(m + 1) * (size_t)m, sizeof(mao_c64));
  mao_c64* cs = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)m);
  mao_c64* sn = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)m);
  mao_c64* g = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)(m + 1));
  mao_c64* t = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  mao_c64* q = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  mao_c64* vn = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)n);
  mao_c64* y = (mao_c64*)malloc(sizeof(mao_c64) * (size_t)m);
#define HH(i, j) H[(size_t)(i) * (size_t)m + (size_t)(j)]
  /* :40-48: r0 = M^-1 (b - A x0); b_norm = |r0| */
  op_apply(n, op_kind, dense, rp, col, val, x, t);
  for (int i = 0; i < n; ++i) t[i] = csub(b[i], t[i]);
  precond_apply(n, rp, col, val, pkind, omega, sweeps, t, q);
  const double bnorm = vnorm(n, q);
  info->iterations = 0; info->restarts = 0; info->converged = 1; info->residual = 0.0;
  int total = 0, restarts = 0, done = 0;
  if (bnorm < 1e-15) done = 2;                                                     /* :52-60 */
  for (int outer = 0; outer < max_it && !done; ++outer) {
    op_apply(n, op_kind, dense, rp, col, val, x, t);                               /* :72-76 */
    for (int i = 0; i < n; ++i) t[i] = csub(b[i], t[i]);
    precond_apply(n, rp, col, val, pkind, omega, sweeps, t, V);
    const double beta = vnorm(n, V);
    double rel = beta / bnorm;
    if (rel < tol) { info->iterations = total; info->restarts = restarts; info->residual = rel; done = 1; break; }
    { const mao_c64 ib = C(1.0 / beta, 0.0); for (int i = 0; i < n; ++i) V[i] = cmul(V[i], ib); }     /* :93 */
    op_apply(n, op_kind, dense, rp, col, val, V, t);                               /* :96-97 z0 = M^-1 A v0 */
    precond_apply(n, rp, col, val, pkind, omega, sweeps, t, Z);
    memset(H, 0, sizeof(mao_c64) * (size_t)(m + 1) * (size_t)m);
    memset(g, 0, sizeof(mao_c64) * (size_t)(m + 1));
    g[0] = C(beta, 0.0);
    int nv = 1, inner_conv = 0, finished = 0;
    for (int j = 0; j < m; ++j) {
      total += 1;
      const mao_c64* zj = Z + (size_t)j * n;
      op_apply(n, op_kind, dense, rp, col, val, zj, t);                            /* :107-110 q = M^-1 A z_j */
      precond_apply(n, rp, col, val, pkind, omega, sweeps, t, q);
      for (int i = 0; i <= j; ++i) HH(i, j) = inner(n, V + (size_t)i * n, zj);     /* :111-118 */
      memcpy(vn, zj, sizeof(mao_c64) * (size_t)n);                                 /* :129-136 */
      for (int i = 0; i <= j; ++i) {
        const mao_c64 f = HH(i, j); const mao_c64* vi = V + (size_t)i * n; const mao_c64* zi = Z + (size_t)i * n;
        for (int p = 0; p < n; ++p) { vn[p] = csub(vn[p], cmul(vi[p], f)); q[p] = csub(q[p], cmul(zi[p], f)); }
      }
      const double nrm = vnorm(n, vn);                                             /* :139-140 */
      HH(j + 1, j) = C(nrm, 0.0);
      if (nrm < 1e-14) inner_conv = 1;                                             /* :143-151 */
      else {
        const mao_c64 in = C(1.0 / nrm, 0.0);
        mao_c64* vp = V + (size_t)nv * n; mao_c64* zp = Z + (size_t)nv * n;
        for (int p = 0; p < n; ++p) { vp[p] = cmul(vn[p], in); zp[p] = cmul(q[p], in); }
        nv += 1;
      }
      for (int i = 0; i < j; ++i) {                                                /* :154-158 */
        mao_c64 tt = cadd(cmul(cconj(cs[i]), HH(i, j)), cmul(cconj(sn[i]), HH(i + 1, j)));
        HH(i + 1, j) = cadd(csub(C(0.0, 0.0), cmul(sn[i], HH(i, j))), cmul(cs[i], HH(i + 1, j)));
        HH(i, j) = tt;
      }
      mao_c64 c, s2; givens(HH(j, j), HH(j + 1, j), &c, &s2);
      cs[j] = c; sn[j] = s2;
      HH(j, j) = cadd(cmul(cconj(c), HH(j, j)), cmul(cconj(s2), HH(j + 1, j)));
      HH(j + 1, j) = C(0.0, 0.0);
      mao_c64 tt = cadd(cmul(cconj(c), g[j]), cmul(cconj(s2), g[j + 1]));
      g[j + 1] = cadd(csub(C(0.0, 0.0), cmul(s2, g[j])), cmul(c, g[j + 1]));
      g[j] = tt;
      rel = cnorm(g[j + 1]) / bnorm;
      if (rel < tol || inner_conv) {                                               /* :181-195 */
        const int kk = j + 1;
        for (int i = kk - 1; i >= 0; --i) {
          mao_c64 sum = g[i];
          for (int p = i + 1; p < kk; ++p) sum = csub(sum, cmul(HH(i, p), y[p]));
          y[i] = cnorm(HH(i, i)) > 1e-30 ? cmul(sum, cinv(HH(i, i))) : C(0.0, 0.0);
        }
        for (int i = 0; i < kk; ++i) axpy(n, y[i], V + (size_t)i * n, x);
        info->iterations = total; info->restarts = restarts; info->residual = rel; finished = 1; done = 1; break;
      }
    }
    if (finished) break;
    for (int i = m - 1; i >= 0; --i) {                                             /* :199-203 */
      mao_c64 sum = g[i];
      for (int p = i + 1; p < m; ++p) sum = csub(sum, cmul(HH(i, p), y[p]));
      y[i] = cnorm(HH(i, i)) > 1e-30 ? cmul(sum, cinv(HH(i, i))) : C(0.0, 0.0);
    }
    for (int i = 0; i < m; ++i) axpy(n, y[i], V + (size_t)i * n, x);
    restarts += 1;
  }
  if (!done) {                                                                     /* :207-219 */
    op_apply(n, op_kind, dense, rp, col, val, x, t);
    for (int i = 0; i < n; ++i) t[i] = csub(b[i], t[i]);
    precond_apply(n, rp, col, val, pkind, omega, sweeps, t, q);
    info->iterations = total; info->restarts = restarts; info->residual = vnorm(n, q) / bnorm; info->converged = 0;
  }
#undef HH
  free(V); free(Z); free(H); free(cs); free(sn); free(g); free(t); free(q); free(vn); free(y);
}
void mao_gmres_pipelined_amg(const mao_amg_hierarchy* Hh, const mao_c64* b, const mao_c64* x0, int restart, int max_iterations, double tol,
                             mao_c64* x, mao_gmres_info* info) {
  g_amg = Hh;
  mao_gmres_pipelined(Hh->n[0], 1, NULL, Hh->a_rp[0], Hh->a_col[0], Hh->a_val[0], 3, 0.0, 0, b, x0, restart, max_iterations, tol, x, info);
  g_amg = NULL;
}
