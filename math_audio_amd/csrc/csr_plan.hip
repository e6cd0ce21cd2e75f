// csr_plan.hip — C-ABI of the sparse FEM side: CSR operator handles, SpMV, residual, Jacobi sweeps.
#include "csr_kernels.hpp"
#include "lu_kernels.hpp"
#include "ma_device_math.hpp"
#include <vector>
#include <cstdlib>
#include <algorithm>
#include <new>
#include <cmath>

using namespace ma;

struct ma_csr {
  int device = 0;
  long long n = 0, nnz = 0;
  long long ncols = 0;            // = n for the square operators; the AMG transfer operators P / R are rectangular (ma_csr_create_rect)
  bool km = false;
  int group = 16;
  long long* d_rowptr = nullptr; int* d_col = nullptr;
  c64* d_val = nullptr; double* d_K = nullptr; double* d_M = nullptr;
  c64* d_dinv = nullptr; double* d_l1 = nullptr;
  c64* d_x = nullptr; c64* d_y = nullptr; c64* d_b = nullptr;   // staging for the host-buffer entry points and ping-pong
  double k2_re = 0.0, k2_im = 0.0;
  bool diag_valid = false;
  double zero_diag_dinv = 1.0;
  // sliced-ELLPACK copy (null when the padding would cost too much or MA_CSR_SELL=0)
  long long* d_sell_ptr = nullptr; int* d_sell_col = nullptr; short* d_sell_col16 = nullptr; c64* d_sell_val = nullptr; double* d_sell_K = nullptr; double* d_sell_M = nullptr;
  int* d_sell_src = nullptr; long long sell_tot = 0;      // CSR index behind every sliced-ELLPACK slot (-1 = padding)
  // boundary matrices of the HelmholtzAssembler (tag -> real values on the shared pattern); with a non-empty coefficient
  // set the complex values are materialised (assemble) and the kernels run in complex-value mode
  std::vector<int> btags; std::vector<double*> d_B;
  bool materialised = false;
  unsigned long long epoch = 0;   // bumped whenever the operator's values change (set_wavenumber / assemble)
  // Gauss-Seidel level schedules (pattern only, built on first use): [0] forward sweep, [1] backward sweep
  int* d_lev_rows[2] = {nullptr, nullptr};
  std::vector<long long> lev_ptr[2];
  // persistent sweep: the level offsets on the device, the barrier words {arrivals, error} and the arrivals issued so far
  long long* d_lev_ptr[2] = {nullptr, nullptr};
  unsigned* d_gs_bar = nullptr; unsigned gs_bar_count = 0, gs_grp_count = 0; int gs_grid = 0;
  // value-as-flag sweep: the rows in level order with every level padded to whole wavefronts (-1), the second iterate array
  int* d_lev_rows_pad[2] = {nullptr, nullptr}; long long lev_npad[2] = {0, 0}; c64* d_gs_xn = nullptr;
  CsrView view() const {
    CsrView v{};
    v.n = n; v.nnz = nnz; v.row_ptr = d_rowptr; v.col = d_col; v.val = reinterpret_cast<const dc*>(d_val); v.K = d_K; v.M = d_M;
    if (km && materialised) { v.K = nullptr; v.M = nullptr; }
    v.k2_re = k2_re; v.k2_im = k2_im; v.dinv = reinterpret_cast<const dc*>(d_dinv); v.l1 = d_l1;
    v.zero_diag_dinv = zero_diag_dinv;
    v.sell_ptr = d_sell_ptr; v.sell_col = d_sell_col; v.sell_col16 = d_sell_col16; v.sell_val = reinterpret_cast<const dc*>(d_sell_val); v.sell_K = d_sell_K; v.sell_M = d_sell_M;
    if (km && materialised && !d_sell_val) v.sell_ptr = nullptr;      // no complex sliced copy: the CSR-vector kernel serves it
    return v;
  }
  bool fused_km() const { return km && !materialised; }   // kernels form K - k^2 M in registers
};

namespace {
int pick_group(long long n, long long nnz) {
  const double mean = n > 0 ? (double)nnz / (double)n : 1.0;
  int g = 4;
  while (g < 64 && g < mean) g <<= 1;
  return g;
}
void free_all(ma_csr* h) {
  void* p[] = {h->d_rowptr, h->d_col, h->d_val, h->d_K, h->d_M, h->d_dinv, h->d_l1, h->d_x, h->d_y, h->d_b,
               h->d_sell_ptr, h->d_sell_col, h->d_sell_col16, h->d_sell_val, h->d_sell_K, h->d_sell_M, h->d_sell_src};
  for (double* b : h->d_B) if (b) (void)hipFree(b);
  for (int* r : h->d_lev_rows) if (r) (void)hipFree(r);
  for (long long* r : h->d_lev_ptr) if (r) (void)hipFree(r);
  if (h->d_gs_bar) (void)hipFree(h->d_gs_bar);
  for (int d = 0; d < 2; ++d) if (h->d_lev_rows_pad[d]) (void)hipFree(h->d_lev_rows_pad[d]);
  if (h->d_gs_xn) (void)hipFree(h->d_gs_xn);
  for (void* q : p) if (q) (void)hipFree(q);
}
int create_common(int64_t n, const int64_t* rowptr, const int64_t* col, int device, ma_csr** out, int64_t ncols = -1) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (ncols < 0) ncols = n;
  MA_REQUIRE(n > 0 && ncols > 0 && rowptr && col, MA_ERR_INVALID, "bad CSR arguments");
  MA_REQUIRE(rowptr[0] == 0, MA_ERR_INVALID, "row_ptrs[0] must be 0");
  const int64_t nnz = rowptr[n];
  MA_REQUIRE(nnz >= 0, MA_ERR_INVALID, "row_ptrs[n] is negative");
  for (int64_t i = 0; i < n; ++i) MA_REQUIRE(rowptr[i + 1] >= rowptr[i], MA_ERR_INVALID, "row_ptrs must be non-decreasing (row %lld)", (long long)i);
  MA_REQUIRE(n < 2147483647LL && ncols < 2147483647LL, MA_ERR_UNSUPPORTED, "more than 2^31-1 rows or columns");
  std::vector<int> c32((size_t)nnz);
  for (int64_t i = 0; i < nnz; ++i) { MA_REQUIRE(col[i] >= 0 && col[i] < ncols, MA_ERR_INVALID, "column index %lld out of range at %lld", (long long)col[i], (long long)i); c32[(size_t)i] = (int)col[i]; }
  int rc = use_device(device);
  if (rc) return rc;
  ma_csr* h = new (std::nothrow) ma_csr();
  MA_REQUIRE(h, MA_ERR_NOMEM, "host allocation failed");
  h->device = device; h->n = n; h->ncols = ncols; h->nnz = nnz; h->group = pick_group(n, nnz);
  const int64_t nvec = std::max<int64_t>(n, ncols);      // staging vectors serve x (ncols entries) and y (n entries) alike
  hipError_t e = hipMalloc(&h->d_rowptr, sizeof(long long) * (size_t)(n + 1));
  if (e == hipSuccess) e = hipMalloc(&h->d_col, sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  if (e == hipSuccess) e = hipMalloc(&h->d_dinv, sizeof(c64) * (size_t)n);
  if (e == hipSuccess) e = hipMalloc(&h->d_l1, sizeof(double) * (size_t)n);
  if (e == hipSuccess) e = hipMalloc(&h->d_x, sizeof(c64) * (size_t)nvec);
  if (e == hipSuccess) e = hipMalloc(&h->d_y, sizeof(c64) * (size_t)nvec);
  if (e == hipSuccess) e = hipMalloc(&h->d_b, sizeof(c64) * (size_t)nvec);
  if (e == hipSuccess) e = hipMemcpy(h->d_rowptr, rowptr, sizeof(long long) * (size_t)(n + 1), hipMemcpyHostToDevice);
  if (e == hipSuccess && nnz > 0) e = hipMemcpy(h->d_col, c32.data(), sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("CSR upload failed: %s", hipGetErrorString(e)); free_all(h); delete h; return e == hipErrorOutOfMemory ? MA_ERR_NOMEM : MA_ERR_HIP; }
  *out = h;
  return MA_OK;
}
// Sliced-ELLPACK copy of the operator (csr_kernels.hip: sell_rows_kernel). Skipped, leaving the CSR-vector kernel in
// charge, when padding every slice to its longest row would add more than 30 % of entries.
int build_sell(ma_csr* h, const int64_t* rowptr, const int64_t* col, const ma_c64* vals, const double* K, const double* M) {
  if (const char* e = getenv("MA_CSR_SELL")) if (atoi(e) == 0) return MA_OK;
  const int64_t n = h->n, nnz = h->nnz;
  if (nnz <= 0) return MA_OK;
  const int64_t ns = (n + 63) / 64;
  std::vector<long long> sp((size_t)ns + 1, 0);
  for (int64_t s = 0; s < ns; ++s) {
    int64_t w = 0;
    for (int64_t r = s * 64; r < std::min<int64_t>(n, s * 64 + 64); ++r) w = std::max<int64_t>(w, rowptr[r + 1] - rowptr[r]);
    sp[(size_t)s + 1] = sp[(size_t)s] + w * 64;
  }
  const long long tot = sp[(size_t)ns];
  if ((double)tot > 1.3 * (double)nnz + 64.0 * 64.0) return MA_OK;
  std::vector<int> sc((size_t)tot, 0), ssrc((size_t)tot, -1); std::vector<c64> sv; std::vector<double> sk, sm;
  if (vals) sv.assign((size_t)tot, c64{0.0, 0.0}); else { sk.assign((size_t)tot, 0.0); sm.assign((size_t)tot, 0.0); }
  for (int64_t s = 0; s < ns; ++s)
    for (int l = 0; l < 64; ++l) {
      const int64_t r = s * 64 + l;
      const long long w = (sp[(size_t)s + 1] - sp[(size_t)s]) / 64;
      const int64_t len = r < n ? rowptr[r + 1] - rowptr[r] : 0;
      for (long long kk = 0; kk < w; ++kk) {
        const size_t q = (size_t)(sp[(size_t)s] + kk * 64 + l);
        if (kk < len) {
          const int64_t t = rowptr[r] + kk;
          sc[q] = (int)col[t]; ssrc[q] = (int)t;
          if (vals) { sv[q].re = vals[t].re; sv[q].im = vals[t].im; } else { sk[q] = K[t]; sm[q] = M[t]; }
        } else sc[q] = (int)std::min<int64_t>(r < n ? r : 0, h->ncols - 1);   // padding: zero coefficient, a column that is in cache anyway (and exists)
      }
    }
  // columns relative to the row in 16 bits when the whole operator allows it
  bool c16 = true;
  std::vector<short> sc16;
  if (c16) {
    sc16.assign((size_t)tot, 0);
    for (int64_t s2 = 0; s2 < ns && c16; ++s2)
      for (int l = 0; l < 64 && c16; ++l) {
        const int64_t r = s2 * 64 + l, rc_ = std::min<int64_t>(r, n - 1);
        const long long w = (sp[(size_t)s2 + 1] - sp[(size_t)s2]) / 64;
        const int64_t len = r < n ? rowptr[r + 1] - rowptr[r] : 0;
        for (long long kk = 0; kk < w; ++kk) {             // padding slots too: their column must exist (rectangular operators)
          const size_t q = (size_t)(sp[(size_t)s2] + kk * 64 + l);
          const long long dlt = (kk < len ? (long long)col[rowptr[r] + kk] : (long long)sc[q]) - (long long)rc_;
          if (dlt < -32768 || dlt > 32767) { c16 = false; break; }
          sc16[q] = (short)dlt;
        }
      }
  }
  hipError_t e = hipMalloc(&h->d_sell_ptr, sizeof(long long) * ((size_t)ns + 1));
  if (e == hipSuccess) e = hipMalloc(&h->d_sell_col, sizeof(int) * (size_t)tot);
  if (e == hipSuccess && c16) e = hipMalloc(&h->d_sell_col16, sizeof(short) * (size_t)tot);
  if (e == hipSuccess && c16) e = hipMemcpy(h->d_sell_col16, sc16.data(), sizeof(short) * (size_t)tot, hipMemcpyHostToDevice);
  if (e == hipSuccess && vals) e = hipMalloc(&h->d_sell_val, sizeof(c64) * (size_t)tot);
  if (e == hipSuccess && !vals) e = hipMalloc(&h->d_sell_K, sizeof(double) * (size_t)tot);
  if (e == hipSuccess && !vals) e = hipMalloc(&h->d_sell_M, sizeof(double) * (size_t)tot);
  if (e == hipSuccess && !vals && nnz < 2147483647LL) e = hipMalloc(&h->d_sell_src, sizeof(int) * (size_t)tot);
  if (e == hipSuccess && h->d_sell_src) e = hipMemcpy(h->d_sell_src, ssrc.data(), sizeof(int) * (size_t)tot, hipMemcpyHostToDevice);
  h->sell_tot = tot;
  if (e == hipSuccess) e = hipMemcpy(h->d_sell_ptr, sp.data(), sizeof(long long) * ((size_t)ns + 1), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->d_sell_col, sc.data(), sizeof(int) * (size_t)tot, hipMemcpyHostToDevice);
  if (e == hipSuccess && vals) e = hipMemcpy(h->d_sell_val, sv.data(), sizeof(c64) * (size_t)tot, hipMemcpyHostToDevice);
  if (e == hipSuccess && !vals) e = hipMemcpy(h->d_sell_K, sk.data(), sizeof(double) * (size_t)tot, hipMemcpyHostToDevice);
  if (e == hipSuccess && !vals) e = hipMemcpy(h->d_sell_M, sm.data(), sizeof(double) * (size_t)tot, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("sliced-ELLPACK upload failed: %s", hipGetErrorString(e)); return MA_ERR_HIP; }
  return MA_OK;
}
int ensure_diag(ma_csr* h, hipStream_t st) {
  if (h->diag_valid) return MA_OK;
  int rc = csr_launch_diag(h->view(), h->fused_km(), h->d_dinv, h->d_l1, st);
  if (!rc) h->diag_valid = true;
  return rc;
}
}  // namespace

extern "C" {

// CsrMatrix<Complex64>::from_raw_parts (csr.rs:69-99): complex values
int ma_csr_create(int64_t n, const int64_t* row_ptrs, const int64_t* col_indices, const ma_c64* values, int device, ma_csr_t** out) {
  MA_REQUIRE(values, MA_ERR_INVALID, "values is NULL");
  int rc = create_common(n, row_ptrs, col_indices, device, out);
  if (rc) return rc;
  ma_csr* h = *out;
  hipError_t e = hipMalloc(&h->d_val, sizeof(c64) * (size_t)(h->nnz > 0 ? h->nnz : 1));
  if (e == hipSuccess && h->nnz > 0) e = hipMemcpy(h->d_val, values, sizeof(c64) * (size_t)h->nnz, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("CSR value upload failed: %s", hipGetErrorString(e)); free_all(h); delete h; *out = nullptr; return MA_ERR_HIP; }
  if ((rc = build_sell(h, row_ptrs, col_indices, values, nullptr, nullptr))) { free_all(h); delete h; *out = nullptr; return rc; }
  return MA_OK;
}

// A rectangular operator (nrows x ncols): the prolongation P (fine x coarse) and restriction R (coarse x fine) of an AMG level
// (amg.rs:236-243). SpMV only (ma_csr_spmv / ma_csr_spmv_dev: x has ncols entries, y nrows); the diagonal-based sweeps and
// the transpose are for square operators.
int ma_csr_create_rect(int64_t nrows, int64_t ncols, const int64_t* row_ptrs, const int64_t* col_indices, const ma_c64* values, int device, ma_csr_t** out) {
  MA_REQUIRE(values, MA_ERR_INVALID, "values is NULL");
  int rc = create_common(nrows, row_ptrs, col_indices, device, out, ncols);
  if (rc) return rc;
  ma_csr* h = *out;
  hipError_t e = hipMalloc(&h->d_val, sizeof(c64) * (size_t)(h->nnz > 0 ? h->nnz : 1));
  if (e == hipSuccess && h->nnz > 0) e = hipMemcpy(h->d_val, values, sizeof(c64) * (size_t)h->nnz, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("CSR value upload failed: %s", hipGetErrorString(e)); free_all(h); delete h; *out = nullptr; return MA_ERR_HIP; }
  if ((rc = build_sell(h, row_ptrs, col_indices, values, nullptr, nullptr))) { free_all(h); delete h; *out = nullptr; return rc; }
  return MA_OK;
}
int ma_csr_num_cols(const ma_csr_t* h, int64_t* ncols) {
  MA_REQUIRE(h && ncols, MA_ERR_INVALID, "NULL argument");
  *ncols = h->ncols;
  return MA_OK;
}

// HelmholtzAssembler (assembler.rs:19-32): one pattern, real K and M values; a_ij = K_ij - k^2 M_ij per frequency
int ma_csr_create_helmholtz(int64_t n, const int64_t* row_ptrs, const int64_t* col_indices, const double* K, const double* M, int device, ma_csr_t** out) {
  MA_REQUIRE(K && M, MA_ERR_INVALID, "K or M is NULL");
  int rc = create_common(n, row_ptrs, col_indices, device, out);
  if (rc) return rc;
  ma_csr* h = *out;
  h->km = true;
  const size_t nz = (size_t)(h->nnz > 0 ? h->nnz : 1);
  hipError_t e = hipMalloc(&h->d_K, sizeof(double) * nz);
  if (e == hipSuccess) e = hipMalloc(&h->d_M, sizeof(double) * nz);
  if (e == hipSuccess && h->nnz > 0) e = hipMemcpy(h->d_K, K, sizeof(double) * (size_t)h->nnz, hipMemcpyHostToDevice);
  if (e == hipSuccess && h->nnz > 0) e = hipMemcpy(h->d_M, M, sizeof(double) * (size_t)h->nnz, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("K/M upload failed: %s", hipGetErrorString(e)); free_all(h); delete h; *out = nullptr; return MA_ERR_HIP; }
  if ((rc = build_sell(h, row_ptrs, col_indices, nullptr, K, M))) { free_all(h); delete h; *out = nullptr; return rc; }
  return MA_OK;
}

int ma_csr_destroy(ma_csr_t* h) {
  if (!h) return MA_OK;
  (void)hipSetDevice(h->device);
  free_all(h);
  delete h;
  return MA_OK;
}

// HelmholtzAssembler::assemble(k, ..) (assembler.rs:216): here only k^2 changes; no values are rewritten
int ma_csr_set_wavenumber(ma_csr_t* h, double k_re, double k_im) {
  MA_REQUIRE(h, MA_ERR_INVALID, "NULL handle");
  MA_REQUIRE(h->km, MA_ERR_INVALID, "handle holds complex values, not K/M");
  h->k2_re = k_re * k_re - k_im * k_im; h->k2_im = 2.0 * k_re * k_im;
  h->diag_valid = false;
  h->materialised = false;
  h->epoch++;
  return MA_OK;
}

// HelmholtzAssembler.boundary_values[tag] (assembler.rs:19-32): real values on the operator's pattern (nnz entries)
int ma_csr_add_boundary(ma_csr_t* h, int32_t tag, const double* values) {
  MA_REQUIRE(h && values, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(h->km, MA_ERR_INVALID, "boundary matrices belong to a K/M handle");
  MA_REQUIRE(h->btags.size() < 8, MA_ERR_UNSUPPORTED, "at most 8 boundary matrices per handle");
  for (int t : h->btags) MA_REQUIRE(t != tag, MA_ERR_INVALID, "boundary tag %d is already present", tag);
  MA_HIP(hipSetDevice(h->device));
  double* d = nullptr;
  const size_t nz = (size_t)(h->nnz > 0 ? h->nnz : 1);
  if (hipMalloc(&d, sizeof(double) * nz) != hipSuccess) { set_error("boundary matrix allocation failed"); return MA_ERR_NOMEM; }
  if (h->nnz > 0 && hipMemcpy(d, values, sizeof(double) * (size_t)h->nnz, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); set_error("boundary matrix upload failed"); return MA_ERR_HIP; }
  h->btags.push_back(tag); h->d_B.push_back(d);
  return MA_OK;
}

// HelmholtzAssembler::assemble(wavenumber, boundary_coeffs) (assembler.rs:216-257): A = K - k^2 M + sum_t c_t B_t over the
// tags present in both maps. Without boundary terms this is ma_csr_set_wavenumber (values formed in registers); with
// them the complex values are written once per call by a device kernel and the SpMV reads them (28 instead of 20 B/nnz).
int ma_csr_assemble(ma_csr_t* h, double k_re, double k_im, int32_t ncoef, const int32_t* tags, const ma_c64* coeffs, void* stream) {
  MA_REQUIRE(h, MA_ERR_INVALID, "NULL handle");
  MA_REQUIRE(h->km, MA_ERR_INVALID, "handle holds complex values, not K/M");
  MA_REQUIRE(ncoef >= 0 && (ncoef == 0 || (tags && coeffs)), MA_ERR_INVALID, "bad coefficient list");
  h->k2_re = k_re * k_re - k_im * k_im; h->k2_im = 2.0 * k_re * k_im;
  h->diag_valid = false;
  h->epoch++;
  const double* B[8]; double cre[8], cim[8]; int nb = 0;
  for (size_t t = 0; t < h->btags.size(); ++t)
    for (int c = 0; c < ncoef; ++c) if (tags[c] == h->btags[t]) { B[nb] = h->d_B[t]; cre[nb] = coeffs[c].re; cim[nb] = coeffs[c].im; ++nb; break; }
  if (nb == 0) { h->materialised = false; return MA_OK; }
  MA_HIP(hipSetDevice(h->device));
  const size_t nz = (size_t)(h->nnz > 0 ? h->nnz : 1);
  if (!h->d_val) MA_HIP(hipMalloc(&h->d_val, sizeof(c64) * nz));
  if (h->d_sell_ptr && h->d_sell_src && !h->d_sell_val) MA_HIP(hipMalloc(&h->d_sell_val, sizeof(c64) * (size_t)h->sell_tot));
  int rc = csr_launch_assemble(h->nnz, h->d_K, h->d_M, h->k2_re, h->k2_im, nb, B, cre, cim, h->d_val, h->sell_tot, h->d_sell_src,
                               (h->d_sell_ptr && h->d_sell_src) ? h->d_sell_val : nullptr, (hipStream_t)stream);
  if (rc) return rc;
  h->materialised = true;
  return MA_OK;
}

int ma_csr_num_rows(const ma_csr_t* h, int64_t* n, int64_t* nnz) {
  MA_REQUIRE(h && n && nnz, MA_ERR_INVALID, "NULL argument");
  *n = h->n; *nnz = h->nnz;
  return MA_OK;
}

int ma_csr_device(const ma_csr_t* h, int* device) {
  MA_REQUIRE(h && device, MA_ERR_INVALID, "NULL argument");
  *device = h->device;
  return MA_OK;
}
// the operator's CSR arrays back on the host (row_ptrs n+1, col_indices nnz, values nnz; values of a K / M handle are K - k^2 M at the
// wavenumber set last, boundary terms excluded unless the handle was assembled with them)
int ma_csr_get(ma_csr_t* h, int64_t* row_ptrs, int64_t* col_indices, ma_c64* values) {
  MA_REQUIRE(h && row_ptrs && col_indices && values, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(h->device));
  const size_t n = (size_t)h->n, nnz = (size_t)h->nnz;
  std::vector<long long> rp(n + 1); std::vector<int> col(std::max<size_t>(nnz, 1));
  MA_HIP(hipMemcpy(rp.data(), h->d_rowptr, sizeof(long long) * (n + 1), hipMemcpyDeviceToHost));
  if (nnz) MA_HIP(hipMemcpy(col.data(), h->d_col, sizeof(int) * nnz, hipMemcpyDeviceToHost));
  for (size_t i = 0; i <= n; ++i) row_ptrs[i] = rp[i];
  for (size_t q = 0; q < nnz; ++q) col_indices[q] = col[q];
  if (!h->km || h->materialised) { if (nnz) MA_HIP(hipMemcpy(values, h->d_val, sizeof(c64) * nnz, hipMemcpyDeviceToHost)); }
  else {
    std::vector<double> K(std::max<size_t>(nnz, 1)), M(std::max<size_t>(nnz, 1));
    if (nnz) { MA_HIP(hipMemcpy(K.data(), h->d_K, sizeof(double) * nnz, hipMemcpyDeviceToHost)); MA_HIP(hipMemcpy(M.data(), h->d_M, sizeof(double) * nnz, hipMemcpyDeviceToHost)); }
    for (size_t q = 0; q < nnz; ++q) { values[q].re = K[q] - h->k2_re * M[q]; values[q].im = -(h->k2_im * M[q]); }
  }
  return MA_OK;
}

// device-pointer forms (x, y, b, r: n complex128 each; distinct buffers)
int ma_csr_spmv_dev(ma_csr_t* h, const void* d_x, void* d_y, void* stream) {
  MA_REQUIRE(h && d_x && d_y, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(h->device));
  return csr_launch_rows(h->view(), h->fused_km(), h->group, 0, (const c64*)d_x, nullptr, (c64*)d_y, 0.0, (hipStream_t)stream);
}
int ma_csr_residual_dev(ma_csr_t* h, const void* d_x, const void* d_b, void* d_r, void* stream) {
  MA_REQUIRE(h && d_x && d_b && d_r, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(h->device));
  return csr_launch_rows(h->view(), h->fused_km(), h->group, 1, (const c64*)d_x, (const c64*)d_b, (c64*)d_r, 0.0, (hipStream_t)stream);
}
// `sweeps` Jacobi sweeps on d_x (in place from the caller's view; d_tmp is a scratch vector of n entries). x_is_zero: the caller knows
// the iterate is zero and has NOT cleared d_x: the first sweep is the diagonal scaling it then amounts to (no pass over the matrix)
static int jacobi_sweeps(ma_csr* h, int epi, void* d_x, const void* d_b, double omega, int sweeps, void* d_tmp, bool x_is_zero, hipStream_t st) {
  int rc = ensure_diag(h, st);
  c64* cur = (c64*)d_x; c64* nxt = (c64*)d_tmp;
  int s0 = 0;
  if (x_is_zero && sweeps >= 1 && !rc) {
    c64* first = (sweeps % 2 == 1) ? (c64*)d_x : (c64*)d_tmp;                   // so that the last sweep lands in d_x
    rc = csr_launch_sweep_from_zero(h->view(), epi == 3 ? 1 : 0, (const c64*)d_b, first, omega, st);
    cur = first; nxt = first == (c64*)d_x ? (c64*)d_tmp : (c64*)d_x; s0 = 1;
  }
  for (int s = s0; s < sweeps && !rc; ++s) { rc = csr_launch_rows(h->view(), h->fused_km(), h->group, epi, cur, (const c64*)d_b, nxt, omega, st); std::swap(cur, nxt); }
  if (!rc && cur != (c64*)d_x) MA_HIP(hipMemcpyAsync(d_x, cur, sizeof(c64) * (size_t)h->n, hipMemcpyDeviceToDevice, st));
  return rc;
}
int ma_csr_jacobi_dev(ma_csr_t* h, void* d_x, const void* d_b, double omega, int sweeps, void* d_tmp, void* stream) {
  MA_REQUIRE(h && h->ncols == h->n, MA_ERR_INVALID, "ma_csr_jacobi_dev needs a square operator");
  MA_REQUIRE(h && d_x && d_b && d_tmp && sweeps >= 0, MA_ERR_INVALID, "bad argument");
  MA_HIP(hipSetDevice(h->device));
  return jacobi_sweeps(h, 2, d_x, d_b, omega, sweeps, d_tmp, false, (hipStream_t)stream);
}
int ma_csr_l1jacobi_dev(ma_csr_t* h, void* d_x, const void* d_b, int sweeps, void* d_tmp, void* stream) {
  MA_REQUIRE(h && h->ncols == h->n, MA_ERR_INVALID, "ma_csr_l1jacobi_dev needs a square operator");
  MA_REQUIRE(h && d_x && d_b && d_tmp && sweeps >= 0, MA_ERR_INVALID, "bad argument");
  MA_HIP(hipSetDevice(h->device));
  return jacobi_sweeps(h, 3, d_x, d_b, 0.0, sweeps, d_tmp, false, (hipStream_t)stream);
}
// the V-cycle's forms (internal to the library): sweeps of an iterate known to be zero (d_x not cleared by the caller), and x += A e
extern "C" int ma_csr_jacobi_from_zero_dev(ma_csr_t* h, void* d_x, const void* d_b, double omega, int sweeps, void* d_tmp, int l1, void* stream) {
  MA_REQUIRE(h && h->ncols == h->n && d_x && d_b && d_tmp && sweeps >= 1, MA_ERR_INVALID, "bad argument");
  MA_HIP(hipSetDevice(h->device));
  return jacobi_sweeps(h, l1 ? 3 : 2, d_x, d_b, l1 ? 0.0 : omega, sweeps, d_tmp, true, (hipStream_t)stream);
}
extern "C" int ma_csr_spmv_add_dev(ma_csr_t* h, const void* d_e, void* d_x_inout, void* stream) {
  MA_REQUIRE(h && d_e && d_x_inout, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(h->device));
  return csr_launch_rows(h->view(), h->fused_km(), h->group, 4, (const c64*)d_e, (const c64*)d_x_inout, (c64*)d_x_inout, 0.0, (hipStream_t)stream);
}

// host-buffer forms: CsrMatrix::matvec(&x) -> y (csr.rs:240), smooth_jacobi / smooth_l1_jacobi (amg.rs:855-929)
static int up(ma_csr* h, c64* d, const ma_c64* s) { MA_HIP(hipMemcpy(d, s, sizeof(c64) * (size_t)h->n, hipMemcpyHostToDevice)); return MA_OK; }
static int down(ma_csr* h, ma_c64* d, const c64* s) { MA_HIP(hipMemcpy(d, s, sizeof(c64) * (size_t)h->n, hipMemcpyDeviceToHost)); return MA_OK; }

int ma_csr_spmv(ma_csr_t* h, const ma_c64* x, ma_c64* y) {
  MA_REQUIRE(h && x && y, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(h->device));
  int rc = MA_OK;
  MA_HIP(hipMemcpy(h->d_x, x, sizeof(c64) * (size_t)h->ncols, hipMemcpyHostToDevice));
  if (!rc) rc = ma_csr_spmv_dev(h, h->d_x, h->d_y, nullptr);
  if (!rc) rc = down(h, y, h->d_y);
  return rc;
}
int ma_csr_residual(ma_csr_t* h, const ma_c64* x, const ma_c64* b, ma_c64* r) {
  MA_REQUIRE(h && x && b && r, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(h->device));
  int rc = up(h, h->d_x, x);
  if (!rc) rc = up(h, h->d_b, b);
  if (!rc) rc = ma_csr_residual_dev(h, h->d_x, h->d_b, h->d_y, nullptr);
  if (!rc) rc = down(h, r, h->d_y);
  return rc;
}
int ma_csr_jacobi(ma_csr_t* h, ma_c64* x_inout, const ma_c64* b, double omega, int sweeps) {
  MA_REQUIRE(h && x_inout && b, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(h->device));
  int rc = up(h, h->d_x, x_inout);
  if (!rc) rc = up(h, h->d_b, b);
  if (!rc) rc = ma_csr_jacobi_dev(h, h->d_x, h->d_b, omega, sweeps, h->d_y, nullptr);
  if (!rc) rc = down(h, x_inout, h->d_x);
  return rc;
}
int ma_csr_l1jacobi(ma_csr_t* h, ma_c64* x_inout, const ma_c64* b, int sweeps) {
  MA_REQUIRE(h && x_inout && b, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(h->device));
  int rc = up(h, h->d_x, x_inout);
  if (!rc) rc = up(h, h->d_b, b);
  if (!rc) rc = ma_csr_l1jacobi_dev(h, h->d_x, h->d_b, sweeps, h->d_y, nullptr);
  if (!rc) rc = down(h, x_inout, h->d_x);
  return rc;
}

// The transposed operator as its own handle (same value mode, same k^2 and zero-diagonal policy): built once on the host
// by a counting sort of the downloaded pattern. CsrMatrix's LinearOperator::apply_transpose (csr.rs:420-440) is then an
// SpMV on it -- a scatter with atomics would not be reproducible.
int ma_csr_transpose(ma_csr_t* h, ma_csr_t** out) {
  MA_REQUIRE(h && h->ncols == h->n, MA_ERR_INVALID, "ma_csr_transpose needs a square operator");
  MA_REQUIRE(h && out, MA_ERR_INVALID, "NULL argument");
  *out = nullptr;
  MA_HIP(hipSetDevice(h->device));
  const size_t n = (size_t)h->n, nnz = (size_t)h->nnz;
  std::vector<long long> rp(n + 1); std::vector<int> ci(nnz ? nnz : 1);
  MA_HIP(hipMemcpy(rp.data(), h->d_rowptr, sizeof(long long) * (n + 1), hipMemcpyDeviceToHost));
  if (nnz) MA_HIP(hipMemcpy(ci.data(), h->d_col, sizeof(int) * nnz, hipMemcpyDeviceToHost));
  std::vector<c64> val; std::vector<double> K, M;
  const bool km = h->km && !h->materialised;            // a handle assembled with boundary terms transposes its complex values
  if (km) { K.resize(nnz ? nnz : 1); M.resize(nnz ? nnz : 1); if (nnz) { MA_HIP(hipMemcpy(K.data(), h->d_K, sizeof(double) * nnz, hipMemcpyDeviceToHost)); MA_HIP(hipMemcpy(M.data(), h->d_M, sizeof(double) * nnz, hipMemcpyDeviceToHost)); } }
  else { val.resize(nnz ? nnz : 1); if (nnz) MA_HIP(hipMemcpy(val.data(), h->d_val, sizeof(c64) * nnz, hipMemcpyDeviceToHost)); }
  std::vector<int64_t> trp(n + 1, 0), tci(nnz ? nnz : 1);
  for (size_t t = 0; t < nnz; ++t) trp[(size_t)ci[t] + 1]++;
  for (size_t i = 0; i < n; ++i) trp[i + 1] += trp[i];
  std::vector<int64_t> pos(trp.begin(), trp.end() - 1);
  std::vector<c64> tval(val.size()); std::vector<double> tK(K.size()), tM(M.size());
  for (size_t i = 0; i < n; ++i)
    for (long long t = rp[i]; t < rp[i + 1]; ++t) {       // rows ascending: the transposed rows come out with sorted columns
      const size_t q = (size_t)pos[(size_t)ci[(size_t)t]]++;
      tci[q] = (int64_t)i;
      if (km) { tK[q] = K[(size_t)t]; tM[q] = M[(size_t)t]; } else tval[q] = val[(size_t)t];
    }
  int rc = km ? ma_csr_create_helmholtz((int64_t)n, trp.data(), tci.data(), tK.data(), tM.data(), h->device, out)
                 : ma_csr_create((int64_t)n, trp.data(), tci.data(), reinterpret_cast<const ma_c64*>(tval.data()), h->device, out);
  if (rc) return rc;
  (*out)->k2_re = h->k2_re; (*out)->k2_im = h->k2_im; (*out)->zero_diag_dinv = h->zero_diag_dinv; (*out)->diag_valid = false;
  return MA_OK;
}

// Keep a transposed handle in step with its source after the source's values changed: a fused K/M pair only needs the new
// k^2; anything else (boundary terms materialised, or a change of mode) asks for a rebuild (*rebuild = 1).
unsigned long long ma_csr_epoch(const ma_csr_t* h) { return h ? h->epoch : 0; }
int ma_csr_refresh_transpose(const ma_csr_t* src, ma_csr_t* dst, int* rebuild) {
  MA_REQUIRE(src && dst && rebuild, MA_ERR_INVALID, "NULL argument");
  *rebuild = 1;
  if (src->fused_km() && dst->km) {
    dst->k2_re = src->k2_re; dst->k2_im = src->k2_im; dst->materialised = false; dst->diag_valid = false; dst->epoch++;
    *rebuild = 0;
  }
  return MA_OK;
}

// ------------------------------------------------------------------ math-fem HelmholtzMatrix (COO) and its smoothers
// HelmholtzMatrix { rows, cols, values, dim } (math-fem/src/assembly/helmholtz.rs:22-33) holds unsummed triplets;
// the reference's sweeps sum duplicates on the fly (smoother.rs:78-92, 124-137). Here they are summed once, in
// triplet order, into a CSR operator; rows whose diagonal is < 1e-15 are left untouched by the sweeps
// (smoother.rs:94-96, 143-146).
int ma_fem_matrix_create(int64_t n, int64_t nnz, const int64_t* rows, const int64_t* cols, const ma_c64* values, int device, ma_csr_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL");
  *out = nullptr;
  MA_REQUIRE(n > 0 && nnz >= 0 && (nnz == 0 || (rows && cols && values)), MA_ERR_INVALID, "bad COO arguments");
  std::vector<int64_t> cnt((size_t)n + 1, 0);
  for (int64_t t = 0; t < nnz; ++t) {
    MA_REQUIRE(rows[t] >= 0 && rows[t] < n && cols[t] >= 0 && cols[t] < n, MA_ERR_INVALID, "triplet %lld is out of range", (long long)t);
    cnt[(size_t)rows[t] + 1]++;
  }
  for (int64_t i = 0; i < n; ++i) cnt[(size_t)i + 1] += cnt[(size_t)i];
  // bucket by row (stable), then sort each row by column (stable) and merge equal columns in triplet order
  std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1), order((size_t)nnz);
  for (int64_t t = 0; t < nnz; ++t) order[(size_t)pos[(size_t)rows[t]]++] = t;
  std::vector<int64_t> rp((size_t)n + 1, 0), ci; std::vector<ma_c64> vv;
  ci.reserve((size_t)nnz); vv.reserve((size_t)nnz);
  for (int64_t i = 0; i < n; ++i) {
    auto b = order.begin() + cnt[(size_t)i], e = order.begin() + cnt[(size_t)i + 1];
    std::stable_sort(b, e, [&](int64_t a, int64_t c) { return cols[a] < cols[c]; });
    for (auto it = b; it != e; ++it) {
      const int64_t t = *it;
      if (it != b && cols[t] == ci.back()) { vv.back().re += values[t].re; vv.back().im += values[t].im; }
      else { ci.push_back(cols[t]); vv.push_back(values[t]); }
    }
    rp[(size_t)i + 1] = (int64_t)ci.size();
  }
  if (ci.empty()) { ci.push_back(0); vv.push_back(ma_c64{0.0, 0.0}); for (int64_t i = 1; i <= n; ++i) rp[(size_t)i] = 1; }   // an all-zero operator still needs storage
  int rc = ma_csr_create(n, rp.data(), ci.data(), vv.data(), device, out);
  if (rc) return rc;
  (*out)->zero_diag_dinv = 0.0;
  (*out)->diag_valid = false;
  return MA_OK;
}

// ---- Gauss-Seidel by level scheduling
// level_fwd(i) = 1 + max{ level_fwd(j) : j < i, a_ij or a_ji stored }; the backward schedule mirrors it from the last row.
// Rows of one level are independent, and ascending levels reproduce the sequential sweep exactly.
static int build_levels(ma_csr* h) {
  if (h->d_lev_rows[0]) return MA_OK;
  const long long n = h->n, nnz = h->nnz;
  std::vector<long long> rp((size_t)n + 1); std::vector<int> col((size_t)std::max<long long>(nnz, 1));
  MA_HIP(hipMemcpy(rp.data(), h->d_rowptr, sizeof(long long) * ((size_t)n + 1), hipMemcpyDeviceToHost));
  if (nnz > 0) MA_HIP(hipMemcpy(col.data(), h->d_col, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToHost));
  for (int dir = 0; dir < 2; ++dir) {
    std::vector<int> lvl((size_t)n, 0), pend((size_t)n, 0);
    int nlev = 0;
    for (long long ii = 0; ii < n; ++ii) {
      const long long i = dir == 0 ? ii : n - 1 - ii;
      int m = pend[(size_t)i];
      for (long long idx = rp[(size_t)i]; idx < rp[(size_t)i + 1]; ++idx) {
        const long long j = col[(size_t)idx];
        if (j < 0 || j >= n || j == i) continue;
        const bool earlier = dir == 0 ? j < i : j > i;
        if (earlier) m = std::max(m, lvl[(size_t)j]);
      }
      const int L = m + 1;
      lvl[(size_t)i] = L; nlev = std::max(nlev, L);
      for (long long idx = rp[(size_t)i]; idx < rp[(size_t)i + 1]; ++idx) {
        const long long j = col[(size_t)idx];
        if (j < 0 || j >= n || j == i) continue;
        const bool later = dir == 0 ? j > i : j < i;
        if (later) pend[(size_t)j] = std::max(pend[(size_t)j], L);
      }
    }
    std::vector<long long>& lp = h->lev_ptr[dir];
    lp.assign((size_t)nlev + 1, 0);
    for (long long i = 0; i < n; ++i) lp[(size_t)lvl[(size_t)i]]++;           // level L counted at slot L (levels are 1-based)
    for (int L = 1; L <= nlev; ++L) lp[(size_t)L] += lp[(size_t)L - 1];        // lp[L] = one past the last row of level L
    std::vector<long long> cur(lp.begin(), lp.end() - 1);                    // cur[L-1] = first slot of level L
    std::vector<int> rows((size_t)n);
    for (long long ii = 0; ii < n; ++ii) {                                    // sweep order inside a level, for locality
      const long long i = dir == 0 ? ii : n - 1 - ii;
      rows[(size_t)cur[(size_t)lvl[(size_t)i] - 1]++] = (int)i;
    }
    MA_HIP(hipMalloc(&h->d_lev_rows[dir], sizeof(int) * (size_t)n));
    MA_HIP(hipMemcpy(h->d_lev_rows[dir], rows.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    MA_HIP(hipMalloc(&h->d_lev_ptr[dir], sizeof(long long) * lp.size()));
    MA_HIP(hipMemcpy(h->d_lev_ptr[dir], lp.data(), sizeof(long long) * lp.size(), hipMemcpyHostToDevice));
    // the flag-driven sweep's list: levels padded to whole wavefronts (4 row groups of 16 lanes)
    std::vector<int> padded; padded.reserve((size_t)n + 4 * lp.size());
    for (size_t L = 0; L + 1 < lp.size(); ++L) {
      for (long long q = lp[L]; q < lp[L + 1]; ++q) padded.push_back(rows[(size_t)q]);
      while (padded.size() % 4) padded.push_back(-1);
    }
    h->lev_npad[dir] = (long long)padded.size();
    if (!padded.empty()) {
      MA_HIP(hipMalloc(&h->d_lev_rows_pad[dir], sizeof(int) * padded.size()));
      MA_HIP(hipMemcpy(h->d_lev_rows_pad[dir], padded.data(), sizeof(int) * padded.size(), hipMemcpyHostToDevice));
    }
  }
  MA_HIP(hipMalloc(&h->d_gs_xn, sizeof(c64) * (size_t)std::max<long long>(n, 1)));
  MA_HIP(hipMalloc(&h->d_gs_bar, 32 * 9 * sizeof(unsigned)));          // [0] arrivals of groups, [1] error, [32 (1 + g)] members of group g
  MA_HIP(hipMemset(h->d_gs_bar, 0, 32 * 9 * sizeof(unsigned)));
  MA_HIP(hipStreamSynchronize(nullptr));                  // (null-stream memset vs. non-blocking caller streams)
  hipDeviceProp_t prop;
  MA_HIP(hipGetDeviceProperties(&prop, h->device));
  h->gs_grid = prop.multiProcessorCount > 0 ? (prop.multiProcessorCount / 8) * 8 : 256;     // one workgroup per CU (a multiple of 8): co-resident whatever else runs
  if (h->gs_grid < 8) h->gs_grid = 8;
  return MA_OK;
}

// one sweep over all rows in index order (backward = 0) or reverse order (1); mode 0 = smoother.rs:71-117, 1 = amg.rs:932-978
int ma_csr_gauss_seidel_sweep_dev(ma_csr_t* h, void* d_x, const void* d_b, int mode, int backward, void* stream) {
  MA_REQUIRE(h && h->ncols == h->n, MA_ERR_INVALID, "ma_csr_gauss_seidel_sweep_dev needs a square operator");
  MA_REQUIRE(h && d_x && d_b && mode >= 0 && mode <= 2, MA_ERR_INVALID, "bad argument");   // 2: a triangular solve (ilu.rs:154-170), stored values only
  MA_REQUIRE(mode != 2 || !h->fused_km(), MA_ERR_UNSUPPORTED, "the triangular-solve sweep runs on stored values");
  MA_HIP(hipSetDevice(h->device));
  int rc = build_levels(h);
  if (rc) return rc;
  const int dir = backward ? 1 : 0;
  const std::vector<long long>& lp = h->lev_ptr[dir];
  const CsrView v = h->view();
  // MA_CSR_GS_PERSISTENT=1: one persistent launch with a device-wide barrier per level instead of a launch per level. Measured on
  // the 10^6-DoF box (298 levels per direction): 3.39 ms per symmetric sweep (5.40 with a single arrival counter) against 2.91 ms
  // for the launches -- a dependent launch costs 4.9 us here, a barrier plus the level's cold coherent loads 5.7 us -- so the
  // launches stay the default; the results are bit-identical either way (tests/test_csr_gpu.py).
  const char* epers = getenv("MA_CSR_GS_PERSISTENT");
  const bool persistent = epers && atoi(epers) != 0 && mode != 2;
  const int nlev = (int)lp.size() - 1;
  // Default: ONE persistent launch in which the new iterate goes to a second array that starts as a sentinel and a row polls the new
  // values it needs until they have left the sentinel (csr_gs_flags_kernel). Bit-identical to the launches (tools/gs_sweep_modes.py).
  // MA_CSR_GS_FLAGS=0: a launch per level.
  const char* eflags = getenv("MA_CSR_GS_FLAGS");
  if (!(eflags && atoi(eflags) == 0) && !persistent && nlev >= 8 && h->d_lev_rows_pad[dir] && h->d_gs_xn)
    return csr_launch_gs_flags(v, h->fused_km(), mode, h->d_lev_rows_pad[dir], h->lev_npad[dir], h->gs_grid, (c64*)d_x, h->d_gs_xn, (const c64*)d_b, backward ? 1 : 0,
                               h->d_gs_bar, (hipStream_t)stream);
  if (persistent && nlev >= 8) {
    rc = csr_launch_gs_persistent(v, h->fused_km(), mode, h->d_lev_rows[dir], h->d_lev_ptr[dir], nlev, h->gs_grid, (c64*)d_x, (const c64*)d_b, h->d_gs_bar,
                                  h->gs_bar_count, h->gs_grp_count, (hipStream_t)stream);
    // per launch every group counter advances by (nlev - 1) x its size and the global one by (nlev - 1) x groups; the groups have
    // equal sizes when the grid is a multiple of 8 (it is: one workgroup per CU), so one running total serves all of them
    h->gs_bar_count += (unsigned)(nlev - 1) * (unsigned)std::min(h->gs_grid, 8);
    h->gs_grp_count += (unsigned)(nlev - 1) * (unsigned)((h->gs_grid + 7) / 8);
    return rc;
  }
  for (size_t L = 0; L + 1 < lp.size() && !rc; ++L)
    rc = csr_launch_gs_level(v, h->fused_km(), mode, h->d_lev_rows[dir] + lp[L], (int)(lp[L + 1] - lp[L]), (c64*)d_x, (const c64*)d_b, (hipStream_t)stream);
  return rc;
}

// smooth_sym_gauss_seidel(matrix, x, b, num_sweeps) (amg.rs:932-978): forward then backward sweep, num_sweeps times
int ma_csr_sym_gauss_seidel_dev(ma_csr_t* h, void* d_x, const void* d_b, int sweeps, void* stream) {
  MA_REQUIRE(h && h->ncols == h->n, MA_ERR_INVALID, "ma_csr_sym_gauss_seidel_dev needs a square operator");
  MA_REQUIRE(sweeps >= 0, MA_ERR_INVALID, "negative sweep count");
  int rc = MA_OK;
  for (int s = 0; s < sweeps && !rc; ++s) {
    rc = ma_csr_gauss_seidel_sweep_dev(h, d_x, d_b, 1, 0, stream);
    if (!rc) rc = ma_csr_gauss_seidel_sweep_dev(h, d_x, d_b, 1, 1, stream);
  }
  return rc;
}
int ma_csr_sym_gauss_seidel(ma_csr_t* h, ma_c64* x_inout, const ma_c64* b, int sweeps) {
  MA_REQUIRE(h && x_inout && b, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(h->device));
  int rc = up(h, h->d_x, x_inout);
  if (!rc) rc = up(h, h->d_b, b);
  if (!rc) rc = ma_csr_sym_gauss_seidel_dev(h, h->d_x, h->d_b, sweeps, nullptr);
  if (!rc) rc = down(h, x_inout, h->d_x);
  if (!rc) rc = ma_csr_status(h);
  return rc;
}
// MA_ERR_HIP if a persistent Gauss-Seidel sweep of this handle gave up at its device-wide barrier (never seen; the spin is bounded
// so that it cannot hang the device); synchronises the device
int ma_csr_status(ma_csr_t* h) {
  MA_REQUIRE(h, MA_ERR_INVALID, "NULL handle");
  if (!h->d_gs_bar) return MA_OK;
  MA_HIP(hipSetDevice(h->device));
  unsigned w[2] = {0, 0};
  MA_HIP(hipDeviceSynchronize());
  MA_HIP(hipMemcpy(w, h->d_gs_bar, sizeof(w), hipMemcpyDeviceToHost));
  if (w[1] != 0) {
    // reported once: the handle's word and the device-wide one are cleared, later sweeps of the handle wait again
    (void)hipMemset(h->d_gs_bar + 1, 0, sizeof(unsigned));
    (void)spin_error_check("ma_csr_status");
    set_error("a Gauss-Seidel sweep was abandoned at its exchange between workgroups");
    return MA_ERR_HIP;
  }
  return MA_OK;
}
// number of dependency levels of the forward / backward Gauss-Seidel schedule (diagnostics: launches per sweep)
int ma_csr_gauss_seidel_levels(ma_csr_t* h, int64_t* forward, int64_t* backward) {
  MA_REQUIRE(h && h->ncols == h->n, MA_ERR_INVALID, "ma_csr_gauss_seidel_levels needs a square operator");
  MA_REQUIRE(h, MA_ERR_INVALID, "NULL handle");
  MA_HIP(hipSetDevice(h->device));
  int rc = build_levels(h);
  if (rc) return rc;
  if (forward) *forward = (int64_t)h->lev_ptr[0].size() - 1;
  if (backward) *backward = (int64_t)h->lev_ptr[1].size() - 1;
  return MA_OK;
}

// smooth(matrix, x, b, config) (smoother.rs:44-68). kind 0 = Gauss-Seidel (the reference's default): `iterations` forward
// sweeps; kind 1 = Jacobi: x <- omega (b - sigma)/a_ii + (1 - omega) x; kind 2 = symmetric Gauss-Seidel: forward then
// backward sweep per iteration. The Gauss-Seidel sweeps run level by level (see build_levels) and equal the sequential ones.
int ma_fem_smooth(ma_csr_t* h, ma_c64* x_inout, const ma_c64* b, int kind, int iterations, double omega) {
  MA_REQUIRE(h && x_inout && b, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(kind >= 0 && kind <= 2 && iterations >= 0, MA_ERR_INVALID, "smoother kind %d / %d iterations", kind, iterations);
  if (kind == 1) return ma_csr_jacobi(h, x_inout, b, omega, iterations);
  MA_HIP(hipSetDevice(h->device));
  int rc = up(h, h->d_x, x_inout);
  if (!rc) rc = up(h, h->d_b, b);
  for (int it = 0; it < iterations && !rc; ++it) {
    rc = ma_csr_gauss_seidel_sweep_dev(h, h->d_x, h->d_b, 0, 0, nullptr);
    if (!rc && kind == 2) rc = ma_csr_gauss_seidel_sweep_dev(h, h->d_x, h->d_b, 0, 1, nullptr);
  }
  if (!rc) rc = down(h, x_inout, h->d_x);
  return rc;
}

// compute_residual (smoother.rs:163-176): r = b - A x
int ma_fem_residual(ma_csr_t* h, const ma_c64* x, const ma_c64* b, ma_c64* r) { return ma_csr_residual(h, x, b, r); }

}  // extern "C"
