// csr_kernels.hpp — device view and launchers of the CSR SpMV / smoother kernels.
#pragma once
#include "ma_common.hpp"

namespace ma {

struct dc;

struct CsrView {
  long long n;                 // rows (square operators on this path)
  long long nnz;
  const long long* row_ptr;    // n + 1
  const int* col;              // nnz (usize on the Rust side; narrowed after a range check)
  const ::ma::dc* val;         // complex values, or nullptr in K/M mode
  const double* K;             // stiffness values (K/M mode)
  const double* M;             // mass values (K/M mode)
  double k2_re, k2_im;         // k^2 of the current frequency (K/M mode)
  const ::ma::dc* dinv;        // 1 / a_ii for the current values
  const double* l1;            // sum_j |a_ij|
  // sliced ELLPACK copy (slices of 64 rows = one wavefront, entries column-major inside a slice), or null
  const long long* sell_ptr;   // n_slices + 1 entry offsets
  const int* sell_col;
  const short* sell_col16;     // column - row as 16 bits when every entry of the matrix allows it (banded FEM operators): 2 B less per entry
  const ::ma::dc* sell_val;
  const double* sell_K;
  const double* sell_M;
  double zero_diag_dinv;       // 1/a_ii stand-in for |a_ii| <= 1e-15: 1 (amg.rs:400-413) or 0 = leave the row alone (smoother.rs:143-146)
};

int csr_launch_rows(const CsrView& A, bool km, int group, int epi, const c64* x, const c64* b, c64* out, double omega, hipStream_t st);
int csr_launch_gs_persistent(const CsrView& A, bool km, int mode, const int* rows, const long long* lev_ptr, int nlev, int grid, c64* x, const c64* b,
                             unsigned* bar, unsigned base, unsigned gbase, hipStream_t st);
int csr_launch_sweep_from_zero(const CsrView& A, int l1mode, const c64* b, c64* out, double omega, hipStream_t st);
int csr_launch_gs_level(const CsrView& A, bool km, int mode, const int* rows, int count, c64* x, const c64* b, hipStream_t st);
int csr_launch_gs_flags(const CsrView& A, bool km, int mode, const int* rows_padded, long long npad, int grid, c64* x, c64* xn, const c64* b, int backward, unsigned* err,
                        hipStream_t st);
int csr_launch_diag(const CsrView& A, bool km, c64* dinv, double* l1, hipStream_t st);
int csr_launch_assemble(long long nnz, const double* K, const double* M, double k2re, double k2im, int nb, const double* const* B, const double* cre, const double* cim,
                        c64* val, long long sell_tot, const int* sell_src, c64* sell_val, hipStream_t st);

}  // namespace ma
