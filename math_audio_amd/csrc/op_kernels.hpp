// op_kernels.hpp — launchers of the operator-apply and BLAS-1 kernels.
#pragma once
#include "bem_kernels.hpp"

namespace ma {

int op_upload_tables(const double tri13_scaled[13][3]);
int op_launch_zgemv(long long n, const c64* A, const c64* x, c64* y, hipStream_t st);
// y = A^T x or A^H x; partial holds op_zgemv_t_chunks() * n entries
int op_launch_zgemv_t(long long n, const c64* A, const c64* x, c64* partial, c64* y, bool conj, hipStream_t st);
int op_zgemv_t_chunks();
int op_launch_conj(long long n, const c64* in, c64* out, hipStream_t st);
int op_launch_diag_invert(long long n, const c64* d, long long ds, const int* map, c64* inv, hipStream_t st);
int op_launch_cmul(long long n, const c64* a, const c64* x, c64* z, hipStream_t st);
// mode 0: out = conj(x).y ; mode 1: out = ||x||_2 (real part)
int op_launch_dot(long long n, const c64* x, const c64* y, int mode, c64* partial, c64* out, hipStream_t st);
int op_launch_axpy_dev(long long n, const c64* alpha_dev, double sgn, const c64* x, c64* y, hipStream_t st);
// one modified Gram-Schmidt step (h_0..h_j, w updated, |w|) as one launch; MA_ERR_UNSUPPORTED when n is too long for it
int op_launch_gmres_mgs(long long n, const c64* V, int j, c64* w, void* slots, c64* scal_out, unsigned* err, hipStream_t st);
int op_mgs_slot_bytes(int m);
// `count` inner products against one vector in two launches; a -= sum h_i V_i (and b -= sum h_i Z_i) in one
int op_launch_multi_dot(long long n, const c64* V, int count, const c64* y, c64* partial /* count x 256 */, c64* out, hipStream_t st);
int op_launch_multi_axpy(long long n, int count, const c64* h_dev, const c64* V, c64* a, const c64* Z, c64* b, hipStream_t st);
int op_launch_axpy_host(long long n, double are, double aim, const c64* x, c64* y, hipStream_t st);
int op_launch_axpby(long long n, double are, double aim, const c64* x, double bre, double bim, const c64* y, c64* out, hipStream_t st);
int op_launch_tbem_matvec_t(const BemGeom& g, const BemPhys& ph, int row0, int row1, int nchunks, const c64* x, c64* partial,
                            const long long* t_off, const int* t_idx, const int2* pairs, const c64* corr, const c64* diag_corr, c64* y, hipStream_t st);
int op_tbem_matvec_strips(int np);
int op_launch_tbem_matvec(const BemGeom& g, const BemPhys& ph, int row0, int row1, int nchunks, const c64* x, c64* partial,
                          const long long* pair_off, const int2* pairs, const c64* corr, const c64* diag_corr, c64* y, hipStream_t st);
int op_launch_pairs13(const BemGeom& g, const BemPhys& ph, const int2* pairs, long long npairs, c64* out, hipStream_t st);
int op_launch_sub_inplace(long long n, c64* corr, const c64* a13, hipStream_t st);

}  // namespace ma
