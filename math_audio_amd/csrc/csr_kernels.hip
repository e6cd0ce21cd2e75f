// csr_kernels.hip — complex CSR SpMV and Jacobi-type smoother sweeps for gfx950 (HBM-bound).
//
// Replaces CsrMatrix::matvec (math-solvers/src/sparse/csr.rs:240-292), the per-frequency value
// update of HelmholtzAssembler::assemble (math-fem/src/assembly/assembler.rs:216-257) and the AMG
// smoothers smooth_jacobi / smooth_l1_jacobi (math-solvers/src/preconditioners/amg.rs:855-929).
//
// Layout: one shared pattern (row_ptr i64, col i32) and either complex values (generic
// CsrMatrix<Complex64>) or the two real arrays K and M of the Helmholtz sweep, from which
// a_ij = K_ij - k^2 M_ij is formed in registers (16 B per non-zero either way; the separate
// "assemble" pass and its 32 B/nnz of traffic disappear).
// Kernel: CSR-vector with sub-wavefront groups: G = 4..64 lanes per row (the power of two at or
// above the mean row length), consecutive rows on consecutive groups so that a wavefront's loads of
// col/val are one contiguous run; x is gathered through L2; the group reduces with DPP shuffles.
// The Jacobi sweep is the same kernel with a fused epilogue x_new = x + w_i (b - A x)_i.
#include "csr_kernels.hpp"
#include "lu_kernels.hpp"
#include "ma_device_math.hpp"

namespace ma {

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// EPI: 0 y = A x; 1 r = b - A x; 2 Jacobi x_new = x + omega dinv (b - A x); 3 l1-Jacobi x_new = x + (b - A x) / l1;
//      4 y = b + A x (the correction x + P e of the V-cycle in one pass; out may be b)
template <int G, bool KM, int EPI>
__global__ __launch_bounds__(256) void csr_rows_kernel(CsrView A, const dc* __restrict__ x, const dc* b /* EPI 4 is used in place: b == out */,
                                                       dc* out, double omega) {
  const int lane_in_group = threadIdx.x & (G - 1);
  const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) / G;
  if (row >= A.n) return;                            // whole groups leave together (G divides 64)
  const long long beg = A.row_ptr[row], end = A.row_ptr[row + 1];
  double sr = 0.0, si = 0.0;
  for (long long idx = beg + lane_in_group; idx < end; idx += G) {
    double ar, ai;
    if (KM) { const double kv = A.K[idx], mv = A.M[idx]; ar = kv - A.k2_re * mv; ai = -(A.k2_im * mv); }
    else { const dc v = A.val[idx]; ar = v.re; ai = v.im; }
    const dc xv = x[A.col[idx]];
    sr += ar * xv.re - ai * xv.im;
    si += ar * xv.im + ai * xv.re;
  }
  sr = group_sum<G>(sr); si = group_sum<G>(si);
  if (lane_in_group == 0) {
    if (EPI == 0) out[row] = dc_make(sr, si);
    else if (EPI == 4) { const dc bb = b[row]; out[row] = dc_make(bb.re + sr, bb.im + si); }
    else {
      const dc bb = b[row];
      const double rr = bb.re - sr, ri = bb.im - si;
      if (EPI == 1) out[row] = dc_make(rr, ri);
      else if (EPI == 2) {
        const dc d = A.dinv[row]; const dc xo = x[row];
        const double wr = omega * d.re, wi = omega * d.im;               // omega * diag_inv[i] (amg.rs:871)
        out[row] = dc_make(xo.re + (wr * rr - wi * ri), xo.im + (wr * ri + wi * rr));
      } else {
        const double l = A.l1[row]; const dc xo = x[row];
        out[row] = dc_make(xo.re + rr / l, xo.im + ri / l);              // r[i] * (1 / l1_diag[i]) (amg.rs:916)
      }
    }
  }
}

// Sliced-ELLPACK form of the same product: lane = row, slice = 64 consecutive rows = one wavefront. The k-th entries of
// the slice's rows sit side by side, so every load of K, M (or values) and col is one contiguous 512-B / 256-B run and
// no cross-lane reduction is needed; rows of a structured FEM mesh have neighbouring columns, so the gather of x is
// nearly coalesced too. Used when padding to the slice's longest row costs < 30 % (a P1 tet mesh: 4 %).
template <bool KM, int EPI, bool C16>
__global__ __launch_bounds__(256) void sell_rows_kernel(CsrView A, const dc* __restrict__ x, const dc* b /* EPI 4 is used in place: b == out */,
                                                        dc* out, double omega) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long slice = row >> 6;
  const int lane = threadIdx.x & 63;
  if (slice * 64 >= A.n) return;
  const long long beg = A.sell_ptr[slice];
  const int width = (int)((A.sell_ptr[slice + 1] - beg) >> 6);
  double sr = 0.0, si = 0.0;
  const long long base = beg + lane;
  const int rowc = (int)(row < A.n ? row : A.n - 1);       // 16-bit columns are stored relative to the row
#pragma unroll 4
  for (int kk = 0; kk < width; ++kk) {
    const long long idx = base + (long long)kk * 64;
    double ar, ai;
    if (KM) { const double kv = A.sell_K[idx], mv = A.sell_M[idx]; ar = kv - A.k2_re * mv; ai = -(A.k2_im * mv); }
    else { const dc v = A.sell_val[idx]; ar = v.re; ai = v.im; }
    const dc xv = x[C16 ? rowc + (int)A.sell_col16[idx] : A.sell_col[idx]];
    sr += ar * xv.re - ai * xv.im;
    si += ar * xv.im + ai * xv.re;
  }
  if (row >= A.n) return;
  if (EPI == 0) out[row] = dc_make(sr, si);
  else if (EPI == 4) { const dc bb = b[row]; out[row] = dc_make(bb.re + sr, bb.im + si); }
  else {
    const dc bb = b[row];
    const double rr = bb.re - sr, ri = bb.im - si;
    if (EPI == 1) out[row] = dc_make(rr, ri);
    else if (EPI == 2) {
      const dc d = A.dinv[row]; const dc xo = x[row];
      const double wr = omega * d.re, wi = omega * d.im;
      out[row] = dc_make(xo.re + (wr * rr - wi * ri), xo.im + (wr * ri + wi * rr));
    } else {
      const double l = A.l1[row]; const dc xo = x[row];
      out[row] = dc_make(xo.re + rr / l, xo.im + ri / l);
    }
  }
}

template <bool KM, bool C16>
static int launch_sell2(const CsrView& A, int epi, const c64* x, const c64* b, c64* out, double omega, hipStream_t st) {
  dim3 grid((unsigned)((A.n + 255) / 256)), block(256);
  const dc* xx = reinterpret_cast<const dc*>(x); const dc* bb = reinterpret_cast<const dc*>(b); dc* oo = reinterpret_cast<dc*>(out);
  switch (epi) {
    case 0: hipLaunchKernelGGL((sell_rows_kernel<KM, 0, C16>), grid, block, 0, st, A, xx, bb, oo, omega); break;
    case 1: hipLaunchKernelGGL((sell_rows_kernel<KM, 1, C16>), grid, block, 0, st, A, xx, bb, oo, omega); break;
    case 2: hipLaunchKernelGGL((sell_rows_kernel<KM, 2, C16>), grid, block, 0, st, A, xx, bb, oo, omega); break;
    case 4: hipLaunchKernelGGL((sell_rows_kernel<KM, 4, C16>), grid, block, 0, st, A, xx, bb, oo, omega); break;
    default: hipLaunchKernelGGL((sell_rows_kernel<KM, 3, C16>), grid, block, 0, st, A, xx, bb, oo, omega); break;
  }
  MA_HIP(hipGetLastError());
  return MA_OK;
}
template <bool KM>
static int launch_sell(const CsrView& A, int epi, const c64* x, const c64* b, c64* out, double omega, hipStream_t st) {
  return A.sell_col16 ? launch_sell2<KM, true>(A, epi, x, b, out, omega, st) : launch_sell2<KM, false>(A, epi, x, b, out, omega, st);
}

// per-wavenumber diagonal data: dinv_i = 1 / a_ii (1 if |a_ii| <= 1e-15, amg.rs:400-413) and
// l1_i = sum_j |a_ij| (1 if <= 1e-15, amg.rs:895-908)
template <bool KM>
__global__ __launch_bounds__(256) void csr_diag_kernel(CsrView A, dc* __restrict__ dinv, double* __restrict__ l1) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= A.n) return;
  double sum = 0.0; dc d = dc_make(0.0, 0.0);
  for (long long idx = A.row_ptr[row]; idx < A.row_ptr[row + 1]; ++idx) {
    double ar, ai;
    if (KM) { const double kv = A.K[idx], mv = A.M[idx]; ar = kv - A.k2_re * mv; ai = -(A.k2_im * mv); }
    else { const dc v = A.val[idx]; ar = v.re; ai = v.im; }
    sum += hypot(ar, ai);
    if (A.col[idx] == row) d = dc_make(ar, ai);          // CsrMatrix::get(i, i): sorted unique columns
  }
  const double nd = hypot(d.re, d.im);
  if (nd > 1e-15) { const double ns = d.re * d.re + d.im * d.im; dinv[row] = dc_make(d.re / ns, -d.im / ns); }   // Complex::inv()
  else dinv[row] = dc_make(A.zero_diag_dinv, 0.0);
  l1[row] = sum > 1e-15 ? sum : 1.0;
}

template <int G, bool KM>
static int launch_g(const CsrView& A, int epi, const c64* x, const c64* b, c64* out, double omega, hipStream_t st) {
  const long long threads = A.n * (long long)G;
  dim3 grid((unsigned)((threads + 255) / 256)), block(256);
  const dc* xx = reinterpret_cast<const dc*>(x); const dc* bb = reinterpret_cast<const dc*>(b); dc* oo = reinterpret_cast<dc*>(out);
  switch (epi) {
    case 0: hipLaunchKernelGGL((csr_rows_kernel<G, KM, 0>), grid, block, 0, st, A, xx, bb, oo, omega); break;
    case 1: hipLaunchKernelGGL((csr_rows_kernel<G, KM, 1>), grid, block, 0, st, A, xx, bb, oo, omega); break;
    case 2: hipLaunchKernelGGL((csr_rows_kernel<G, KM, 2>), grid, block, 0, st, A, xx, bb, oo, omega); break;
    case 4: hipLaunchKernelGGL((csr_rows_kernel<G, KM, 4>), grid, block, 0, st, A, xx, bb, oo, omega); break;
    default: hipLaunchKernelGGL((csr_rows_kernel<G, KM, 3>), grid, block, 0, st, A, xx, bb, oo, omega); break;
  }
  MA_HIP(hipGetLastError());
  return MA_OK;
}

template <bool KM>
static int launch_km(const CsrView& A, int group, int epi, const c64* x, const c64* b, c64* out, double omega, hipStream_t st) {
  switch (group) {
    case 4: return launch_g<4, KM>(A, epi, x, b, out, omega, st);
    case 8: return launch_g<8, KM>(A, epi, x, b, out, omega, st);
    case 16: return launch_g<16, KM>(A, epi, x, b, out, omega, st);
    case 32: return launch_g<32, KM>(A, epi, x, b, out, omega, st);
    default: return launch_g<64, KM>(A, epi, x, b, out, omega, st);
  }
}

// the first Jacobi / l1-Jacobi sweep of an iterate that is zero: x_new = 0 + omega dinv (b - 0) resp. 0 + (b - 0) / l1, the EPI 2 / 3
// epilogues with A x = 0 spelled out -- no pass over the matrix and no clearing of x beforehand
__global__ __launch_bounds__(256) void csr_sweep_from_zero_kernel(long long n, const dc* __restrict__ dinv, const double* __restrict__ l1, const dc* __restrict__ b,
                                                                  double omega, int l1mode, dc* __restrict__ out) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  const dc bb = b[row];
  const double rr = bb.re - 0.0, ri = bb.im - 0.0;
  if (!l1mode) {
    const dc d = dinv[row];
    const double wr = omega * d.re, wi = omega * d.im;
    out[row] = dc_make(0.0 + (wr * rr - wi * ri), 0.0 + (wr * ri + wi * rr));
  } else {
    const double l = l1[row];
    out[row] = dc_make(0.0 + rr / l, 0.0 + ri / l);
  }
}
int csr_launch_sweep_from_zero(const CsrView& A, int l1mode, const c64* b, c64* out, double omega, hipStream_t st) {
  if (A.n <= 0) return MA_OK;
  hipLaunchKernelGGL(csr_sweep_from_zero_kernel, dim3((unsigned)((A.n + 255) / 256)), dim3(256), 0, st, A.n, A.dinv, A.l1, reinterpret_cast<const dc*>(b), omega, l1mode,
                     reinterpret_cast<dc*>(out));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int csr_launch_rows(const CsrView& A, bool km, int group, int epi, const c64* x, const c64* b, c64* out, double omega, hipStream_t st) {
  if (A.n <= 0) return MA_OK;
  if (A.sell_ptr) return km ? launch_sell<true>(A, epi, x, b, out, omega, st) : launch_sell<false>(A, epi, x, b, out, omega, st);
  return km ? launch_km<true>(A, group, epi, x, b, out, omega, st) : launch_km<false>(A, group, epi, x, b, out, omega, st);
}

// HelmholtzAssembler::assemble with boundary terms (assembler.rs:216-257): val[i] = K[i] - k^2 M[i] + sum_t c_t B_t[i],
// a term only where B_t[i] != 0
struct CsrBoundary { int nb; const double* B[8]; double cre[8], cim[8]; };
__global__ __launch_bounds__(256) void csr_assemble_kernel(long long nnz, const double* __restrict__ K, const double* __restrict__ M, double k2re, double k2im,
                                                           CsrBoundary bd, dc* __restrict__ val) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nnz) return;
  const double mv = M[i];
  double vr = K[i] - k2re * mv, vi = -(k2im * mv);
  for (int t = 0; t < bd.nb; ++t) { const double b = bd.B[t][i]; if (b != 0.0) { vr += bd.cre[t] * b; vi += bd.cim[t] * b; } }
  val[i] = dc_make(vr, vi);
}
__global__ __launch_bounds__(256) void sell_gather_kernel(long long tot, const int* __restrict__ src, const dc* __restrict__ val, dc* __restrict__ out) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= tot) return;
  const int s = src[q];
  out[q] = s >= 0 ? val[s] : dc_make(0.0, 0.0);
}
int csr_launch_assemble(long long nnz, const double* K, const double* M, double k2re, double k2im, int nb, const double* const* B, const double* cre, const double* cim,
                        c64* val, long long sell_tot, const int* sell_src, c64* sell_val, hipStream_t st) {
  if (nnz <= 0) return MA_OK;
  CsrBoundary bd{}; bd.nb = nb;
  for (int t = 0; t < nb; ++t) { bd.B[t] = B[t]; bd.cre[t] = cre[t]; bd.cim[t] = cim[t]; }
  hipLaunchKernelGGL(csr_assemble_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, nnz, K, M, k2re, k2im, bd, reinterpret_cast<dc*>(val));
  if (sell_src && sell_val && sell_tot > 0)
    hipLaunchKernelGGL(sell_gather_kernel, dim3((unsigned)((sell_tot + 255) / 256)), dim3(256), 0, st, sell_tot, sell_src, reinterpret_cast<const dc*>(val), reinterpret_cast<dc*>(sell_val));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int csr_launch_diag(const CsrView& A, bool km, c64* dinv, double* l1, hipStream_t st) {
  if (A.n <= 0) return MA_OK;
  dim3 grid((unsigned)((A.n + 255) / 256)), block(256);
  if (km) hipLaunchKernelGGL(csr_diag_kernel<true>, grid, block, 0, st, A, reinterpret_cast<dc*>(dinv), l1);
  else hipLaunchKernelGGL(csr_diag_kernel<false>, grid, block, 0, st, A, reinterpret_cast<dc*>(dinv), l1);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// One Gauss-Seidel step for the rows of one dependency level. Level scheduling (csr_plan.hip: build_levels): a row's
// level is one more than the highest level among the earlier rows it shares an entry with (in either direction), so the
// rows of a level neither read nor write each other and everything a row reads from earlier rows is final: the sweep
// equals the reference's sequential loop over the rows in index order. 16 lanes per row (the launches are latency-bound:
// two dependent loads per entry); the row's products are summed by a lane tree, i.e. in a different association than the
// reference's running sum (rounding-level difference only).
// MODE 0 (smoother.rs:71-117): sigma = sum_{j != i} a_ij x_j, x_i = (b_i - sigma) / a_ii, rows with |a_ii| < 1e-15 skipped
// MODE 1 (amg.rs:932-978):     x_i = (b_i - sum_{j != i} a_ij x_j) * a_ii.inv(), a_ii = 1 when the row stores none,
//                              rows with |a_ii| <= 1e-15 skipped
template <bool KM, int MODE>
__global__ __launch_bounds__(256) void csr_gs_level_kernel(CsrView A, const int* __restrict__ rows, int count, dc* x, const dc* __restrict__ b) {
  constexpr int G = 16;
  const int t = (blockIdx.x * 256 + threadIdx.x) / G, lg = threadIdx.x & (G - 1);
  if (t >= count) return;                               // whole groups leave together
  const int i = rows[t];
  const long long beg = A.row_ptr[i], end = A.row_ptr[i + 1];
  double sr = 0.0, si = 0.0, dr = 0.0, di = 0.0, have = 0.0;
  for (long long idx = beg + lg; idx < end; idx += G) {
    double ar, ai;
    if (KM) { const double kv = A.K[idx], mv = A.M[idx]; ar = kv - A.k2_re * mv; ai = -(A.k2_im * mv); }
    else { const dc v = A.val[idx]; ar = v.re; ai = v.im; }
    const int j = A.col[idx];
    if (j == i) { dr += ar; di += ai; have = 1.0; }       // one stored diagonal per row on this path (duplicates are pre-summed)
    else { const dc xv = x[j]; sr += ar * xv.re - ai * xv.im; si += ar * xv.im + ai * xv.re; }
  }
  sr = group_sum<G>(sr); si = group_sum<G>(si); dr = group_sum<G>(dr); di = group_sum<G>(di); have = group_sum<G>(have);
  if (lg != 0) return;
  if (MODE >= 1 && have == 0.0) { dr = 1.0; di = 0.0; }
  const double nd = hypot(dr, di);
  if (MODE == 2) {          // IluPreconditioner::apply's back substitution (ilu.rs:154-170): the sum always, the division only for |u_ii| > 1e-30
    const dc bb = b[i];
    const double nr = bb.re - sr, ni = bb.im - si, ns = dr * dr + di * di;
    if (nd > 1e-30) { const double ir = dr / ns, ii = -di / ns; x[i] = dc_make(nr * ir - ni * ii, nr * ii + ni * ir); }
    else x[i] = dc_make(nr, ni);
    return;
  }
  if (MODE == 1 ? !(nd > 1e-15) : (nd < 1e-15)) return;
  const dc bb = b[i];
  const double nr = bb.re - sr, ni = bb.im - si, ns = dr * dr + di * di;
  if (MODE == 1) {          // sum * diag.inv(): inv = conj / norm_sqr
    const double ir = dr / ns, ii = -di / ns;
    x[i] = dc_make(nr * ir - ni * ii, nr * ii + ni * ir);
  } else {                  // (b - sigma) / diag
    x[i] = dc_make((nr * dr + ni * di) / ns, (ni * dr - nr * di) / ns);
  }
}

// One Gauss-Seidel sweep as ONE persistent launch: the workgroups walk the dependency levels together, with a device-wide barrier
// between two levels instead of a kernel boundary (298 levels per direction on the 10^6-DoF box: 596 launches of ~3000 rows
// each per symmetric sweep were launch-latency-bound). Rows are handled exactly as in csr_gs_level_kernel (16 lanes per row, the
// same reduction), so the results are the level launches' bit for bit. x is the only array written during the sweep and read
// by other workgroups afterwards: its loads and stores go to the device coherence point (agent-scope relaxed atomics = sc1
// accesses on gfx950: the per-XCD L2s are not coherent with each other inside a kernel), a workgroup's stores are drained
// (s_waitcnt vmcnt(0)) before it arrives at the barrier, and the barrier is a monotonic counter (bar[0]) that workgroup
// leaders increment and poll. The grid is sized by the caller to be co-resident (no LDS, few registers: one workgroup per
// CU always fits next to anything that terminates); every spin is bounded (2 s) and reports through bar[1].
__device__ __forceinline__ dc ld_coherent(const dc* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return dc_make(__longlong_as_double((long long)a), __longlong_as_double((long long)b));
}
__device__ __forceinline__ void st_coherent(dc* p, dc v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  __hip_atomic_store(q, (unsigned long long)__double_as_longlong(v.re), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, (unsigned long long)__double_as_longlong(v.im), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool KM, int MODE>
__global__ __launch_bounds__(256) void csr_gs_persistent_kernel(CsrView A, const int* __restrict__ rows, const long long* __restrict__ lev_ptr, int nlev,
                                                                dc* x, const dc* __restrict__ b, unsigned* bar, unsigned base, unsigned gbase) {
  constexpr int G = 16;
  const int lg = threadIdx.x & (G - 1);
  const long long group0 = ((long long)blockIdx.x * 256 + threadIdx.x) / G, ngroups = (long long)gridDim.x * (256 / G);
  for (int L = 0; L < nlev; ++L) {
    const long long beg_l = lev_ptr[L], end_l = lev_ptr[L + 1];
    for (long long t = beg_l + group0; t < end_l; t += ngroups) {
      const int i = rows[t];
      const long long beg = A.row_ptr[i], end = A.row_ptr[i + 1];
      double sr = 0.0, si = 0.0, dr = 0.0, di = 0.0, have = 0.0;
      for (long long idx = beg + lg; idx < end; idx += G) {
        double ar, ai;
        if (KM) { const double kv = A.K[idx], mv = A.M[idx]; ar = kv - A.k2_re * mv; ai = -(A.k2_im * mv); }
        else { const dc v = A.val[idx]; ar = v.re; ai = v.im; }
        const int j = A.col[idx];
        if (j == i) { dr += ar; di += ai; have = 1.0; }
        else { const dc xv = ld_coherent(x + j); sr += ar * xv.re - ai * xv.im; si += ar * xv.im + ai * xv.re; }
      }
      sr = group_sum<G>(sr); si = group_sum<G>(si); dr = group_sum<G>(dr); di = group_sum<G>(di); have = group_sum<G>(have);
      if (lg != 0) continue;
      if (MODE == 1 && have == 0.0) { dr = 1.0; di = 0.0; }
      const double nd = hypot(dr, di);
      if (MODE == 1 ? !(nd > 1e-15) : (nd < 1e-15)) continue;
      const dc bb = b[i];
      const double nr = bb.re - sr, ni = bb.im - si, ns = dr * dr + di * di;
      if (MODE == 1) { const double ir = dr / ns, ii = -di / ns; st_coherent(x + i, dc_make(nr * ir - ni * ii, nr * ii + ni * ir)); }
      else st_coherent(x + i, dc_make((nr * dr + ni * di) / ns, (ni * dr - nr * di) / ns));
    }
    if (L + 1 == nlev) break;                              // the kernel boundary orders the last level
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's x stores have reached the coherence point
    __shared__ unsigned s_dead;
    if (threadIdx.x == 0) s_dead = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
      // two-level arrival: 8 groups (workgroup b -> group b % 8 = its XCD under the round-robin placement) count their members on
      // their own cache lines; the last member of a group arrives at the global counter, which everybody polls
      const unsigned ngrp = gridDim.x < 8u ? gridDim.x : 8u, grp = blockIdx.x % ngrp;
      const unsigned gsize = (gridDim.x - grp + ngrp - 1) / ngrp;
      const unsigned old = __hip_atomic_fetch_add(bar + 32 * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((old + 1u - gbase) == (unsigned)(L + 1) * gsize) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = base + (unsigned)(L + 1) * ngrp;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      // a barrier that has failed once (this launch or an earlier one of this handle: the word is sticky until ma_csr_status) is not
      // waited on again: with the arrival counters out of step every remaining level would spin for its full 2 s
      if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) s_dead = 1u;
      unsigned spins = 0;
      while (!s_dead && (int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0u && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) s_dead = 1u;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_dead = 1u; }   // 2 s at 100 MHz
      }
    }
    __syncthreads();
    if (s_dead) return;                                    // the whole workgroup leaves (ma_csr_status / the Krylov drivers report it)
  }
}
// The same sweep without a barrier and without flags: the new iterate is written to a second array xn that starts as a sentinel
// (a NaN payload no arithmetic produces), and a row that needs a NEW value polls xn[j] until both words have left the sentinel --
// the value is its own flag, so a dependency costs one coherent round trip after the write lands (2.4 us per level against 3.7 us
// with a flag per row and 4.9 us per dependent launch). Old values come from x, which the sweep does not touch: no row can
// overwrite what another still has to read, so every pattern qualifies. The rows come in level order, every level padded to whole
// wavefronts (four 16-lane row groups) so that no wavefront holds a row together with one that waits for it; the grid is
// co-resident (one workgroup per CU), so the lowest unfinished level can always run. Arithmetic per row as in
// csr_gs_level_kernel: bit-identical results. A result word that happens to equal the sentinel gets its lowest payload bit
// flipped (still the same NaN class). Every wait is bounded (2 s); an abandoned wait raises err[1], after which nobody waits.
constexpr unsigned long long GS_SENTINEL = 0x7FFC0DE0DEADBEEFull;
__device__ __forceinline__ unsigned long long gs_word(double v) {
  const unsigned long long w = (unsigned long long)__double_as_longlong(v);
  return w == GS_SENTINEL ? (w ^ 1ull) : w;
}
__global__ __launch_bounds__(256) void csr_gs_fill_sentinel_kernel(long long n2, unsigned long long* __restrict__ p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n2) p[i] = GS_SENTINEL;
}
template <bool KM, int MODE>
__global__ __launch_bounds__(256) void csr_gs_flags_kernel(CsrView A, const int* __restrict__ rows, long long npad, const dc* __restrict__ x, dc* xn,
                                                           const dc* __restrict__ b, int backward, unsigned* err, unsigned* gerr) {
  constexpr int G = 16;
  const int lg = threadIdx.x & (G - 1);
  const long long group0 = ((long long)blockIdx.x * 256 + threadIdx.x) / G, ngroups = (long long)gridDim.x * (256 / G);
  bool dead = false;
  for (long long t = group0; t < npad; t += ngroups) {
    const int i = rows[t];
    double sr = 0.0, si = 0.0, dr = 0.0, di = 0.0, have = 0.0;
    if (i >= 0) {
      const long long beg = A.row_ptr[i], end = A.row_ptr[i + 1];
      for (long long idx = beg + lg; idx < end; idx += G) {
        double ar, ai;
        if (KM) { const double kv = A.K[idx], mv = A.M[idx]; ar = kv - A.k2_re * mv; ai = -(A.k2_im * mv); }
        else { const dc v = A.val[idx]; ar = v.re; ai = v.im; }
        const int j = A.col[idx];
        if (j == i) { dr += ar; di += ai; have = 1.0; continue; }
        dc xv;
        if (backward ? j > i : j < i) {                       // a new value: wait until it is there
          const unsigned long long* q = reinterpret_cast<const unsigned long long*>(xn + j);
          unsigned long long wa = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), wb = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((wa == GS_SENTINEL || wb == GS_SENTINEL) && !dead) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned spins = 0;
            do {
              __builtin_amdgcn_s_sleep(1);
              wa = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); wb = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if ((++spins & 1023u) == 0u) {
                if (__hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { dead = true; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
                  __hip_atomic_store(err + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  if (gerr) __hip_atomic_store(gerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // the device-wide word the Krylov drivers read
                  dead = true; break;
                }
              }
            } while (wa == GS_SENTINEL || wb == GS_SENTINEL);
          }
          xv = dc_make(__longlong_as_double((long long)wa), __longlong_as_double((long long)wb));
        } else xv = x[j];                                     // an old value: x is not written during the sweep
        sr += ar * xv.re - ai * xv.im; si += ar * xv.im + ai * xv.re;
      }
    }
    sr = group_sum<G>(sr); si = group_sum<G>(si); dr = group_sum<G>(dr); di = group_sum<G>(di); have = group_sum<G>(have);
    if (lg != 0 || i < 0) continue;
    if (MODE >= 1 && have == 0.0) { dr = 1.0; di = 0.0; }
    const double nd = hypot(dr, di);
    dc out = x[i];                                            // a skipped row keeps its value
    if (MODE == 2) {                                          // ilu.rs:154-170: the sum always, the division only for |u_ii| > 1e-30
      const dc bb = b[i];
      const double nr = bb.re - sr, ni = bb.im - si, ns = dr * dr + di * di;
      if (nd > 1e-30) { const double ir = dr / ns, ii = -di / ns; out = dc_make(nr * ir - ni * ii, nr * ii + ni * ir); }
      else out = dc_make(nr, ni);
    } else if (!(MODE == 1 ? !(nd > 1e-15) : (nd < 1e-15))) {
      const dc bb = b[i];
      const double nr = bb.re - sr, ni = bb.im - si, ns = dr * dr + di * di;
      if (MODE == 1) { const double ir = dr / ns, ii = -di / ns; out = dc_make(nr * ir - ni * ii, nr * ii + ni * ir); }
      else out = dc_make((nr * dr + ni * di) / ns, (ni * dr - nr * di) / ns);
    }
    unsigned long long* q = reinterpret_cast<unsigned long long*>(xn + i);
    __hip_atomic_store(q, gs_word(out.re), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, gs_word(out.im), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// x <- one sweep of x; xn is the handle's second array (n entries)
int csr_launch_gs_flags(const CsrView& A, bool km, int mode, const int* rows_padded, long long npad, int grid, c64* x, c64* xn, const c64* b, int backward, unsigned* err,
                        hipStream_t st) {
  if (npad <= 0 || A.n <= 0) return MA_OK;
  const dc* xx = reinterpret_cast<const dc*>(x); dc* nn = reinterpret_cast<dc*>(xn); const dc* bb = reinterpret_cast<const dc*>(b);
  // the sweep's workgroups wait for one another's rows: the launch goes through the same admission window as the LU panel kernels
  // (registers of this instantiation from the runtime, once; the CUs of the device)
  MA_REQUIRE(mode >= 0 && mode <= 2 && !(km && mode == 2), MA_ERR_INVALID, "sweep mode %d", mode);   // mode 2 (triangular solve) only on stored values
  static int regs_of[5] = {0, 0, 0, 0, 0}; static int ncu_of[16] = {};
  const int which = mode == 2 ? 4 : (km ? 2 : 0) + (mode ? 1 : 0);
  int dev = 0; MA_HIP(hipGetDevice(&dev));
  if (regs_of[which] == 0) {
    const void* f = km ? (mode ? reinterpret_cast<const void*>(csr_gs_flags_kernel<true, 1>) : reinterpret_cast<const void*>(csr_gs_flags_kernel<true, 0>))
                       : (mode == 2 ? reinterpret_cast<const void*>(csr_gs_flags_kernel<false, 2>)
                                    : (mode ? reinterpret_cast<const void*>(csr_gs_flags_kernel<false, 1>) : reinterpret_cast<const void*>(csr_gs_flags_kernel<false, 0>)));
    hipFuncAttributes fa; MA_HIP(hipFuncGetAttributes(&fa, f));
    regs_of[which] = fa.numRegs > 0 ? fa.numRegs : 128;
  }
  if (dev >= 0 && dev < 16 && ncu_of[dev] == 0) { hipDeviceProp_t prop; MA_HIP(hipGetDeviceProperties(&prop, dev)); ncu_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
  const int ncu = (dev >= 0 && dev < 16) ? ncu_of[dev] : 256;
  hipLaunchKernelGGL(csr_gs_fill_sentinel_kernel, dim3((unsigned)((2 * A.n + 255) / 256)), dim3(256), 0, st, 2 * A.n, reinterpret_cast<unsigned long long*>(xn));
  SpinLaunch guard;
  { const int arc = guard.admit(st, grid, 0, regs_of[which], ncu); if (arc) return arc; }
  unsigned* gerr = spin_error_word();
  dim3 g((unsigned)grid), block(256);
  if (km) { if (mode) hipLaunchKernelGGL((csr_gs_flags_kernel<true, 1>), g, block, 0, st, A, rows_padded, npad, xx, nn, bb, backward, err, gerr);
            else hipLaunchKernelGGL((csr_gs_flags_kernel<true, 0>), g, block, 0, st, A, rows_padded, npad, xx, nn, bb, backward, err, gerr); }
  else { if (mode == 2) hipLaunchKernelGGL((csr_gs_flags_kernel<false, 2>), g, block, 0, st, A, rows_padded, npad, xx, nn, bb, backward, err, gerr);
         else if (mode) hipLaunchKernelGGL((csr_gs_flags_kernel<false, 1>), g, block, 0, st, A, rows_padded, npad, xx, nn, bb, backward, err, gerr);
         else hipLaunchKernelGGL((csr_gs_flags_kernel<false, 0>), g, block, 0, st, A, rows_padded, npad, xx, nn, bb, backward, err, gerr); }
  MA_HIP(hipGetLastError());
  { const int crc = guard.commit(); if (crc) return crc; }
  MA_HIP(hipMemcpyAsync(x, xn, sizeof(c64) * (size_t)A.n, hipMemcpyDeviceToDevice, st));
  return MA_OK;
}
// returns the number of barrier arrivals the launch adds to bar[0] (the caller keeps the running total for `base`)
int csr_launch_gs_persistent(const CsrView& A, bool km, int mode, const int* rows, const long long* lev_ptr, int nlev, int grid, c64* x, const c64* b,
                             unsigned* bar, unsigned base, unsigned gbase, hipStream_t st) {
  if (nlev <= 0) return MA_OK;
  dc* xx = reinterpret_cast<dc*>(x); const dc* bb = reinterpret_cast<const dc*>(b);
  dim3 g((unsigned)grid), block(256);
  // a device-wide barrier per level: every workgroup must be resident -> the admission window of the spinning kernels
  int dev = 0; MA_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop; MA_HIP(hipGetDeviceProperties(&prop, dev));
  SpinLaunch guard;
  { const int arc = guard.admit(st, grid, 0, 128, prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256); if (arc) return arc; }
  if (km) { if (mode) hipLaunchKernelGGL((csr_gs_persistent_kernel<true, 1>), g, block, 0, st, A, rows, lev_ptr, nlev, xx, bb, bar, base, gbase);
            else hipLaunchKernelGGL((csr_gs_persistent_kernel<true, 0>), g, block, 0, st, A, rows, lev_ptr, nlev, xx, bb, bar, base, gbase); }
  else { if (mode) hipLaunchKernelGGL((csr_gs_persistent_kernel<false, 1>), g, block, 0, st, A, rows, lev_ptr, nlev, xx, bb, bar, base, gbase);
         else hipLaunchKernelGGL((csr_gs_persistent_kernel<false, 0>), g, block, 0, st, A, rows, lev_ptr, nlev, xx, bb, bar, base, gbase); }
  MA_HIP(hipGetLastError());
  return guard.commit();
}

int csr_launch_gs_level(const CsrView& A, bool km, int mode, const int* rows, int count, c64* x, const c64* b, hipStream_t st) {
  if (count <= 0) return MA_OK;
  dim3 grid((unsigned)(((long long)count * 16 + 255) / 256)), block(256);
  dc* xx = reinterpret_cast<dc*>(x); const dc* bb = reinterpret_cast<const dc*>(b);
  if (km) { if (mode) hipLaunchKernelGGL((csr_gs_level_kernel<true, 1>), grid, block, 0, st, A, rows, count, xx, bb);
            else hipLaunchKernelGGL((csr_gs_level_kernel<true, 0>), grid, block, 0, st, A, rows, count, xx, bb); }
  else { if (mode == 2) hipLaunchKernelGGL((csr_gs_level_kernel<false, 2>), grid, block, 0, st, A, rows, count, xx, bb);
         else if (mode) hipLaunchKernelGGL((csr_gs_level_kernel<false, 1>), grid, block, 0, st, A, rows, count, xx, bb);
         else hipLaunchKernelGGL((csr_gs_level_kernel<false, 0>), grid, block, 0, st, A, rows, count, xx, bb); }
  MA_HIP(hipGetLastError());
  return MA_OK;
}

}  // namespace ma
