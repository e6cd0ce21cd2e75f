// op_kernels.hip — operator applies and the BLAS-1 pieces of device GMRES for gfx950.
//
//   zgemv_kernel        dense y = A x (DenseOperator::apply = matrix.dot(x), fmm_interface.rs:46-48); HBM-bound.
//   tbem_matvec_kernel  on-the-fly y = A x of the TBEM operator without storing A: lane = collocation
//                       row (its point and normal live in VGPRs), the field panels stream past as
//                       wave-uniform data on the scalar path; each pair is integrated with the same
//                       13-point routine as tbem_far_kernel and immediately multiplied by x_j. Column
//                       chunks go to blockIdx.y and are summed by a second deterministic pass.
//   tbem_corr_*         sparse corrections (near pairs and the diagonal): A_true - A_13pt, computed once
//                       per frequency with the K2/K3 kernels, applied as a small CSR product.
//   dot / axpy / scale  inner_product (conj(x).y), axpy and vector_norm of blas_helpers.rs:21-73 with the
//                       scalar kept on the device between the dot and the axpy of modified Gram-Schmidt.
#include "op_kernels.hpp"
#include "lu_kernels.hpp"
#include <algorithm>
#include "ma_device_math.hpp"

namespace ma {

// ------------------------------------------------------------------ dense y = A x (one wavefront per row)
__global__ __launch_bounds__(256) void zgemv_kernel(long long n, const dc* __restrict__ A, const dc* __restrict__ x, dc* __restrict__ y) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + wave;
  if (row >= n) return;
  const dc* Ar = A + row * n;
  double sr = 0.0, si = 0.0, tr = 0.0, ti = 0.0;
  long long j = lane;
  for (; j + 192 < n; j += 256) {                         // four independent 1-KB loads of the row in flight per wavefront
    const dc a0 = Ar[j], a1 = Ar[j + 64], a2 = Ar[j + 128], a3 = Ar[j + 192];
    const dc v0 = x[j], v1 = x[j + 64], v2 = x[j + 128], v3 = x[j + 192];
    sr += a0.re * v0.re - a0.im * v0.im; si += a0.re * v0.im + a0.im * v0.re;
    tr += a1.re * v1.re - a1.im * v1.im; ti += a1.re * v1.im + a1.im * v1.re;
    sr += a2.re * v2.re - a2.im * v2.im; si += a2.re * v2.im + a2.im * v2.re;
    tr += a3.re * v3.re - a3.im * v3.im; ti += a3.re * v3.im + a3.im * v3.re;
  }
  for (; j < n; j += 64) {
    const dc a = Ar[j], v = x[j];
    sr += a.re * v.re - a.im * v.im; si += a.re * v.im + a.im * v.re;
  }
  sr = wave_sum(sr + tr); si = wave_sum(si + ti);
  if (lane == 0) y[row] = dc_make(sr, si);
}

// ------------------------------------------------------------------ BLAS-1 with device-resident scalars
#define RED_BLOCKS 256
// partial[b] = sum over the block's slice of conj(x) * y   (mode 0) or |x|^2 (mode 1, y unused)
__global__ __launch_bounds__(256) void dot_partial_kernel(long long n, const dc* __restrict__ x, const dc* __restrict__ y, int mode, dc* __restrict__ partial) {
  __shared__ double sr[4], si[4];
  double ar = 0.0, ai = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const dc a = x[i];
    if (mode == 0) { const dc b = y[i]; ar += a.re * b.re + a.im * b.im; ai += a.re * b.im - a.im * b.re; }
    else ar += a.re * a.re + a.im * a.im;
  }
  ar = wave_sum(ar); ai = wave_sum(ai);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sr[wave] = ar; si[wave] = ai; }
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = dc_make((sr[0] + sr[1]) + (sr[2] + sr[3]), (si[0] + si[1]) + (si[2] + si[3]));
}
// out = sum(partial[0..nb)) (mode 0), or sqrt of its real part (mode 1)
__global__ __launch_bounds__(256) void dot_final_kernel(int nb, const dc* __restrict__ partial, int mode, dc* __restrict__ out) {
  __shared__ double sr[4], si[4];
  double ar = 0.0, ai = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) { ar += partial[i].re; ai += partial[i].im; }
  ar = wave_sum(ar); ai = wave_sum(ai);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sr[wave] = ar; si[wave] = ai; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double r = (sr[0] + sr[1]) + (sr[2] + sr[3]), i2 = (si[0] + si[1]) + (si[2] + si[3]);
    *out = mode == 0 ? dc_make(r, i2) : dc_make(__builtin_sqrt(r), 0.0);
  }
}
// y += s * alpha * x with alpha read from device memory (s = +1 / -1), or with a host scalar when alpha == nullptr
__global__ __launch_bounds__(256) void axpy_kernel(long long n, const dc* __restrict__ alpha, double sgn, double hre, double him,
                                                   const dc* __restrict__ x, dc* __restrict__ y) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double are = hre, aim = him;
  if (alpha) { are = sgn * alpha->re; aim = sgn * alpha->im; }
  const dc xv = x[i]; dc yv = y[i];
  yv.re += are * xv.re - aim * xv.im; yv.im += are * xv.im + aim * xv.re;
  y[i] = yv;
}
// out = a * x + b * y (host scalars; y may be nullptr when b == 0)
__global__ __launch_bounds__(256) void axpby_kernel(long long n, double are, double aim, const dc* __restrict__ x, double bre, double bim,
                                                    const dc* __restrict__ y, dc* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const dc xv = x[i];
  dc o = dc_make(are * xv.re - aim * xv.im, are * xv.im + aim * xv.re);
  if (y) { const dc yv = y[i]; o.re += bre * yv.re - bim * yv.im; o.im += bre * yv.im + bim * yv.re; }
  out[i] = o;
}

// ------------------------------------------------------------------ on-the-fly TBEM matvec, 13-point part
__constant__ double c_op_tri13[13][3];
int op_upload_tables(const double tri13_scaled[13][3]) {
  MA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_op_tri13), tri13_scaled, sizeof(double) * 39));
  return MA_OK;
}

#define MA_INV4PI 0.07957747154594767280

// One (i, j) pair with the un-subdivided 13-point rule: returns the Burton–Miller coefficient exactly as
// tbem_far_kernel forms it (same operation order), so that "dense A x" and "on-the-fly A x" differ by
// summation order only.
__device__ __forceinline__ dc pair_coeff_13(double d0x, double d0y, double d0z, double e1x, double e1y, double e1z, double e2x, double e2y,
                                            double e2z, double nyx, double nyy, double nyz, double nxx, double nxy, double nxz, double jw,
                                            int fbc, const BemPhys& ph, double k, double k2) {
  const double m = nxx * nyx + nxy * nyy + nxz * nyz;
  // as tbem_far_kernel: (y - x) . n_y is constant over the flat panel, (y - x) . n_x affine in (xi, eta)
  const double dny = d0x * nyx + d0y * nyy + d0z * nyz;
  const double dnx0 = d0x * nxx + d0y * nxy + d0z * nxz;
  const double e1nx = e1x * nxx + e1y * nxy + e1z * nxz, e2nx = e2x * nxx + e2y * nxy + e2z * nxz;
  double g_re = 0, g_im = 0, h_re = 0, h_im = 0, t_re = 0, t_im = 0, e_re = 0, e_im = 0;
#pragma unroll
  for (int q = 0; q < 13; ++q) {
    const double xi = c_op_tri13[q][0], eta = c_op_tri13[q][1], w4pi = c_op_tri13[q][2] * jw;
    const double dx = __builtin_fma(eta, e2x, __builtin_fma(xi, e1x, d0x));
    const double dy = __builtin_fma(eta, e2y, __builtin_fma(xi, e1y, d0y));
    const double dz = __builtin_fma(eta, e2z, __builtin_fma(xi, e1z, d0z));
    const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
    if (!(r2 >= 1e-30)) continue;
    double r, ri; sqrt_rsqrt(r2, r, ri);
    double sn, cs; sincos_bounded(k * r, sn, cs);
    const double gsc = w4pi * ri;
    const double gre = cs * gsc, gim = sn * gsc;
    const double bre = -(gre * ri) - gim * k, bim = gre * k - gim * ri;
    const double a = dny * ri;
    const double b = -(__builtin_fma(eta, e2nx, __builtin_fma(xi, e1nx, dnx0)) * ri);
    const double rq = a * b, ri2 = ri * ri;
    const double fr = (3.0 * ri2 - k2) * rq + m * ri2;
    const double fi = -(k * ri) * (3.0 * rq + m);
    g_re += gre; g_im += gim;
    h_re = __builtin_fma(bre, a, h_re); h_im = __builtin_fma(bim, a, h_im);
    t_re = __builtin_fma(bre, b, t_re); t_im = __builtin_fma(bim, b, t_im);
    e_re += gre * fr - gim * fi; e_im += gre * fi + gim * fr;
  }
  const double gt = ph.gamma * ph.tau;
  if (fbc == 0) {
    const double hr = h_re * ph.sign, hi = h_im * ph.sign;
    return dc_make(hr * gt + (e_re * ph.beta_re - e_im * ph.beta_im), hi * gt + (e_re * ph.beta_im + e_im * ph.beta_re));
  } else if (fbc == 1) {
    return dc_make(-(g_re * gt + (t_re * ph.beta_re - t_im * ph.beta_im)), -(g_im * gt + (t_re * ph.beta_im + t_im * ph.beta_re)));
  }
  return dc_make(0.0, 0.0);
}

// Transposed product y = A^T x of the streamed operator: lane = FIELD panel j (its geometry stays in registers), the
// collocation rows of the chunk go by on the scalar path -- the loop nest of tbem_far_kernel, accumulating instead of
// storing. grid.x: strips of 256 panels; grid.y: row chunks of [row0, row1). partial[chunk][j] = sum_i A13_ij x[dof_i]
__global__ __launch_bounds__(256) void tbem_matvec_t_kernel(BemGeom g, BemPhys ph, int row0, int row1, int chunk_rows, const dc* __restrict__ x,
                                                            dc* __restrict__ partial) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool valid = j < g.np;
  const int jj = valid ? j : g.np - 1;
  const double p0x = g.p0[0][jj], p0y = g.p0[1][jj], p0z = g.p0[2][jj];
  const double e1x = g.e1[0][jj], e1y = g.e1[1][jj], e1z = g.e1[2][jj];
  const double e2x = g.e2[0][jj], e2y = g.e2[1][jj], e2z = g.e2[2][jj];
  const double nyx = g.ny[0][jj], nyy = g.ny[1][jj], nyz = g.ny[2][jj];
  const double jw = g.jac[jj] * MA_INV4PI;
  const int fbc = g.bc_type[jj];
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k;
  const int i0 = row0 + blockIdx.y * chunk_rows, i1 = min(i0 + chunk_rows, row1);
  double yr = 0.0, yi = 0.0;
  for (int i = i0; i < i1; ++i) {
    const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];          // wave-uniform collocation row (scalar loads)
    const dc xi = x[g.dof[i]];
    const dc a = pair_coeff_13(p0x - cx, p0y - cy, p0z - cz, e1x, e1y, e1z, e2x, e2y, e2z, nyx, nyy, nyz, g.nx[0][i], g.nx[1][i], g.nx[2][i],
                               jw, fbc, ph, k, k2);
    yr += a.re * xi.re - a.im * xi.im; yi += a.re * xi.im + a.im * xi.re;
  }
  if (valid) partial[(size_t)blockIdx.y * g.np + j] = dc_make(yr, yi);
}

// y[dof_j] = sum_chunks partial[c][j] + diag_corr[j] x[dof_j] + sum over the near pairs (i, j) of column j of corr[q] x[dof_i];
// t_off / t_idx list the plan's pairs by column (built once per operator); rows outside [row0, row1) belong to other shards
__global__ __launch_bounds__(256) void tbem_matvec_t_finish_kernel(BemGeom g, int row0, int row1, int nchunks, const dc* __restrict__ partial,
                                                                  const long long* __restrict__ t_off, const int* __restrict__ t_idx,
                                                                  const int2* __restrict__ pairs, const dc* __restrict__ corr,
                                                                  const dc* __restrict__ diag_corr, const dc* __restrict__ x, dc* __restrict__ y) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= g.np) return;
  double yr = 0.0, yi = 0.0;
  for (int c = 0; c < nchunks; ++c) { const dc p = partial[(size_t)c * g.np + j]; yr += p.re; yi += p.im; }
  if (j >= row0 && j < row1) { const dc d = diag_corr[j], xv = x[g.dof[j]]; yr += d.re * xv.re - d.im * xv.im; yi += d.re * xv.im + d.im * xv.re; }
  for (long long t = t_off[j]; t < t_off[j + 1]; ++t) {
    const int q = t_idx[t];
    const int i = pairs[q].x;
    if (i < row0 || i >= row1) continue;
    const dc a = corr[q], xv = x[g.dof[i]];
    yr += a.re * xv.re - a.im * xv.im; yi += a.re * xv.im + a.im * xv.re;
  }
  y[g.dof[j]] = dc_make(yr, yi);
}

// y = A x with lane = FIELD panel j (geometry and x_j in registers) and the collocation rows on the scalar path -- the loop
// nest of the assembly kernel, 1/3 faster than lane = row (16 scalar loads per panel there, 7 per row here). The 64 lanes'
// products of a row are summed with DPP; a wavefront collects the sums of 64 consecutive rows in its lanes, the workgroup's
// four wavefronts are added through LDS, and partial[strip of 256 panels][row] goes out coalesced.
// grid.x: strips of 256 panels; grid.y: tiles of `rows_per_block` rows (a multiple of 64) of [row0, row1)
__global__ __launch_bounds__(256) void tbem_matvec_lp_kernel(BemGeom g, BemPhys ph, int row0, int row1, int rows_per_block, const dc* __restrict__ x,
                                                             dc* __restrict__ partial) {
  __shared__ double s_re[4][64], s_im[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int jj = j < g.np ? j : g.np - 1;
  const bool valid = j < g.np && !(g.nquad > 0 && g.ptype[jj] == 4);   // Quad4 columns: tbem_matvec_quad_kernel's strips
  const double p0x = g.p0[0][jj], p0y = g.p0[1][jj], p0z = g.p0[2][jj];
  const double e1x = g.e1[0][jj], e1y = g.e1[1][jj], e1z = g.e1[2][jj];
  const double e2x = g.e2[0][jj], e2y = g.e2[1][jj], e2z = g.e2[2][jj];
  const double nyx = g.ny[0][jj], nyy = g.ny[1][jj], nyz = g.ny[2][jj];
  const double jw = g.jac[jj] * MA_INV4PI;
  const int fbc = g.bc_type[jj];
  const dc xj = valid ? x[g.dof[jj]] : dc_make(0.0, 0.0);
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k;
  const int nr = row1 - row0;
  const int t0 = row0 + blockIdx.y * rows_per_block, t1 = min(t0 + rows_per_block, row1);
  for (int base = t0; base < t1; base += 64) {
    double acc_re = 0.0, acc_im = 0.0;                     // lane l: the wavefront's sum for row base + l
    const int iend = min(base + 64, t1);
    for (int i = base; i < iend; ++i) {
      const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];          // wave-uniform collocation row (scalar loads)
      const dc a = pair_coeff_13(p0x - cx, p0y - cy, p0z - cz, e1x, e1y, e1z, e2x, e2y, e2z, nyx, nyy, nyz, g.nx[0][i], g.nx[1][i], g.nx[2][i],
                                 jw, fbc, ph, k, k2);
      const double pr = wave_sum_lane63(a.re * xj.re - a.im * xj.im), pi = wave_sum_lane63(a.re * xj.im + a.im * xj.re);
      const int tl = i - base;                              // uniform
      const int rl = __builtin_amdgcn_readlane(__double2loint(pr), 63), rh = __builtin_amdgcn_readlane(__double2hiint(pr), 63);
      const int il = __builtin_amdgcn_readlane(__double2loint(pi), 63), ih = __builtin_amdgcn_readlane(__double2hiint(pi), 63);
      if (lane == tl) { acc_re = __hiloint2double(rh, rl); acc_im = __hiloint2double(ih, il); }
    }
    s_re[wave][lane] = acc_re; s_im[wave][lane] = acc_im;
    __syncthreads();
    if (wave == 0 && base + lane < iend)
      partial[(size_t)blockIdx.x * nr + (base + lane - row0)] = dc_make(s_re[0][lane] + s_re[1][lane] + s_re[2][lane] + s_re[3][lane],
                                                                        s_im[0][lane] + s_im[1][lane] + s_im[2][lane] + s_im[3][lane]);
    __syncthreads();
  }
}

// 13-point coefficient of listed pairs (to form corrections A_true - A_13): out[q] for pairs[q] = (i, j)
__global__ __launch_bounds__(256) void tbem_pairs13_kernel(BemGeom g, BemPhys ph, const int2* __restrict__ pairs, long long npairs, dc* __restrict__ out) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= npairs) return;
  const int i = pairs[q].x, j = pairs[q].y;
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k;
  out[q] = pair_coeff_13(g.p0[0][j] - g.c[0][i], g.p0[1][j] - g.c[1][i], g.p0[2][j] - g.c[2][i], g.e1[0][j], g.e1[1][j], g.e1[2][j],
                         g.e2[0][j], g.e2[1][j], g.e2[2][j], g.ny[0][j], g.ny[1][j], g.ny[2][j], g.nx[0][i], g.nx[1][i], g.nx[2][i],
                         g.jac[j] * MA_INV4PI, g.bc_type[j], ph, k, k2);
}

// y[dof_i] = sum_chunks partial[c][i] + diag_corr[i] x[dof_i] + sum_{near pairs of row i} corr[q] x[dof_j]
// pair_off[i]..pair_off[i+1] delimit row i's pairs (the plan's list is sorted by row).
__global__ __launch_bounds__(256) void tbem_matvec_finish_kernel(BemGeom g, int row0, int row1, int nchunks, const dc* __restrict__ partial,
                                                                const long long* __restrict__ pair_off, const int2* __restrict__ pairs,
                                                                const dc* __restrict__ corr, const dc* __restrict__ diag_corr,
                                                                const dc* __restrict__ x, dc* __restrict__ y) {
  const int i = row0 + blockIdx.x * 256 + threadIdx.x;
  if (i >= row1) return;
  const int nr = row1 - row0;
  double yr = 0.0, yi = 0.0;
  for (int c = 0; c < nchunks; ++c) { const dc p = partial[(size_t)c * nr + (i - row0)]; yr += p.re; yi += p.im; }
  { const dc d = diag_corr[i], xv = x[g.dof[i]]; yr += d.re * xv.re - d.im * xv.im; yi += d.re * xv.im + d.im * xv.re; }
  for (long long q = pair_off[i]; q < pair_off[i + 1]; ++q) {
    const dc a = corr[q], xv = x[g.dof[pairs[q].y]];
    yr += a.re * xv.re - a.im * xv.im; yi += a.re * xv.im + a.im * xv.re;
  }
  y[g.dof[i]] = dc_make(yr, yi);
}

// corr[q] = A_true[q] - A_13[q] in place (A_true arrives in `corr`, A_13 in `a13`)
__global__ __launch_bounds__(256) void sub_inplace_kernel(long long n, dc* __restrict__ corr, const dc* __restrict__ a13) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  corr[q] = dc_make(corr[q].re - a13[q].re, corr[q].im - a13[q].im);
}

// y = A^T x (CONJ: A^H x) for a dense row-major A: lane = column (coalesced along the rows of A), the 4 wavefronts of a
// workgroup take every 4th row of the workgroup's row chunk, partials[chunk][col] are summed by zgemv_t_finish in chunk
// order -- reproducible, unlike a scatter with atomics.
#define ZT_CHUNKS 16
template <bool CONJ>
__global__ __launch_bounds__(256) void zgemv_t_kernel(long long n, const dc* __restrict__ A, const dc* __restrict__ x, dc* __restrict__ partial) {
  __shared__ dc red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long col = (long long)blockIdx.x * 64 + lane;
  const long long rpc = (n + ZT_CHUNKS - 1) / ZT_CHUNKS;
  const long long r0 = (long long)blockIdx.y * rpc, r1 = min(n, r0 + rpc);
  double sr = 0.0, si = 0.0;
  if (col < n)
    for (long long r = r0 + wave; r < r1; r += 4) {
      const dc a = A[r * n + col]; const dc v = x[r];
      const double ai = CONJ ? -a.im : a.im;
      sr += a.re * v.re - ai * v.im; si += a.re * v.im + ai * v.re;
    }
  red[wave][lane] = dc_make(sr, si);
  __syncthreads();
  if (wave == 0 && col < n) {
    dc t = red[0][lane];
    for (int w = 1; w < 4; ++w) { t.re += red[w][lane].re; t.im += red[w][lane].im; }
    partial[(long long)blockIdx.y * n + col] = t;
  }
}
__global__ __launch_bounds__(256) void zgemv_t_finish_kernel(long long n, const dc* __restrict__ partial, dc* __restrict__ y) {
  const long long col = (long long)blockIdx.x * 256 + threadIdx.x;
  if (col >= n) return;
  dc t = partial[col];
  for (int c = 1; c < ZT_CHUNKS; ++c) { const dc p = partial[(long long)c * n + col]; t.re += p.re; t.im += p.im; }
  y[col] = t;
}
__global__ __launch_bounds__(256) void conj_kernel(long long n, const dc* in, dc* out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { const dc v = in[i]; out[i] = dc_make(v.re, -v.im); }
}

// DiagonalPreconditioner::from_diagonal (fmm_interface.rs:196-206): inv[slot(i)] = 1 / d_i, or 1 where |d_i| <= 1e-15. `d` is
// read with stride `ds` (n + 1: the diagonal of a dense row-major matrix); slot(i) = map[i] (panel -> dof) or i
__global__ __launch_bounds__(256) void diag_invert_kernel(long long n, const dc* __restrict__ d, long long ds, const int* __restrict__ map, dc* __restrict__ inv) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const dc v = d[i * ds];
  const double ns = v.re * v.re + v.im * v.im;
  inv[map ? map[i] : i] = hypot(v.re, v.im) > 1e-15 ? dc_make(v.re / ns, -v.im / ns) : dc_make(1.0, 0.0);
}
// z = a .* x (Preconditioner::apply of the diagonal preconditioner, fmm_interface.rs:208-212)
__global__ __launch_bounds__(256) void cmul_kernel(long long n, const dc* __restrict__ a, const dc* __restrict__ x, dc* __restrict__ z) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) z[i] = a[i] * x[i];
}

// ------------------------------------------------------------------ launchers
// partial: ZT_CHUNKS * n entries
int op_launch_zgemv_t(long long n, const c64* A, const c64* x, c64* partial, c64* y, bool conj, hipStream_t st) {
  if (n <= 0) return MA_OK;
  dim3 grid((unsigned)((n + 63) / 64), ZT_CHUNKS);
  if (conj) hipLaunchKernelGGL(zgemv_t_kernel<true>, grid, dim3(256), 0, st, n, reinterpret_cast<const dc*>(A), reinterpret_cast<const dc*>(x), reinterpret_cast<dc*>(partial));
  else hipLaunchKernelGGL(zgemv_t_kernel<false>, grid, dim3(256), 0, st, n, reinterpret_cast<const dc*>(A), reinterpret_cast<const dc*>(x), reinterpret_cast<dc*>(partial));
  hipLaunchKernelGGL(zgemv_t_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, reinterpret_cast<const dc*>(partial), reinterpret_cast<dc*>(y));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int op_zgemv_t_chunks() { return ZT_CHUNKS; }
int op_launch_diag_invert(long long n, const c64* d, long long ds, const int* map, c64* inv, hipStream_t st) {
  if (n <= 0) return MA_OK;
  hipLaunchKernelGGL(diag_invert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, reinterpret_cast<const dc*>(d), ds, map, reinterpret_cast<dc*>(inv));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int op_launch_cmul(long long n, const c64* a, const c64* x, c64* z, hipStream_t st) {
  if (n <= 0) return MA_OK;
  hipLaunchKernelGGL(cmul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, reinterpret_cast<const dc*>(a), reinterpret_cast<const dc*>(x), reinterpret_cast<dc*>(z));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int op_launch_conj(long long n, const c64* in, c64* out, hipStream_t st) {
  if (n <= 0) return MA_OK;
  hipLaunchKernelGGL(conj_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, reinterpret_cast<const dc*>(in), reinterpret_cast<dc*>(out));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int op_launch_zgemv(long long n, const c64* A, const c64* x, c64* y, hipStream_t st) {
  if (n <= 0) return MA_OK;
  hipLaunchKernelGGL(zgemv_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, n, reinterpret_cast<const dc*>(A), reinterpret_cast<const dc*>(x),
                     reinterpret_cast<dc*>(y));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
// One modified Gram-Schmidt step of GMRES (gmres.rs:178-196 / :344-362) as ONE launch: for i = 0..j: h_i = <v_i, w>, w -= h_i v_i;
// then |w|. The separate kernels (two per inner product, one per update: 3 (j + 1) + 2 dependent launches per Krylov step) were
// what a step cost on the sparse side. Here the workgroups keep their elements of w in registers, every inner product is reduced
// exactly as dot_partial_kernel + dot_final_kernel reduce it (same element-to-thread map, same trees: bit-identical h_i), and
// the workgroups exchange their partial sums through slots that start as a sentinel and are polled until both words have left it
// (the value is its own flag, as in csr_gs_flags_kernel). Grid = the dot kernels' grid (<= 256 workgroups, one per CU:
// co-resident); every wait is bounded (2 s) and an abandoned wait raises err[0], after which nobody waits.
constexpr unsigned long long MGS_SENTINEL = 0x7FFC0DE0DEADBEEFull;
constexpr int MGS_EMAX = 32;                                   // elements of w per thread (n <= 2^21 with 256 x 256 threads)
__device__ __forceinline__ unsigned long long mgs_word(double v) {
  const unsigned long long w = (unsigned long long)__double_as_longlong(v);
  return w == MGS_SENTINEL ? (w ^ 1ull) : w;
}
__global__ __launch_bounds__(256) void mgs_fill_kernel(long long nwords, unsigned long long* __restrict__ p) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < nwords) p[i] = MGS_SENTINEL;
}
template <int E>
__global__ __launch_bounds__(256) void gmres_mgs_kernel(long long n, const dc* __restrict__ V, int j, dc* __restrict__ w, unsigned long long* slots /* (j + 2) x gridDim.x x 2 */,
                                                        dc* __restrict__ scal_out /* h_0..h_j, (|w|, 0), then (1, 0) if a wait was abandoned else (0, 0) */, unsigned* err,
                                                        unsigned* gerr) {
  __shared__ double sr[4], si[4], tr[4], ti[4], hb[2][2];        // two sets of reduction slots and two h slots: a barrier less per phase
  const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const long long stride = (long long)nb * 256, i0 = (long long)b * 256 + tid;
  dc wv[E];
#pragma unroll
  for (int e = 0; e < E; ++e) { const long long idx = i0 + e * stride; wv[e] = idx < n ? w[idx] : dc_make(0.0, 0.0); }
  bool dead = false;
  for (int step = 0; step <= j + 1; ++step) {
    const bool last = step == j + 1;
    const dc* v = V + (long long)step * n;
    double ar = 0.0, ai = 0.0;
    dc vv[E];
    if (!last) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const long long idx = i0 + e * stride;
        if (idx < n) { const dc a = v[idx]; vv[e] = a; const dc bb = wv[e]; ar += a.re * bb.re + a.im * bb.im; ai += a.re * bb.im - a.im * bb.re; }
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) { const long long idx = i0 + e * stride; if (idx < n) { const dc a = wv[e]; ar += a.re * a.re + a.im * a.im; } }
    }
    ar = wave_sum(ar); ai = wave_sum(ai);
    if (lane == 0) { sr[wave] = ar; si[wave] = ai; }
    __syncthreads();
    unsigned long long* row = slots + (long long)step * nb * 2;
    if (tid == 0) {
      __hip_atomic_store(row + 2 * b, mgs_word((sr[0] + sr[1]) + (sr[2] + sr[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(row + 2 * b + 1, mgs_word((si[0] + si[1]) + (si[2] + si[3])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // dot_final_kernel's reduction over the nb partial sums, done by every workgroup (its own LDS slots: no barrier needed here)
    double pr = 0.0, pi = 0.0;
    if (tid < nb) {
      unsigned long long wa = __hip_atomic_load(row + 2 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), wb = __hip_atomic_load(row + 2 * tid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((wa == MGS_SENTINEL || wb == MGS_SENTINEL) && !dead) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned spins = 0;
        do {
          __builtin_amdgcn_s_sleep(1);
          wa = __hip_atomic_load(row + 2 * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); wb = __hip_atomic_load(row + 2 * tid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((++spins & 1023u) == 0u) {
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { dead = true; break; }
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
              __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (gerr) __hip_atomic_store(gerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              dead = true; break;
            }
          }
        } while (wa == MGS_SENTINEL || wb == MGS_SENTINEL);
      }
      pr += __longlong_as_double((long long)wa); pi += __longlong_as_double((long long)wb);
    }
    pr = wave_sum(pr); pi = wave_sum(pi);
    if (lane == 0) { tr[wave] = pr; ti[wave] = pi; }
    __syncthreads();
    const int hs = step & 1;
    if (tid == 0) {
      const double r = (tr[0] + tr[1]) + (tr[2] + tr[3]), i2 = (ti[0] + ti[1]) + (ti[2] + ti[3]);
      hb[hs][0] = last ? __builtin_sqrt(r) : r; hb[hs][1] = last ? 0.0 : i2;
      if (b == 0) scal_out[step] = dc_make(hb[hs][0], hb[hs][1]);
    }
    __syncthreads();
    if (!last) {                                                // w -= h v   (axpy_kernel with alpha = h, sgn = -1)
      const double are = -1.0 * hb[hs][0], aim = -1.0 * hb[hs][1];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const long long idx = i0 + e * stride;
        if (idx < n) { const dc xv = vv[e]; wv[e].re += are * xv.re - aim * xv.im; wv[e].im += are * xv.im + aim * xv.re; }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) { const long long idx = i0 + e * stride; if (idx < n) w[idx] = wv[e]; }
  // workgroup 0 waits for every workgroup's slots in every step: if any wait of this launch was abandoned, one of its threads knows
  if (b == 0) {
    const int any = __syncthreads_or(dead ? 1 : 0);
    if (tid == 0) scal_out[j + 2] = dc_make(any ? 1.0 : 0.0, 0.0);
  }
}
// returns MA_ERR_UNSUPPORTED when the vector is too long for the register-resident form, or when the grid cannot be co-resident on
// this device on its own (the caller then runs the separate kernels)
template <int E> static const void* mgs_fn() { return reinterpret_cast<const void*>(gmres_mgs_kernel<E>); }
int op_launch_gmres_mgs(long long n, const c64* V, int j, c64* w, void* slots, c64* scal_out, unsigned* err, hipStream_t st) {
  int nb = (int)((n + 255) / 256); if (nb > RED_BLOCKS) nb = RED_BLOCKS; if (nb < 1) nb = 1;
  const long long per = (n + (long long)nb * 256 - 1) / ((long long)nb * 256);
  if (per > MGS_EMAX) return MA_ERR_UNSUPPORTED;
  const int ei = per <= 1 ? 0 : per <= 2 ? 1 : per <= 4 ? 2 : per <= 8 ? 3 : per <= 16 ? 4 : 5;
  // Residency: the workgroups of this launch wait for one another. The grid is the dot kernels' grid (<= 256: the partial sums are
  // then theirs bit for bit), so it must fit the device at the occupancy the runtime reports for this instantiation (registers:
  // E = 32 keeps 2 x 32 complex values per lane) -- otherwise the separate kernels run -- and the launch goes through the admission
  // window of the spinning kernels (lu_kernels.hip, "Residency"), which also keeps it apart from LU panel grids and flag-driven
  // sweeps of other streams and host threads when they do not all fit.
  static int regs_of[6] = {}, occ_of[6] = {}; static int ncu_of[16] = {};
  int dev = 0; MA_HIP(hipGetDevice(&dev));
  const void* fn = ei == 0 ? mgs_fn<1>() : ei == 1 ? mgs_fn<2>() : ei == 2 ? mgs_fn<4>() : ei == 3 ? mgs_fn<8>() : ei == 4 ? mgs_fn<16>() : mgs_fn<32>();
  if (regs_of[ei] == 0) {
    hipFuncAttributes fa; MA_HIP(hipFuncGetAttributes(&fa, fn));
    int occ = 0; MA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 256, 0));
    occ_of[ei] = occ; regs_of[ei] = fa.numRegs > 0 ? fa.numRegs : 256;
  }
  if (dev >= 0 && dev < 16 && ncu_of[dev] == 0) { hipDeviceProp_t prop; MA_HIP(hipGetDeviceProperties(&prop, dev)); ncu_of[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
  const int ncu = (dev >= 0 && dev < 16) ? ncu_of[dev] : 256;
  if (occ_of[ei] < 1 || (long long)nb > (long long)std::min(occ_of[ei], lu_panel_slots_per_cu(1, regs_of[ei])) * ncu) return MA_ERR_UNSUPPORTED;
  const long long nwords = (long long)(j + 2) * nb * 2;
  hipLaunchKernelGGL(mgs_fill_kernel, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, nwords, reinterpret_cast<unsigned long long*>(slots));
  const dc* Vd = reinterpret_cast<const dc*>(V); dc* wd = reinterpret_cast<dc*>(w); dc* sd = reinterpret_cast<dc*>(scal_out);
  unsigned long long* sl = reinterpret_cast<unsigned long long*>(slots);
  unsigned* gerr = spin_error_word();
  SpinLaunch guard;
  { const int arc = guard.admit(st, nb, 0, regs_of[ei], ncu); if (arc) return arc; }
  if (ei == 0) hipLaunchKernelGGL(gmres_mgs_kernel<1>, dim3(nb), dim3(256), 0, st, n, Vd, j, wd, sl, sd, err, gerr);
  else if (ei == 1) hipLaunchKernelGGL(gmres_mgs_kernel<2>, dim3(nb), dim3(256), 0, st, n, Vd, j, wd, sl, sd, err, gerr);
  else if (ei == 2) hipLaunchKernelGGL(gmres_mgs_kernel<4>, dim3(nb), dim3(256), 0, st, n, Vd, j, wd, sl, sd, err, gerr);
  else if (ei == 3) hipLaunchKernelGGL(gmres_mgs_kernel<8>, dim3(nb), dim3(256), 0, st, n, Vd, j, wd, sl, sd, err, gerr);
  else if (ei == 4) hipLaunchKernelGGL(gmres_mgs_kernel<16>, dim3(nb), dim3(256), 0, st, n, Vd, j, wd, sl, sd, err, gerr);
  else hipLaunchKernelGGL(gmres_mgs_kernel<32>, dim3(nb), dim3(256), 0, st, n, Vd, j, wd, sl, sd, err, gerr);
  MA_HIP(hipGetLastError());
  return guard.commit();
}
int op_mgs_slot_bytes(int m) { return (int)sizeof(unsigned long long) * 2 * RED_BLOCKS * (m + 2); }
int op_launch_dot(long long n, const c64* x, const c64* y, int mode, c64* partial, c64* out, hipStream_t st) {
  int nb = (int)((n + 255) / 256); if (nb > RED_BLOCKS) nb = RED_BLOCKS; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(256), 0, st, n, reinterpret_cast<const dc*>(x), reinterpret_cast<const dc*>(y), mode,
                     reinterpret_cast<dc*>(partial));
  hipLaunchKernelGGL(dot_final_kernel, dim3(1), dim3(256), 0, st, nb, reinterpret_cast<const dc*>(partial), mode, reinterpret_cast<dc*>(out));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
// `count` inner products <v_i, y> against one y (the classical Gram-Schmidt row of the pipelined GMRES) in two launches instead of
// two per product: blockIdx.y selects the vector, every (vector, block) pair reduces exactly as dot_partial_kernel / dot_final_kernel
// do, so each h_i is the separate kernels' bit for bit. partial: count x RED_BLOCKS entries.
__global__ __launch_bounds__(256) void multi_dot_partial_kernel(long long n, const dc* __restrict__ V, const dc* __restrict__ y, dc* __restrict__ partial) {
  __shared__ double sr[4], si[4];
  const dc* x = V + (long long)blockIdx.y * n;
  double ar = 0.0, ai = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const dc a = x[i]; const dc b = y[i];
    ar += a.re * b.re + a.im * b.im; ai += a.re * b.im - a.im * b.re;
  }
  ar = wave_sum(ar); ai = wave_sum(ai);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sr[wave] = ar; si[wave] = ai; }
  __syncthreads();
  if (threadIdx.x == 0) partial[(long long)blockIdx.y * RED_BLOCKS + blockIdx.x] = dc_make((sr[0] + sr[1]) + (sr[2] + sr[3]), (si[0] + si[1]) + (si[2] + si[3]));
}
__global__ __launch_bounds__(256) void multi_dot_final_kernel(int nb, const dc* __restrict__ partial, dc* __restrict__ out) {
  __shared__ double sr[4], si[4];
  const dc* p = partial + (long long)blockIdx.x * RED_BLOCKS;
  double ar = 0.0, ai = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) { ar += p[i].re; ai += p[i].im; }
  ar = wave_sum(ar); ai = wave_sum(ai);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { sr[wave] = ar; si[wave] = ai; }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = dc_make((sr[0] + sr[1]) + (sr[2] + sr[3]), (si[0] + si[1]) + (si[2] + si[3]));
}
int op_launch_multi_dot(long long n, const c64* V, int count, const c64* y, c64* partial /* count x 256 */, c64* out, hipStream_t st) {
  if (count <= 0) return MA_OK;
  int nb = (int)((n + 255) / 256); if (nb > RED_BLOCKS) nb = RED_BLOCKS; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(multi_dot_partial_kernel, dim3(nb, count), dim3(256), 0, st, n, reinterpret_cast<const dc*>(V), reinterpret_cast<const dc*>(y), reinterpret_cast<dc*>(partial));
  hipLaunchKernelGGL(multi_dot_final_kernel, dim3(count), dim3(256), 0, st, nb, reinterpret_cast<const dc*>(partial), reinterpret_cast<dc*>(out));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
// a -= sum_i h_i V_i and (when Z is given) b -= sum_i h_i Z_i, the updates applied in ascending i per element as the separate axpy
// launches apply them (bit-identical), in one launch
__global__ __launch_bounds__(256) void multi_axpy_kernel(long long n, int count, const dc* __restrict__ h, const dc* __restrict__ V, dc* __restrict__ a,
                                                         const dc* __restrict__ Z, dc* __restrict__ b) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  dc av = a[idx], bv = Z ? b[idx] : dc_make(0.0, 0.0);
  for (int i = 0; i < count; ++i) {
    const double are = -1.0 * h[i].re, aim = -1.0 * h[i].im;
    const dc xv = V[(long long)i * n + idx];
    av.re += are * xv.re - aim * xv.im; av.im += are * xv.im + aim * xv.re;
    if (Z) { const dc zv = Z[(long long)i * n + idx]; bv.re += are * zv.re - aim * zv.im; bv.im += are * zv.im + aim * zv.re; }
  }
  a[idx] = av;
  if (Z) b[idx] = bv;
}
int op_launch_multi_axpy(long long n, int count, const c64* h_dev, const c64* V, c64* a, const c64* Z, c64* b, hipStream_t st) {
  if (n <= 0 || count <= 0) return MA_OK;
  hipLaunchKernelGGL(multi_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, count, reinterpret_cast<const dc*>(h_dev), reinterpret_cast<const dc*>(V),
                     reinterpret_cast<dc*>(a), reinterpret_cast<const dc*>(Z), reinterpret_cast<dc*>(b));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int op_launch_axpy_dev(long long n, const c64* alpha_dev, double sgn, const c64* x, c64* y, hipStream_t st) {
  if (n <= 0) return MA_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, reinterpret_cast<const dc*>(alpha_dev), sgn, 0.0, 0.0,
                     reinterpret_cast<const dc*>(x), reinterpret_cast<dc*>(y));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int op_launch_axpy_host(long long n, double are, double aim, const c64* x, c64* y, hipStream_t st) {
  if (n <= 0) return MA_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, (const dc*)nullptr, 1.0, are, aim,
                     reinterpret_cast<const dc*>(x), reinterpret_cast<dc*>(y));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int op_launch_axpby(long long n, double are, double aim, const c64* x, double bre, double bim, const c64* y, c64* out, hipStream_t st) {
  if (n <= 0) return MA_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, are, aim, reinterpret_cast<const dc*>(x), bre, bim,
                     reinterpret_cast<const dc*>(y), reinterpret_cast<dc*>(out));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int op_tbem_matvec_strips(int np) { return (np + 255) / 256; }
// partial: (op_tbem_matvec_strips(np) + bem_quad_strips(g)) x (row1 - row0) entries
int op_launch_tbem_matvec(const BemGeom& g, const BemPhys& ph, int row0, int row1, int nchunks, const c64* x, c64* partial,
                          const long long* pair_off, const int2* pairs, const c64* corr, const c64* diag_corr, c64* y, hipStream_t st) {
  (void)nchunks;
  const int nr = row1 - row0;
  if (nr <= 0) return MA_OK;
  const int strips = op_tbem_matvec_strips(g.np);
  int rpb = 1024;                                          // rows per workgroup: enough workgroups to fill the chip several times over
  while (rpb > 64 && (long long)strips * ((nr + rpb - 1) / rpb) < 2048) rpb >>= 1;
  dim3 grid(strips, (nr + rpb - 1) / rpb), block(256);
  hipLaunchKernelGGL(tbem_matvec_lp_kernel, grid, block, 0, st, g, ph, row0, row1, rpb, reinterpret_cast<const dc*>(x), reinterpret_cast<dc*>(partial));
  // Quad4 columns: their own strips behind the Tri3 ones (bem_kernels.hip; nothing on a Tri3 mesh)
  const int qstrips = bem_quad_strips(g);
  int rc = bem_launch_quad_matvec(g, ph, row0, row1, rpb, x, partial + (size_t)strips * (size_t)nr, st);
  if (rc) return rc;
  hipLaunchKernelGGL(tbem_matvec_finish_kernel, dim3((nr + 255) / 256), block, 0, st, g, row0, row1, strips + qstrips, reinterpret_cast<const dc*>(partial),
                     pair_off, pairs, reinterpret_cast<const dc*>(corr), reinterpret_cast<const dc*>(diag_corr), reinterpret_cast<const dc*>(x),
                     reinterpret_cast<dc*>(y));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int op_launch_tbem_matvec_t(const BemGeom& g, const BemPhys& ph, int row0, int row1, int nchunks, const c64* x, c64* partial,
                            const long long* t_off, const int* t_idx, const int2* pairs, const c64* corr, const c64* diag_corr, c64* y, hipStream_t st) {
  const int nr = row1 - row0;
  if (nr <= 0 || g.np <= 0) return MA_OK;
  const int chunk_rows = (nr + nchunks - 1) / nchunks;
  dim3 grid((g.np + 255) / 256, nchunks), block(256);
  hipLaunchKernelGGL(tbem_matvec_t_kernel, grid, block, 0, st, g, ph, row0, row1, chunk_rows, reinterpret_cast<const dc*>(x), reinterpret_cast<dc*>(partial));
  int rc = bem_launch_quad_matvec_t(g, ph, row0, row1, nchunks, chunk_rows, x, partial, st);      // overwrites the Quad4 panels' slots
  if (rc) return rc;
  hipLaunchKernelGGL(tbem_matvec_t_finish_kernel, dim3((g.np + 255) / 256), block, 0, st, g, row0, row1, nchunks, reinterpret_cast<const dc*>(partial),
                     t_off, t_idx, pairs, reinterpret_cast<const dc*>(corr), reinterpret_cast<const dc*>(diag_corr), reinterpret_cast<const dc*>(x),
                     reinterpret_cast<dc*>(y));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int op_launch_pairs13(const BemGeom& g, const BemPhys& ph, const int2* pairs, long long npairs, c64* out, hipStream_t st) {
  if (npairs <= 0) return MA_OK;
  hipLaunchKernelGGL(tbem_pairs13_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, g, ph, pairs, npairs, reinterpret_cast<dc*>(out));
  MA_HIP(hipGetLastError());
  return bem_launch_quad_pairs_far(g, ph, pairs, npairs, out, st);       // pairs with a Quad4 field panel: what tbem_matvec_quad_kernel streams
}
int op_launch_sub_inplace(long long n, c64* corr, const c64* a13, hipStream_t st) {
  if (n <= 0) return MA_OK;
  hipLaunchKernelGGL(sub_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, reinterpret_cast<dc*>(corr), reinterpret_cast<const dc*>(a13));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

}  // namespace ma
