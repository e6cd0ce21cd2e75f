// field_kernels.hip — the two remaining pairwise sweeps next to the assembly:
//   room_matrix_kernel      point-collocation room-acoustics matrix, build_bem_matrix_parallel
//                           (math-bem/src/room_acoustics/solver.rs:448-493, kernel :28-35): ONE kernel evaluation
//                           per pair, 16 B written per pair -> this assembly IS HBM-write-bound.
//   scattered_field_kernel  field evaluation compute_scattered_field (math-bem/src/core/postprocess/pressure.rs:
//                           81-258): p(x) = sum_j p_j int dG/dn_y - v_j int G with the 7-point rule; lane =
//                           evaluation point, panels stream past on the scalar path, column chunks reduced
//                           deterministically.
#include "bem_kernels.hpp"
#include "ma_device_math.hpp"
#include "ma_tables.h"
#include <vector>

namespace ma {

#define MA_PI 3.14159265358979323846

// lane = column j (coalesced 16-B stores along a row); `rows` rows per workgroup, row data wave-uniform
__global__ __launch_bounds__(256) void room_matrix_kernel(int n, const double* __restrict__ c, const double* __restrict__ nr, const double* __restrict__ ar,
                                                          double k, dc* __restrict__ A, int rows) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool valid = j < n;
  const int jj = valid ? j : n - 1;
  const double cjx = c[3 * jj], cjy = c[3 * jj + 1], cjz = c[3 * jj + 2], aj = ar[jj];
  const int i0 = blockIdx.y * rows, i1 = min(i0 + rows, n);
  for (int i = i0; i < i1; ++i) {
    const double dx = c[3 * i] - cjx, dy = c[3 * i + 1] - cjy, dz = c[3 * i + 2] - cjz;
    const double r2 = dx * dx + dy * dy + dz * dz;
    dc v = dc_make(0.0, 0.0);
    if (i == jj) v = dc_make(0.0, -k / (2.0 * MA_PI) * aj);                    // solver.rs:470
    else if (r2 >= 1e-20) {                                                     // r < 1e-10 -> 0 (solver.rs:29)
      double r, ri; sqrt_rsqrt(r2, r, ri);
      double sn, cs; sincos_fast(k * r, sn, cs);
      const double cosang = (dx * nr[3 * i] + dy * nr[3 * i + 1] + dz * nr[3 * i + 2]) * ri;   // cos from n_i (solver.rs:473-478)
      // (i k r - 1) e^{ikr} / (4 pi r^2) * cos * A_j
      const double kr = k * r;
      const double fre = -cs - kr * sn, fim = kr * cs - sn;
      const double s = cosang * aj * ri * ri / (4.0 * MA_PI);
      v = dc_make(fre * s, fim * s);
    }
    if (valid) A[(size_t)i * n + j] = v;
  }
}

__constant__ double c_tri7[7][3];

// partial[chunk][m] = sum over the chunk's panels; lane = evaluation point
__global__ __launch_bounds__(256) void scattered_field_kernel(BemGeom g, int n_eval, const double* __restrict__ ep, const dc* __restrict__ ps,
                                                              const dc* __restrict__ vs, double wavruim, int chunk_cols, dc* __restrict__ partial) {
  const int mI = blockIdx.x * 256 + threadIdx.x;
  const bool valid = mI < n_eval;
  const int mm = valid ? mI : n_eval - 1;
  const double x = ep[3 * mm], y = ep[3 * mm + 1], z = ep[3 * mm + 2];
  const int j0 = blockIdx.y * chunk_cols, j1 = min(j0 + chunk_cols, g.np);
  double ar = 0.0, ai = 0.0;
  for (int j = j0; j < j1; ++j) {
    const double jac = g.jac[j];
    if (jac < 1e-15) continue;                                                  // pressure.rs:215-217
    const double d0x = g.p0[0][j] - x, d0y = g.p0[1][j] - y, d0z = g.p0[2][j] - z;
    const double e1x = g.e1[0][j], e1y = g.e1[1][j], e1z = g.e1[2][j], e2x = g.e2[0][j], e2y = g.e2[1][j], e2z = g.e2[2][j];
    const double nyx = g.ny[0][j], nyy = g.ny[1][j], nyz = g.ny[2][j];
    const dc p = ps[j];                                                         // surface values are indexed by boundary-element order (pressure.rs:111-114)
    const dc v = vs ? vs[j] : dc_make(0.0, 0.0);
    const bool has_v = __builtin_sqrt(v.re * v.re + v.im * v.im) > 1e-15;
    double sre = 0.0, sim = 0.0, gre_s = 0.0, gim_s = 0.0;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      const double xi = c_tri7[q][0], eta = c_tri7[q][1], w = c_tri7[q][2];
      const double dx = __builtin_fma(eta, e2x, __builtin_fma(xi, e1x, d0x));
      const double dy = __builtin_fma(eta, e2y, __builtin_fma(xi, e1y, d0y));
      const double dz = __builtin_fma(eta, e2z, __builtin_fma(xi, e1z, d0z));
      const double r2 = dx * dx + dy * dy + dz * dz;
      if (!(r2 >= 1e-30)) continue;
      double r, ri; sqrt_rsqrt(r2, r, ri);
      double sn, cs; sincos_fast(wavruim * r, sn, cs);
      const double gs = jac * w * ri / (4.0 * MA_PI);
      const double gre = cs * gs, gim = sn * gs;                               // G * vjacwe
      const double bre = -(gre * ri) - gim * wavruim, bim = gre * wavruim - gim * ri;
      const double drdn = (dx * nyx + dy * nyy + dz * nyz) * ri;
      sre += bre * drdn; sim += bim * drdn;                                    // dG/dn_y * vjacwe
      gre_s += gre; gim_s += gim;
    }
    ar += p.re * sre - p.im * sim; ai += p.re * sim + p.im * sre;
    if (has_v) { ar -= v.re * gre_s - v.im * gim_s; ai -= v.re * gim_s + v.im * gre_s; }
  }
  if (valid) partial[(size_t)blockIdx.y * n_eval + mI] = dc_make(ar, ai);
}

__global__ __launch_bounds__(256) void sum_chunks_kernel(int n, int nchunks, const dc* __restrict__ partial, dc* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double r = 0.0, im = 0.0;
  for (int c = 0; c < nchunks; ++c) { const dc p = partial[(size_t)c * n + i]; r += p.re; im += p.im; }
  out[i] = dc_make(r, im);
}

}  // namespace ma

using namespace ma;

extern "C" {

// build_bem_matrix_parallel (solver.rs:448-493) from the element centres / normals / areas (:38-122), device output
int ma_room_build_matrix_dev(int32_t n, const void* d_center, const void* d_normal, const void* d_area, double k, void* d_A, void* stream) {
  MA_REQUIRE(n > 0 && d_center && d_normal && d_area && d_A, MA_ERR_INVALID, "bad argument");
  const int rows = 32;
  dim3 grid((n + 255) / 256, (n + rows - 1) / rows), block(256);
  hipLaunchKernelGGL(room_matrix_kernel, grid, block, 0, (hipStream_t)stream, n, (const double*)d_center, (const double*)d_normal, (const double*)d_area, k,
                     reinterpret_cast<dc*>(d_A), rows);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int ma_room_build_matrix(int32_t n, const double* center, const double* normal, const double* area, double k, ma_c64* A) {
  MA_REQUIRE(n > 0 && center && normal && area && A, MA_ERR_INVALID, "bad argument");
  int dev = 0; if (const char* s = getenv("MA_DEVICE")) dev = atoi(s);
  int rc = use_device(dev); if (rc) return rc;
  double *dc_ = nullptr, *dn = nullptr, *da = nullptr; c64* dA = nullptr;
  hipError_t e = hipMalloc(&dc_, sizeof(double) * 3 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc(&dn, sizeof(double) * 3 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc(&da, sizeof(double) * (size_t)n);
  if (e == hipSuccess) e = hipMalloc(&dA, sizeof(c64) * (size_t)n * (size_t)n);
  if (e == hipSuccess) e = hipMemcpy(dc_, center, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dn, normal, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(da, area, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("room matrix buffers: %s", hipGetErrorString(e)); rc = MA_ERR_NOMEM; }
  if (!rc) rc = ma_room_build_matrix_dev(n, dc_, dn, da, k, dA, nullptr);
  if (!rc) { e = hipMemcpy(A, dA, sizeof(c64) * (size_t)n * (size_t)n, hipMemcpyDeviceToHost); if (e != hipSuccess) { set_error("copy back: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; } }
  void* p[] = {dc_, dn, da, dA}; for (void* q : p) if (q) (void)hipFree(q);
  return rc;
}

// compute_scattered_field (pressure.rs:81-137): eval_points n_eval x 3, surface_pressure / surface_velocity one value per
// boundary element in plan order (velocity may be NULL), out n_eval.
int ma_bem_plan_scattered_field(ma_bem_plan_t* P, const ma_physics_t* physics, int32_t n_eval, const double* eval_points,
                                const ma_c64* surface_pressure, const ma_c64* surface_velocity, ma_c64* out) {
  MA_REQUIRE(P && physics && n_eval > 0 && eval_points && surface_pressure && out, MA_ERR_INVALID, "bad argument");
  // Quad4 panels: the reference integrates the triangle of their first three nodes with the 7-point rule
  // (pressure.rs:166-198); the plan's Tri3 arrays (p0, e1, e2, n_y, jac) of a quad are exactly that triangle.
  MA_HIP(hipSetDevice(P->device));
  double t7[7][3];
  for (int q = 0; q < 7; ++q) { t7[q][0] = mat_tri7[q][0]; t7[q][1] = mat_tri7[q][1]; t7[q][2] = mat_tri7[q][2] * 0.5; }
  MA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tri7), t7, sizeof(t7)));
  const int strips = (n_eval + 255) / 256;
  int nch = (1024 + strips - 1) / strips; if (nch < 1) nch = 1; if (nch > 128) nch = 128; if (nch > P->np) nch = P->np;
  const int chunk_cols = (P->np + nch - 1) / nch;
  double* dep = nullptr; c64 *dps = nullptr, *dvs = nullptr, *dpart = nullptr, *dout = nullptr;
  hipError_t e = hipMalloc(&dep, sizeof(double) * 3 * (size_t)n_eval);
  if (e == hipSuccess) e = hipMalloc(&dps, sizeof(c64) * (size_t)P->np);
  if (e == hipSuccess && surface_velocity) e = hipMalloc(&dvs, sizeof(c64) * (size_t)P->np);
  if (e == hipSuccess) e = hipMalloc(&dpart, sizeof(c64) * (size_t)nch * (size_t)n_eval);
  if (e == hipSuccess) e = hipMalloc(&dout, sizeof(c64) * (size_t)n_eval);
  if (e == hipSuccess) e = hipMemcpy(dep, eval_points, sizeof(double) * 3 * (size_t)n_eval, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dps, surface_pressure, sizeof(c64) * (size_t)P->np, hipMemcpyHostToDevice);
  if (e == hipSuccess && surface_velocity) e = hipMemcpy(dvs, surface_velocity, sizeof(c64) * (size_t)P->np, hipMemcpyHostToDevice);
  int rc = MA_OK;
  if (e != hipSuccess) { set_error("field evaluation buffers: %s", hipGetErrorString(e)); rc = MA_ERR_NOMEM; }
  if (!rc) {
    dim3 grid(strips, nch), block(256);
    hipLaunchKernelGGL(scattered_field_kernel, grid, block, 0, nullptr, P->geom, n_eval, dep, reinterpret_cast<const dc*>(dps), reinterpret_cast<const dc*>(dvs),
                       physics->wave_number * physics->harmonic_factor, chunk_cols, reinterpret_cast<dc*>(dpart));
    hipLaunchKernelGGL(sum_chunks_kernel, dim3(strips), block, 0, nullptr, n_eval, nch, reinterpret_cast<const dc*>(dpart), reinterpret_cast<dc*>(dout));
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(c64) * (size_t)n_eval, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("field evaluation failed: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; }
  }
  void* p[] = {dep, dps, dvs, dpart, dout}; for (void* q : p) if (q) (void)hipFree(q);
  return rc;
}

}  // extern "C"
