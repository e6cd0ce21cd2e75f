// room_kernels.hip — the rest of the room-acoustics path around build_bem_matrix_parallel (field_kernels.hip):
//   ma_room_element_data            element_center_and_normal / element_area / element_characteristic_length
//                                   (math-bem/src/room_acoustics/solver.rs:38-122, 600-611) -- O(N) host arithmetic
//   ma_room_build_matrix_adaptive   build_bem_matrix_adaptive (:500-597): point collocation for far pairs, and for near
//                                   pairs (r < 2 (l_i + l_j) or i == j) the double-layer part of
//                                   singular_integration_with_params on the first three nodes of element j
//   ma_room_incident_derivative     calculate_incident_field_derivative_parallel (:638-678)
//   ma_room_field_pressure          calculate_field_pressure_bem_parallel (:687-748)
#include "ma_common.hpp"
#include "ma_device_math.hpp"
#include "ma_tables.h"
#include <cmath>
#include <vector>

namespace ma {

#define MA_PI 3.14159265358979323846
#define MA_INV4PI 0.07957747154594767280

__constant__ double c_rgl_x[94];
__constant__ double c_rgl_w[94];
__constant__ int c_rgl_index[21][2];
__device__ __constant__ double c_rcsi6[6] = {0.0, 1.0, 0.0, 0.5, 0.5, 0.0};
__device__ __constant__ double c_reta6[6] = {0.0, 0.0, 1.0, 0.0, 0.5, 0.5};

struct RoomGeom {
  int n;
  const double* c;      // centres [n][3]
  const double* nr;     // unit normals [n][3]
  const double* ar;     // areas
  const double* cl;     // characteristic lengths
  const double* tri;    // first three nodes [n][9]
};

// (i k r - 1) e^{ikr} / (4 pi r^2) * cos, zero below r = 1e-10 (greens_function_derivative, solver.rs:28-35)
__device__ __forceinline__ dc room_dgdn(double dx, double dy, double dz, double nx, double ny, double nz, double k) {
  const double r2 = dx * dx + dy * dy + dz * dz;
  if (!(r2 >= 1e-20)) return dc_make(0.0, 0.0);
  double r, ri; sqrt_rsqrt(r2, r, ri);
  double sn, cs; sincos_fast(k * r, sn, cs);
  const double kr = k * r;
  const double s = ((dx * nx + dy * ny + dz * nz) * ri) * ri * ri * MA_INV4PI;
  return dc_make((-cs - kr * sn) * s, (kr * cs - sn) * s);
}

// One wavefront per collocation row. Far pairs: lane = column, point collocation. Near pairs of the row are then taken
// one at a time by the whole wavefront: lanes = the collapsed-square points of the 3 x nsec2 sub-triangles
// (singular.rs:257-354), only the double-layer sum is kept (row[j] = result.dg_dn_integral, solver.rs:571).
__global__ __launch_bounds__(256) void room_adaptive_kernel(RoomGeom g, double k, int use_adaptive, dc* __restrict__ A) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= g.n) return;
  const double cx = g.c[3 * i], cy = g.c[3 * i + 1], cz = g.c[3 * i + 2];
  const double nxx = g.nr[3 * i], nxy = g.nr[3 * i + 1], nxz = g.nr[3 * i + 2];
  const double cli = g.cl[i];
  for (int j0 = 0; j0 < g.n; j0 += 64) {
    const int j = j0 + lane;
    bool near = false;
    if (j < g.n) {
      const double dx = cx - g.c[3 * j], dy = cy - g.c[3 * j + 1], dz = cz - g.c[3 * j + 2];
      const double r = __builtin_sqrt(dx * dx + dy * dy + dz * dz);
      near = use_adaptive && (r < 2.0 * (cli + g.cl[j]) || i == j);
      if (!near) {
        dc v;
        if (i == j) v = dc_make(0.0, -k / (2.0 * MA_PI) * g.ar[j]);                       // solver.rs:575
        else { v = room_dgdn(dx, dy, dz, nxx, nxy, nxz, k); v.re *= g.ar[j]; v.im *= g.ar[j]; }
        A[(size_t)i * g.n + j] = v;
      }
    }
    unsigned long long mask = __ballot(near);
    while (mask) {
      const int jj = j0 + __builtin_ctzll(mask);
      mask &= mask - 1;
      const double* P = g.tri + 9 * (size_t)jj;
      const double e1x = P[3] - P[0], e1y = P[4] - P[1], e1z = P[5] - P[2];
      const double e2x = P[6] - P[0], e2y = P[7] - P[1], e2z = P[8] - P[2];
      double nyx = e1y * e2z - e1z * e2y, nyy = e1z * e2x - e1x * e2z, nyz = e1x * e2y - e1y * e2x;
      const double jac = __builtin_sqrt(nyx * nyx + nyy * nyy + nyz * nyz);
      const double ij = jac > 1e-15 ? 1.0 / jac : 0.0;
      nyx *= ij; nyy *= ij; nyz *= ij;
      const double ka = k * g.cl[jj];
      int ngausin, nsec2;
      if (ka < 0.3)      { ngausin = 4; nsec2 = 2; }
      else if (ka < 1.0) { ngausin = 5; nsec2 = 2; }
      else if (ka < 2.0) { ngausin = 6; nsec2 = 3; }
      else               { ngausin = 7; nsec2 = 4; }
      const int so = c_rgl_index[ngausin][0], ns = c_rgl_index[ngausin][1];
      const int per_edge = nsec2 * ns * ns;
      double hre = 0.0, him = 0.0;
      for (int t = lane; t < 3 * per_edge; t += 64) {
        const int ieg = t / per_edge, v2 = t - ieg * per_edge;
        const int ig1 = (ieg + 1) % 3, ig2 = ieg + 3;
        const int isec = v2 / (ns * ns);
        const int q = v2 - isec * ns * ns;
        const int ii = q / ns, kk = q - ii * ns;
        const double aresub = 1.0 / 24.0 / (double)nsec2;
        double ss1, ss2, ts1, ts2;
        if (isec == 0) { ss1 = c_rcsi6[ieg]; ss2 = c_rcsi6[ig2]; ts1 = c_reta6[ieg]; ts2 = c_reta6[ig2]; }
        else           { ss1 = c_rcsi6[ig2]; ss2 = c_rcsi6[ig1]; ts1 = c_reta6[ig2]; ts2 = c_reta6[ig1]; }
        const double sga = c_rgl_x[so + ii], tga = c_rgl_x[so + kk];
        const double wei = c_rgl_w[so + ii] * c_rgl_w[so + kk];
        const double sgg = 0.5 * (1.0 - sga) * (1.0 / 3.0) + 0.25 * (1.0 + sga) * ((1.0 - tga) * ss1 + (1.0 + tga) * ss2);
        const double tgg = 0.5 * (1.0 - sga) * (1.0 / 3.0) + 0.25 * (1.0 + sga) * ((1.0 - tga) * ts1 + (1.0 + tga) * ts2);
        const double n0 = 1.0 - sgg - tgg;
        const double dx = (n0 * P[0] + sgg * P[3] + tgg * P[6]) - cx;
        const double dy = (n0 * P[1] + sgg * P[4] + tgg * P[7]) - cy;
        const double dz = (n0 * P[2] + sgg * P[5] + tgg * P[8]) - cz;
        const double r2 = dx * dx + dy * dy + dz * dz;
        if (r2 >= 1e-30) {
          double r, ri; sqrt_rsqrt(r2, r, ri);
          double sn, cs; sincos_fast(k * r, sn, cs);
          const double gs = wei * (1.0 + sga) * aresub * jac * MA_INV4PI * ri;
          const double gre = cs * gs, gim = sn * gs;
          const double bre = -(gre * ri) - gim * k, bim = gre * k - gim * ri;
          const double a = (dx * nyx + dy * nyy + dz * nyz) * ri;
          hre += bre * a; him += bim * a;
        }
      }
      hre = wave_sum(hre); him = wave_sum(him);
      if (lane == 0) A[(size_t)i * g.n + jj] = dc_make(hre, him);
    }
  }
}

// rhs_i = - sum_s dG/dn(c_i - s) amp
__global__ __launch_bounds__(256) void room_incident_kernel(int n, const double* __restrict__ c, const double* __restrict__ nr, int nsrc, const double* __restrict__ sp,
                                                            const double* __restrict__ amp, int per_point, double k, dc* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double sr = 0.0, si = 0.0;
  for (int q = 0; q < nsrc; ++q) {
    const dc v = room_dgdn(c[3 * i] - sp[3 * q], c[3 * i + 1] - sp[3 * q + 1], c[3 * i + 2] - sp[3 * q + 2], nr[3 * i], nr[3 * i + 1], nr[3 * i + 2], k);
    const double a = per_point ? amp[(size_t)q * n + i] : amp[q];
    sr += v.re * a; si += v.im * a;
  }
  out[i] = dc_make(-sr, -si);
}

// one wavefront per field point: incident part (lane 0), then the surface sum with lanes over the elements
__global__ __launch_bounds__(256) void room_field_kernel(RoomGeom g, const dc* __restrict__ ps, int nsrc, const double* __restrict__ sp, const double* __restrict__ amp,
                                                         int per_point, int npts, const double* __restrict__ pts, double k, dc* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  if (m >= npts) return;
  const double x = pts[3 * m], y = pts[3 * m + 1], z = pts[3 * m + 2];
  double sr = 0.0, si = 0.0;
  for (int q = lane; q < nsrc; q += 64) {
    const double dx = x - sp[3 * q], dy = y - sp[3 * q + 1], dz = z - sp[3 * q + 2];
    const double r2 = dx * dx + dy * dy + dz * dz;
    if (r2 >= 1e-20) {                                                          // greens_function_3d, solver.rs:18-24
      double r, ri; sqrt_rsqrt(r2, r, ri);
      double sn, cs; sincos_fast(k * r, sn, cs);
      const double a = (per_point ? amp[(size_t)q * npts + m] : amp[q]) * MA_INV4PI * ri;
      sr += cs * a; si += sn * a;
    }
  }
  for (int j = lane; j < g.n; j += 64) {
    const dc v = room_dgdn(x - g.c[3 * j], y - g.c[3 * j + 1], z - g.c[3 * j + 2], g.nr[3 * j], g.nr[3 * j + 1], g.nr[3 * j + 2], k);
    const dc p = ps[j];
    const double a = g.ar[j];
    sr += (v.re * p.re - v.im * p.im) * a; si += (v.re * p.im + v.im * p.re) * a;
  }
  sr = wave_sum(sr); si = wave_sum(si);
  if (lane == 0) out[m] = dc_make(sr, si);
}

namespace {
struct DevBuf {
  std::vector<void*> ptrs;
  ~DevBuf() { for (void* p : ptrs) if (p) (void)hipFree(p); }
  template <class T> T* up(const T* h, size_t count, int* rc) {
    T* d = nullptr;
    if (*rc) return nullptr;
    if (hipMalloc(&d, sizeof(T) * (count ? count : 1)) != hipSuccess) { set_error("device allocation of %zu bytes failed", sizeof(T) * count); *rc = MA_ERR_NOMEM; return nullptr; }
    ptrs.push_back(d);
    if (h && count && hipMemcpy(d, h, sizeof(T) * count, hipMemcpyHostToDevice) != hipSuccess) { set_error("upload failed"); *rc = MA_ERR_HIP; }
    return d;
  }
};
int room_device() {
  int dev = 0; if (const char* s = getenv("MA_DEVICE")) dev = atoi(s);
  return use_device(dev);
}
int room_tables() {
  MA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_rgl_x), mat_gl_x, sizeof(double) * 94));
  MA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_rgl_w), mat_gl_w, sizeof(double) * 94));
  MA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_rgl_index), mat_gl_index, sizeof(int) * 42));
  return MA_OK;
}
}  // namespace

}  // namespace ma

using namespace ma;

extern "C" {

// element_center_and_normal (solver.rs:38-67), element_area (:70-122), element_characteristic_length (:600-611).
// conn: 4 ids per element, -1 in the fourth slot of a triangle. Host arithmetic (O(N)); needs no device.
int ma_room_element_data(int32_t n_elem, const double* nodes, const int32_t* conn, double* center, double* normal, double* area, double* charlen) {
  MA_REQUIRE(n_elem > 0 && nodes && conn && center && normal && area && charlen, MA_ERR_INVALID, "bad argument");
  for (int e = 0; e < n_elem; ++e) {
    const int32_t* cn = conn + 4 * e;
    const int nn = cn[3] < 0 ? 3 : 4;
    const double* p[4];
    for (int a = 0; a < nn; ++a) { MA_REQUIRE(cn[a] >= 0, MA_ERR_INVALID, "element %d has a negative node id", e); p[a] = nodes + 3 * cn[a]; }
    for (int d = 0; d < 3; ++d) { double s = 0.0; for (int a = 0; a < nn; ++a) s += p[a][d]; center[3 * e + d] = s / (double)nn; }
    double v1[3], v2[3];
    for (int d = 0; d < 3; ++d) { v1[d] = p[1][d] - p[0][d]; v2[d] = p[2][d] - p[0][d]; }
    const double nx = v1[1] * v2[2] - v1[2] * v2[1], ny = v1[2] * v2[0] - v1[0] * v2[2], nz = v1[0] * v2[1] - v1[1] * v2[0];
    const double nrm = std::sqrt(nx * nx + ny * ny + nz * nz);
    normal[3 * e] = nx / nrm; normal[3 * e + 1] = ny / nrm; normal[3 * e + 2] = nz / nrm;
    double a1 = 0.5 * nrm;
    if (nn == 4) {
      double v3[3];
      for (int d = 0; d < 3; ++d) v3[d] = p[3][d] - p[0][d];
      const double cx = v2[1] * v3[2] - v2[2] * v3[1], cy = v2[2] * v3[0] - v2[0] * v3[2], cz = v2[0] * v3[1] - v2[1] * v3[0];
      a1 += 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
    }
    area[e] = a1;
    double d01 = 0.0, d12 = 0.0, d20 = 0.0;
    for (int d = 0; d < 3; ++d) {
      d01 += (p[0][d] - p[1][d]) * (p[0][d] - p[1][d]); d12 += (p[1][d] - p[2][d]) * (p[1][d] - p[2][d]); d20 += (p[2][d] - p[0][d]) * (p[2][d] - p[0][d]);
    }
    charlen[e] = (std::sqrt(d01) + std::sqrt(d12) + std::sqrt(d20)) / 3.0;
  }
  return MA_OK;
}

// build_bem_matrix_adaptive(mesh, k, use_adaptive) (solver.rs:500-597): A is n_elem x n_elem row-major on the host
int ma_room_build_matrix_adaptive(int32_t n_nodes, const double* nodes, int32_t n_elem, const int32_t* conn, double k, int use_adaptive, ma_c64* A) {
  MA_REQUIRE(n_nodes > 0 && nodes && n_elem > 0 && conn && A, MA_ERR_INVALID, "bad argument");
  for (int e = 0; e < n_elem; ++e) for (int a = 0; a < 3; ++a) MA_REQUIRE(conn[4 * e + a] >= 0 && conn[4 * e + a] < n_nodes, MA_ERR_INVALID, "element %d references node %d", e, conn[4 * e + a]);
  const size_t n = (size_t)n_elem;
  std::vector<double> c(3 * n), nr(3 * n), ar(n), cl(n), tri(9 * n);
  int rc = ma_room_element_data(n_elem, nodes, conn, c.data(), nr.data(), ar.data(), cl.data());
  if (rc) return rc;
  for (size_t e = 0; e < n; ++e) for (int a = 0; a < 3; ++a) for (int d = 0; d < 3; ++d) tri[9 * e + 3 * a + d] = nodes[3 * conn[4 * e + a] + d];
  if ((rc = room_device())) return rc;
  if ((rc = room_tables())) return rc;
  DevBuf B;
  RoomGeom g{};
  g.n = n_elem;
  g.c = B.up(c.data(), 3 * n, &rc); g.nr = B.up(nr.data(), 3 * n, &rc); g.ar = B.up(ar.data(), n, &rc); g.cl = B.up(cl.data(), n, &rc); g.tri = B.up(tri.data(), 9 * n, &rc);
  c64* dA = B.up<c64>(nullptr, n * n, &rc);
  if (rc) return rc;
  hipLaunchKernelGGL(room_adaptive_kernel, dim3((n_elem + 3) / 4), dim3(256), 0, nullptr, g, k, use_adaptive ? 1 : 0, reinterpret_cast<dc*>(dA));
  MA_HIP(hipGetLastError());
  MA_HIP(hipMemcpy(A, dA, sizeof(c64) * n * n, hipMemcpyDeviceToHost));
  return MA_OK;
}

// calculate_incident_field_derivative_parallel (solver.rs:638-678). amp: [nsrc] (the same towards every element) or
// [nsrc][n] when per_point != 0 (Source::amplitude_towards evaluated by the caller, math-xem-common/src/source.rs:203-219)
int ma_room_incident_derivative(int32_t n, const double* center, const double* normal, int32_t nsrc, const double* src_pos, const double* amp, int per_point,
                                double k, ma_c64* out) {
  MA_REQUIRE(n > 0 && center && normal && nsrc >= 0 && (nsrc == 0 || (src_pos && amp)) && out, MA_ERR_INVALID, "bad argument");
  int rc = room_device(); if (rc) return rc;
  DevBuf B;
  const double* dc_ = B.up(center, 3 * (size_t)n, &rc); const double* dn = B.up(normal, 3 * (size_t)n, &rc);
  const double* ds = B.up(src_pos, 3 * (size_t)nsrc, &rc); const double* da = B.up(amp, (size_t)nsrc * (per_point ? (size_t)n : 1), &rc);
  c64* dout = B.up<c64>(nullptr, (size_t)n, &rc);
  if (rc) return rc;
  hipLaunchKernelGGL(room_incident_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, dc_, dn, nsrc, ds, da, per_point ? 1 : 0, k, reinterpret_cast<dc*>(dout));
  MA_HIP(hipGetLastError());
  MA_HIP(hipMemcpy(out, dout, sizeof(c64) * (size_t)n, hipMemcpyDeviceToHost));
  return MA_OK;
}

// calculate_field_pressure_bem_parallel (solver.rs:687-748): incident field of the sources + double-layer sum over the elements
int ma_room_field_pressure(int32_t n, const double* center, const double* normal, const double* area, const ma_c64* surface_pressure, int32_t nsrc,
                           const double* src_pos, const double* amp, int per_point, int32_t npts, const double* pts, double k, ma_c64* out) {
  MA_REQUIRE(n > 0 && center && normal && area && surface_pressure && npts > 0 && pts && out && nsrc >= 0 && (nsrc == 0 || (src_pos && amp)), MA_ERR_INVALID, "bad argument");
  int rc = room_device(); if (rc) return rc;
  DevBuf B;
  RoomGeom g{};
  g.n = n;
  g.c = B.up(center, 3 * (size_t)n, &rc); g.nr = B.up(normal, 3 * (size_t)n, &rc); g.ar = B.up(area, (size_t)n, &rc);
  const c64* dps = B.up(reinterpret_cast<const c64*>(surface_pressure), (size_t)n, &rc);
  const double* ds = B.up(src_pos, 3 * (size_t)nsrc, &rc); const double* da = B.up(amp, (size_t)nsrc * (per_point ? (size_t)npts : 1), &rc);
  const double* dp = B.up(pts, 3 * (size_t)npts, &rc);
  c64* dout = B.up<c64>(nullptr, (size_t)npts, &rc);
  if (rc) return rc;
  hipLaunchKernelGGL(room_field_kernel, dim3((npts + 3) / 4), dim3(256), 0, nullptr, g, reinterpret_cast<const dc*>(dps), nsrc, ds, da, per_point ? 1 : 0, npts, dp, k,
                     reinterpret_cast<dc*>(dout));
  MA_HIP(hipGetLastError());
  MA_HIP(hipMemcpy(out, dout, sizeof(c64) * (size_t)npts, hipMemcpyDeviceToHost));
  return MA_OK;
}

}  // extern "C"
