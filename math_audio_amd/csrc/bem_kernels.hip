// bem_kernels.hip — TBEM (Burton–Miller) dense assembly kernels for gfx950.
//
// What is computed follows the reference's build_tbem_system_with_beta
// (math-bem/src/core/assembly/tbem.rs:96-222) and the panel integrals it calls
// (integration/regular.rs:33-182, integration/singular.rs:123-394, :497-660). How it is
// computed is laid out for CDNA4:
//   K1 tbem_far_kernel   every (collocation i, field panel j) pair with the un-subdivided
//                        13-point rule; one lane per field panel (coalesced 16-B stores along
//                        a matrix row), the collocation point in SGPRs, a strip of rows per block.
//   K2 tbem_near_kernel  one wavefront per near pair (level-0 distance ratio < 3): the
//                        reference's level-by-level 4-way subdivision run lane-per-triangle
//                        (<= 60 per level fits one 64-wide wave), leaves collected in LDS, then
//                        all (leaf, point) tasks spread over the lanes and reduced with DPP
//                        butterflies. Overwrites the K1 value.
//   K3 tbem_self_kernel  one wavefront per panel for the singular self term (edge line
//                        integrals + collapsed-square sub-triangle rule) plus the free term.
// The near-pair list depends on geometry only and is built once per mesh (plan kernels below).
#include "bem_kernels.hpp"
#include "ma_device_math.hpp"

namespace ma {

// ------------------------------------------------------------------ constant tables (device)
// 13-point triangle rule, weights already scaled by 0.5 (reference gauss.rs:67-89, 386-400);
// uploaded by bem_upload_tables().
__constant__ double c_tri13[13][3];
__constant__ double c_gl_x[94];
__constant__ double c_gl_w[94];
__constant__ int c_gl_index[21][2];

int bem_upload_tables(const double tri13_scaled[13][3], const double* glx, const double* glw, const int glidx[21][2]) {
  MA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tri13), tri13_scaled, sizeof(double) * 39));
  MA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_gl_x), glx, sizeof(double) * 94));
  MA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_gl_w), glw, sizeof(double) * 94));
  MA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_gl_index), glidx, sizeof(int) * 42));
  return MA_OK;
}

#define MA_INV4PI 0.07957747154594767280

// ------------------------------------------------------------------ per-point Green's kernels
// One quadrature point of regular.rs:113-154 given d = y - x, the weight w (= rule weight x
// Jacobian / 4pi), the two normals and m = n_x . n_y. Accumulates G, dG/dn_y, dG/dn_x, d2G/dn_x dn_y.
struct Acc4 { dc g, h, ht, e; };

__device__ __forceinline__ void green_point(double dx, double dy, double dz, double w4pi, double k, double k2,
                                            double nyx, double nyy, double nyz, double nxx, double nxy, double nxz,
                                            double m, Acc4& acc) {
  double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
  if (!(r2 >= 1e-30)) return;                       // dis_fsp < 1e-15 -> continue (regular.rs:120)
  double r, ri;
  sqrt_rsqrt(r2, r, ri);
  double sn, cs;
  sincos_bounded(k * r, sn, cs);
  double gsc = w4pi * ri;
  double gre = cs * gsc, gim = sn * gsc;             // zg
  // zhh_base = zg * (-1/r + i k)
  double bre = -(gre * ri) - gim * k;
  double bim = gre * k - gim * ri;
  double a = (dx * nyx + dy * nyy + dz * nyz) * ri;  // (y-x).n_y / r
  double b = -((dx * nxx + dy * nxy + dz * nxz) * ri);
  double rq = a * b;
  double ri2 = ri * ri;
  double fr = (3.0 * ri2 - k2) * rq + m * ri2;
  double fi = -(k * ri) * (3.0 * rq + m);
  acc.g.re += gre; acc.g.im += gim;
  acc.h.re = __builtin_fma(bre, a, acc.h.re); acc.h.im = __builtin_fma(bim, a, acc.h.im);
  acc.ht.re = __builtin_fma(bre, b, acc.ht.re); acc.ht.im = __builtin_fma(bim, b, acc.ht.im);
  acc.e.re += gre * fr - gim * fi;
  acc.e.im += gre * fi + gim * fr;
}

// The same point for an UN-SUBDIVIDED flat triangle (the far kernel and the streamed operator): (y - x) . n_y is the same at
// every point of the panel (n_y is orthogonal to both edges) and (y - x) . n_x is affine in (xi, eta), so the two three-term
// dot products per point become one constant and two FMAs -- 4 of the ~125 vector instructions of a point.
__device__ __forceinline__ void green_point_flat(double dx, double dy, double dz, double w4pi, double k, double k2, double dny, double dnx, double m, Acc4& acc) {
  double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
  if (!(r2 >= 1e-30)) return;
  double r, ri;
  sqrt_rsqrt(r2, r, ri);
  double sn, cs;
  sincos_bounded(k * r, sn, cs);
  double gsc = w4pi * ri;
  double gre = cs * gsc, gim = sn * gsc;
  double bre = -(gre * ri) - gim * k;
  double bim = gre * k - gim * ri;
  double a = dny * ri;
  double b = -(dnx * ri);
  double rq = a * b;
  double ri2 = ri * ri;
  double fr = (3.0 * ri2 - k2) * rq + m * ri2;
  double fi = -(k * ri) * (3.0 * rq + m);
  acc.g.re += gre; acc.g.im += gim;
  acc.h.re = __builtin_fma(bre, a, acc.h.re); acc.h.im = __builtin_fma(bim, a, acc.h.im);
  acc.ht.re = __builtin_fma(bre, b, acc.ht.re); acc.ht.im = __builtin_fma(bim, b, acc.ht.im);
  acc.e.re += gre * fr - gim * fi;
  acc.e.im += gre * fi + gim * fr;
}

// Burton–Miller coefficient of one pair (assemble_tbem, tbem.rs:311-345; sign switch :203)
__device__ __forceinline__ dc bm_coeff(const Acc4& s, int field_bc, const BemPhys& ph) {
  double gt = ph.gamma * ph.tau;
  if (field_bc == 0) {
    dc hh = s.h * ph.sign;
    return dc_make(hh.re * gt + (s.e.re * ph.beta_re - s.e.im * ph.beta_im),
                   hh.im * gt + (s.e.re * ph.beta_im + s.e.im * ph.beta_re));
  } else if (field_bc == 1) {
    return dc_make(-(s.g.re * gt + (s.ht.re * ph.beta_re - s.ht.im * ph.beta_im)),
                   -(s.g.im * gt + (s.ht.re * ph.beta_im + s.ht.im * ph.beta_re)));
  }
  return dc_make(0.0, 0.0);
}

// ------------------------------------------------------------------ K1: far pairs
// grid.x: strips of 256 field panels, grid.y: strips of `rows_per_block` collocation rows.
// NF systems of one mesh at NF wavenumbers in one pass (the systems a sweep keeps in flight): of the ~120 vector instructions of a
// quadrature point, position, distance, reciprocal square root and the two normal projections (about 40) do not depend on the
// wavenumber and are computed once; sin / cos and the kernel values (about 80) per system. VEL: every panel carries a velocity-type
// condition -- assemble_tbem then reads only H and E (tbem.rs:311-330), so G and dG/dn_x are not accumulated (the per-lane
// condition type keeps the compiler from dropping them on its own).
struct FarMulti { BemPhys ph[3]; dc* A[3]; };
template <int NF, bool VEL>
__global__ __launch_bounds__(256) void tbem_far_kernel(BemGeom g, FarMulti fm, int rows_per_block, int row_blk0) {
  const int np = g.np;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int jj = j < np ? j : np - 1;
  const bool valid = j < np && !(g.nquad > 0 && g.ptype[jj] == 4);     // Quad4 columns belong to tbem_far_quad_kernel
  const double p0x = g.p0[0][jj], p0y = g.p0[1][jj], p0z = g.p0[2][jj];
  const double e1x = g.e1[0][jj], e1y = g.e1[1][jj], e1z = g.e1[2][jj];
  const double e2x = g.e2[0][jj], e2y = g.e2[1][jj], e2z = g.e2[2][jj];
  const double nyx = g.ny[0][jj], nyy = g.ny[1][jj], nyz = g.ny[2][jj];
  const double jw = g.jac[jj] * MA_INV4PI;
  const int fbc = VEL ? 0 : g.bc_type[jj];
  const long long col = g.dof[jj];
  double kk[NF], k2[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) { kk[f] = fm.ph[f].k * fm.ph[f].harmonic; k2[f] = fm.ph[f].k * fm.ph[f].k; }   // wavruim, k^2 (regular.rs:44-45)
  const int i0 = ((int)blockIdx.y + row_blk0) * rows_per_block;   // row_blk0: the first strip of a partial pass (ma_bem_plan_assemble_multi_part_dev)
  const int i1 = min(i0 + rows_per_block, np);
  for (int i = i0; i < i1; ++i) {
    // wave-uniform collocation data (scalar loads)
    const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];
    const double nxx = g.nx[0][i], nxy = g.nx[1][i], nxz = g.nx[2][i];
    const double d0x = p0x - cx, d0y = p0y - cy, d0z = p0z - cz;
    const double m = nxx * nyx + nxy * nyy + nxz * nyz;
    const double dny = d0x * nyx + d0y * nyy + d0z * nyz;             // (y - x) . n_y: constant over the flat panel
    const double dnx0 = d0x * nxx + d0y * nxy + d0z * nxz;            // (y - x) . n_x = dnx0 + xi e1.n_x + eta e2.n_x
    const double e1nx = e1x * nxx + e1y * nxy + e1z * nxz, e2nx = e2x * nxx + e2y * nxy + e2z * nxz;
    Acc4 s[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) s[f].g = s[f].h = s[f].ht = s[f].e = dc_make(0.0, 0.0);
#pragma unroll
    for (int q = 0; q < 13; ++q) {
      const double xi = c_tri13[q][0], eta = c_tri13[q][1], w = c_tri13[q][2];
      const double dx = __builtin_fma(eta, e2x, __builtin_fma(xi, e1x, d0x));
      const double dy = __builtin_fma(eta, e2y, __builtin_fma(xi, e1y, d0y));
      const double dz = __builtin_fma(eta, e2z, __builtin_fma(xi, e1z, d0z));
      const double dnx = __builtin_fma(eta, e2nx, __builtin_fma(xi, e1nx, dnx0));
      // green_point_flat, its wavenumber-free half once ...
      const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
      if (!(r2 >= 1e-30)) continue;
      double r, ri;
      sqrt_rsqrt(r2, r, ri);
      const double gsc = (w * jw) * ri;
      const double a = dny * ri;
      const double b = -(dnx * ri);
      const double rq = a * b;
      const double ri2 = ri * ri;
      const double mri2 = m * ri2, rq3m = 3.0 * rq + m, ri23 = 3.0 * ri2;
      // ... and the other half per system
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        double sn, cs;
        sincos_bounded(kk[f] * r, sn, cs);
        const double gre = cs * gsc, gim = sn * gsc;
        const double bre = -(gre * ri) - gim * kk[f];
        const double bim = gre * kk[f] - gim * ri;
        const double fr = (ri23 - k2[f]) * rq + mri2;
        const double fi = -(kk[f] * ri) * rq3m;
        if (!VEL) {
          s[f].g.re += gre; s[f].g.im += gim;
          s[f].ht.re = __builtin_fma(bre, b, s[f].ht.re); s[f].ht.im = __builtin_fma(bim, b, s[f].ht.im);
        }
        s[f].h.re = __builtin_fma(bre, a, s[f].h.re); s[f].h.im = __builtin_fma(bim, a, s[f].h.im);
        s[f].e.re += gre * fr - gim * fi;
        s[f].e.im += gre * fi + gim * fr;
      }
    }
    if (valid) {
      const long long at = (long long)g.dof[i] * g.nd + col;
#pragma unroll
      for (int f = 0; f < NF; ++f) fm.A[f][at] = bm_coeff(s[f], fbc, fm.ph[f]);
    }
  }
}

// ------------------------------------------------------------------ level-0 near test
// Bit-for-bit the reference's criterion (singular.rs:542-556 at the first level): centre of the
// un-split element in local coordinates, mapped with the shape functions, distance to the
// collocation point divided by sqrt(area). Contraction is off so that every product and sum is
// rounded exactly as the CPU restatement rounds it: the decision, not just the value, must agree.
#pragma clang fp contract(off)
__device__ __forceinline__ double tri_ratio(double s0, double t0, double s1, double t1, double s2, double t2,
                                            const double* v /* p0,p1,p2 = 9 */, double cx, double cy, double cz,
                                            double sq_arels) {
  double scent = (((0.0 + s0) + s1) + s2) / 3.0;
  double tcent = (((0.0 + t0) + t1) + t2) / 3.0;
  double n0 = 1.0 - scent - tcent;
  double px = ((0.0 + n0 * v[0]) + scent * v[3]) + tcent * v[6];
  double py = ((0.0 + n0 * v[1]) + scent * v[4]) + tcent * v[7];
  double pz = ((0.0 + n0 * v[2]) + scent * v[5]) + tcent * v[8];
  double dx = px - cx, dy = py - cy, dz = pz - cz;
  double dist = __builtin_sqrt(((0.0 + dx * dx) + dy * dy) + dz * dz);
  return dist / sq_arels;
}

// Quad4: bilinear shape functions on [-1,1]^2 (regular.rs:211-234) and the same level-0 / level-L criterion on the mean of
// the sub-element's four local vertices (singular.rs:542-556), rounded as the CPU restatement rounds it.
__device__ __forceinline__ double quad_ratio(const double* s, const double* t, const double* v /* 4 vertices x 3 */, double cx, double cy, double cz,
                                             double sq_arels) {
  double scent = ((((0.0 + s[0]) + s[1]) + s[2]) + s[3]) / 4.0;
  double tcent = ((((0.0 + t[0]) + t[1]) + t[2]) + t[3]) / 4.0;
  double s1 = 0.25 * (scent + 1.0), s2 = 0.25 * (scent - 1.0), t1 = tcent + 1.0, t2 = tcent - 1.0;
  double n0 = s1 * t1, n1 = -s2 * t1, n2 = s2 * t2, n3 = -s1 * t2;
  double px = (((0.0 + n0 * v[0]) + n1 * v[3]) + n2 * v[6]) + n3 * v[9];
  double py = (((0.0 + n0 * v[1]) + n1 * v[4]) + n2 * v[7]) + n3 * v[10];
  double pz = (((0.0 + n0 * v[2]) + n1 * v[5]) + n2 * v[8]) + n3 * v[11];
  double dx = px - cx, dy = py - cy, dz = pz - cz;
  double dist = __builtin_sqrt(((0.0 + dx * dx) + dy * dy) + dz * dz);
  return dist / sq_arels;
}
// compute_gauss_order (singular.rs:663-693): smallest order in [4, 7] whose three error estimates fall below 5e-4
__device__ __forceinline__ int quad_gauss_order(double ratdis) {
  const double disfac = 0.5 / ratdis;
  for (int order = 4; order <= 7; ++order) {
    const double base = disfac / (2.0 * (double)order + 1.0);
    double eg = 1.0;
    for (int e = 0; e < 2 * order + 1; ++e) eg *= base;
    const double eh = eg * base, ee = eh * base;
    if (eg < 0.0005 && eh < 0.0005 && ee < 0.0005) return order;
  }
  return 7;
}
__device__ __forceinline__ void quad_load(const BemGeom& g, int j, double* v) {
#pragma unroll
  for (int d = 0; d < 3; ++d) { v[d] = g.p0[d][j]; v[3 + d] = g.p1[d][j]; v[6 + d] = g.p2[d][j]; v[9 + d] = g.p3[d][j]; }
}

__device__ __forceinline__ bool pair_is_near(const BemGeom& g, int i, int j) {
  if (g.nquad > 0 && g.ptype[j] == 4) {
    double v4[12]; quad_load(g, j, v4);
    const double cs[4] = {1.0, -1.0, -1.0, 1.0}, ct[4] = {1.0, 1.0, -1.0, -1.0};
    double faclin = 2.0 * 0.5;
    double arels = g.area[j] * faclin * faclin;
    return quad_ratio(cs, ct, v4, g.c[0][i], g.c[1][i], g.c[2][i], __builtin_sqrt(arels)) < 3.0;
  }
  double v[9] = {g.p0[0][j], g.p0[1][j], g.p0[2][j], g.p1[0][j], g.p1[1][j], g.p1[2][j], g.p2[0][j], g.p2[1][j], g.p2[2][j]};
  double faclin = 2.0 * 0.5;
  double arels = g.area[j] * faclin * faclin;
  double r = tri_ratio(0.0, 0.0, 1.0, 0.0, 0.0, 1.0, v, g.c[0][i], g.c[1][i], g.c[2][i], __builtin_sqrt(arels));
  return r < 3.0;
}
#pragma clang fp contract(fast)

// One wavefront per collocation row: count / list the field panels j != i that need subdivision.
// pass 0: counts[i]; pass 1: pairs[offsets[i] + ...] in increasing j.
__global__ __launch_bounds__(256) void near_list_kernel(BemGeom g, int pass, int* __restrict__ counts,
                                                        const long long* __restrict__ offsets, int2* __restrict__ pairs) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= g.np) return;
  long long base = pass ? offsets[i] : 0;
  int cnt = 0;
  for (int j0 = 0; j0 < g.np; j0 += 64) {
    int j = j0 + lane;
    bool near = (j < g.np) && (j != i) && pair_is_near(g, i, j);
    unsigned long long m = __ballot(near);
    if (pass && near) {
      int pos = __popcll(m & lanemask_lt());
      pairs[base + cnt + pos] = make_int2(i, j);
    }
    cnt += __popcll(m);
  }
  if (!pass && lane == 0) counts[i] = cnt;
}

// ------------------------------------------------------------------ K2: near pairs
// LDS per wave: leaf list (110 x 6 doubles) and the next level's slots (60 x 6 doubles).
#define MA_MAX_LEAVES 110
#define MA_MAX_NSE 60

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// generate_subelements (singular.rs:497-660) for the Tri3 panel with vertices v seen from (cx, cy, cz): one lane per
// sub-triangle of the level; the leaves (local vertex coordinates) land in s_leaf[wave][0..nleaf). Returns nleaf.
__device__ __forceinline__ int near_build_leaves(const double* v, double area, double cx, double cy, double cz, int wave, int lane,
                                                 double (*s_leaf)[MA_MAX_LEAVES][6], double (*s_next)[MA_MAX_NSE][6]) {
  double s0 = 0.0, t0 = 0.0, s1 = 1.0, t1 = 0.0, s2 = 0.0, t2 = 1.0;
  int nsel = 1, nleaf = 0;
  double faclin = 2.0;
  for (;;) {
    faclin *= 0.5;
    double sq;
    {
#pragma clang fp contract(off)
      double arels = area * faclin * faclin;
      sq = __builtin_sqrt(arels);
    }
    const bool active = lane < nsel;
    double ratio = active ? tri_ratio(s0, t0, s1, t1, s2, t2, v, cx, cy, cz, sq) : 1e300;
    const bool split = active && (ratio < 3.0);
    const unsigned long long smask = __ballot(split);
    const int nsplit = __popcll(smask);
    const int rank = __popcll(smask & lanemask_lt());
    // `ndie > 15 => break` abandons the rest of the level from the 16th element that wants a split
    int cut = 64;
    if (nsplit > 15) {
      unsigned long long mm = smask;
      for (int q = 0; q < 15; ++q) mm &= mm - 1;     // clear the 15 lowest set bits
      cut = __builtin_ctzll(mm);
    }
    const bool isleaf = active && !split && lane < cut;
    const unsigned long long lmask = __ballot(isleaf);
    const int lrank = nleaf + __popcll(lmask & lanemask_lt());
    if (isleaf && lrank < MA_MAX_LEAVES) {
      double* L = s_leaf[wave][lrank];
      L[0] = s0; L[1] = t0; L[2] = s1; L[3] = t1; L[4] = s2; L[5] = t2;
    }
    nleaf += __popcll(lmask);
    if (nleaf >= MA_MAX_LEAVES) { nleaf = MA_MAX_LEAVES; break; }   // early return at 110 stored
    if (nsplit == 0) break;
    if (split && rank < 15) {
      // midpoints m0=(v0+v1)/2, m1=(v1+v2)/2, m2=(v2+v0)/2; children in slot order
      // [v0,m0,m2] [v1,m1,m0] [v2,m2,m1] [m0,m1,m2]   (singular.rs:567-611)
      double m0s = (s0 + s1) / 2.0, m0t = (t0 + t1) / 2.0;
      double m1s = (s1 + s2) / 2.0, m1t = (t1 + t2) / 2.0;
      double m2s = (s2 + s0) / 2.0, m2t = (t2 + t0) / 2.0;
      double* Cn = s_next[wave][rank * 4];
      Cn[0] = s0;  Cn[1] = t0;  Cn[2] = m0s; Cn[3] = m0t; Cn[4] = m2s; Cn[5] = m2t;
      Cn[6] = s1;  Cn[7] = t1;  Cn[8] = m1s; Cn[9] = m1t; Cn[10] = m0s; Cn[11] = m0t;
      Cn[12] = s2; Cn[13] = t2; Cn[14] = m2s; Cn[15] = m2t; Cn[16] = m1s; Cn[17] = m1t;
      Cn[18] = m0s; Cn[19] = m0t; Cn[20] = m1s; Cn[21] = m1t; Cn[22] = m2s; Cn[23] = m2t;
    }
    wave_lds_sync();
    nsel = (nsplit > 15 ? 15 : nsplit) * 4;
    if (lane < nsel) {
      const double* Cn = s_next[wave][lane];
      s0 = Cn[0]; t0 = Cn[1]; s1 = Cn[2]; t1 = Cn[3]; s2 = Cn[4]; t2 = Cn[5];
    }
    wave_lds_sync();
  }
  wave_lds_sync();

  return nleaf;
}

// PROBE = true: instead of the matrix entry, write the leaf count and the four raw integrals
// (G, H, H^T, E) of pair pid to out[5*pid..] — used by the parity tests on arbitrary (i != j) pairs.
template <int MODE>   // 0: write A[i][j]; 1: probe {leaves, G, H, H^T, E} to out[5 pid]; 2: coefficient to out[pid]
__global__ __launch_bounds__(256) void tbem_near_kernel(BemGeom g, BemPhys ph, const int2* __restrict__ pairs,
                                                        long long npairs, dc* __restrict__ A) {
  __shared__ double s_leaf[4][MA_MAX_LEAVES][6];
  __shared__ double s_next[4][MA_MAX_NSE][6];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long pid = (long long)blockIdx.x * 4 + wave;
  if (pid >= npairs) return;                         // whole wave leaves; no block barrier below
  const int2 pr = pairs[pid];
  const int i = pr.x, j = pr.y;
  if (g.nquad > 0 && g.ptype[j] == 4) return;        // Quad4 field panel: tbem_near_quad_kernel
  const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];
  const double nxx = g.nx[0][i], nxy = g.nx[1][i], nxz = g.nx[2][i];
  double v[9] = {g.p0[0][j], g.p0[1][j], g.p0[2][j], g.p1[0][j], g.p1[1][j], g.p1[2][j], g.p2[0][j], g.p2[1][j], g.p2[2][j]};
  const double area = g.area[j];

  const int nleaf = near_build_leaves(v, area, cx, cy, cz, wave, lane, s_leaf, s_next);

  // ---- integrate: tasks = (leaf, point); affine map of the 13-point rule into each leaf
  // (regular.rs:76-96), shape functions of the parent triangle (regular.rs:193-260).
  const double e1x = v[3] - v[0], e1y = v[4] - v[1], e1z = v[5] - v[2];
  const double e2x = v[6] - v[0], e2y = v[7] - v[1], e2z = v[8] - v[2];
  const double nyx = g.ny[0][j], nyy = g.ny[1][j], nyz = g.ny[2][j];
  const double jw = g.jac[j] * MA_INV4PI;
  const double d0x = v[0] - cx, d0y = v[1] - cy, d0z = v[2] - cz;
  const double m = nxx * nyx + nxy * nyy + nxz * nyz;
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k;
  Acc4 s;
  s.g = s.h = s.ht = s.e = dc_make(0.0, 0.0);
  const int ntask = nleaf * 13;
  for (int t = lane; t < ntask; t += 64) {
    const int lf = t / 13, q = t - lf * 13;
    const double* L = s_leaf[wave][lf];
    const double a0 = L[0], b0 = L[1], a1 = L[2], b1 = L[3], a2 = L[4], b2 = L[5];
    const double xi = c_tri13[q][0], eta = c_tri13[q][1], w = c_tri13[q][2];
    const double l0 = 1.0 - xi - eta;
    const double xio = a0 * l0 + a1 * xi + a2 * eta;
    const double eto = b0 * l0 + b1 * xi + b2 * eta;
    const double det = __builtin_fabs((a1 - a0) * (b2 - b0) - (a2 - a0) * (b1 - b0));
    double dx = __builtin_fma(eto, e2x, __builtin_fma(xio, e1x, d0x));
    double dy = __builtin_fma(eto, e2y, __builtin_fma(xio, e1y, d0y));
    double dz = __builtin_fma(eto, e2z, __builtin_fma(xio, e1z, d0z));
    green_point(dx, dy, dz, w * det * jw, k, k2, nyx, nyy, nyz, nxx, nxy, nxz, m, s);
  }
  s.g.re = wave_sum(s.g.re); s.g.im = wave_sum(s.g.im);
  s.h.re = wave_sum(s.h.re); s.h.im = wave_sum(s.h.im);
  s.ht.re = wave_sum(s.ht.re); s.ht.im = wave_sum(s.ht.im);
  s.e.re = wave_sum(s.e.re); s.e.im = wave_sum(s.e.im);
  if (lane == 0) {
    if (MODE == 1) {
      dc* o = A + 5 * pid;
      o[0] = dc_make((double)nleaf, 0.0); o[1] = s.g; o[2] = s.h; o[3] = s.ht; o[4] = s.e;
    } else if (MODE == 2) {
      A[pid] = bm_coeff(s, g.bc_type[j], ph);
    } else {
      A[(long long)g.dof[i] * g.nd + g.dof[j]] = bm_coeff(s, g.bc_type[j], ph);
    }
  }
}

// The near pairs of NF systems of one mesh in one pass (round 4): the sub-triangle leaves of a pair (generate_subelements,
// singular.rs:497-660) depend on the geometry alone, and so do position, distance, reciprocal square root and the two normal
// projections of every quadrature point; sin / cos and the kernel values are per system. Per system the operations and their order
// are those of tbem_near_kernel<0> (green_point); the compiler contracts them into FMAs on its own terms in either kernel, so the entries
// agree to rounding (1e-13 of the row scale, test_multi_frequency_assembly_equals_the_single_one), not bit for bit.
template <int NF>
__global__ __launch_bounds__(256) void tbem_near_multi_kernel(BemGeom g, FarMulti fm, const int2* __restrict__ pairs, long long npairs) {
  __shared__ double s_leaf[4][MA_MAX_LEAVES][6];
  __shared__ double s_next[4][MA_MAX_NSE][6];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long pid = (long long)blockIdx.x * 4 + wave;
  if (pid >= npairs) return;                         // whole wave leaves; no block barrier below
  const int2 pr = pairs[pid];
  const int i = pr.x, j = pr.y;
  if (g.nquad > 0 && g.ptype[j] == 4) return;        // Quad4 field panel: tbem_near_quad_kernel, per system
  const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];
  const double nxx = g.nx[0][i], nxy = g.nx[1][i], nxz = g.nx[2][i];
  double v[9] = {g.p0[0][j], g.p0[1][j], g.p0[2][j], g.p1[0][j], g.p1[1][j], g.p1[2][j], g.p2[0][j], g.p2[1][j], g.p2[2][j]};
  const double area = g.area[j];
  const int nleaf = near_build_leaves(v, area, cx, cy, cz, wave, lane, s_leaf, s_next);
  const double e1x = v[3] - v[0], e1y = v[4] - v[1], e1z = v[5] - v[2];
  const double e2x = v[6] - v[0], e2y = v[7] - v[1], e2z = v[8] - v[2];
  const double nyx = g.ny[0][j], nyy = g.ny[1][j], nyz = g.ny[2][j];
  const double jw = g.jac[j] * MA_INV4PI;
  const double d0x = v[0] - cx, d0y = v[1] - cy, d0z = v[2] - cz;
  const double m = nxx * nyx + nxy * nyy + nxz * nyz;
  double kk[NF], kk2[NF];
  Acc4 s[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) { kk[f] = fm.ph[f].k * fm.ph[f].harmonic; kk2[f] = fm.ph[f].k * fm.ph[f].k; s[f].g = s[f].h = s[f].ht = s[f].e = dc_make(0.0, 0.0); }
  const int ntask = nleaf * 13;
  for (int t = lane; t < ntask; t += 64) {
    const int lf = t / 13, q = t - lf * 13;
    const double* L = s_leaf[wave][lf];
    const double a0 = L[0], b0 = L[1], a1 = L[2], b1 = L[3], a2 = L[4], b2 = L[5];
    const double xi = c_tri13[q][0], eta = c_tri13[q][1], w = c_tri13[q][2];
    const double l0 = 1.0 - xi - eta;
    const double xio = a0 * l0 + a1 * xi + a2 * eta;
    const double eto = b0 * l0 + b1 * xi + b2 * eta;
    const double det = __builtin_fabs((a1 - a0) * (b2 - b0) - (a2 - a0) * (b1 - b0));
    const double dx = __builtin_fma(eto, e2x, __builtin_fma(xio, e1x, d0x));
    const double dy = __builtin_fma(eto, e2y, __builtin_fma(xio, e1y, d0y));
    const double dz = __builtin_fma(eto, e2z, __builtin_fma(xio, e1z, d0z));
    // green_point, its wavenumber-free half once ...
    const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
    if (!(r2 >= 1e-30)) continue;
    double r, ri;
    sqrt_rsqrt(r2, r, ri);
    const double gsc = (w * det * jw) * ri;
    const double a = (dx * nyx + dy * nyy + dz * nyz) * ri;
    const double b = -((dx * nxx + dy * nxy + dz * nxz) * ri);
    const double rq = a * b, ri2 = ri * ri;
    // ... and the rest per system
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const double k = kk[f];
      double sn, cs;
      sincos_bounded(k * r, sn, cs);
      const double gre = cs * gsc, gim = sn * gsc;
      const double bre = -(gre * ri) - gim * k;
      const double bim = gre * k - gim * ri;
      const double fr = (3.0 * ri2 - kk2[f]) * rq + m * ri2;
      const double fi = -(k * ri) * (3.0 * rq + m);
      Acc4& acc = s[f];
      acc.g.re += gre; acc.g.im += gim;
      acc.h.re = __builtin_fma(bre, a, acc.h.re); acc.h.im = __builtin_fma(bim, a, acc.h.im);
      acc.ht.re = __builtin_fma(bre, b, acc.ht.re); acc.ht.im = __builtin_fma(bim, b, acc.ht.im);
      acc.e.re += gre * fr - gim * fi;
      acc.e.im += gre * fi + gim * fr;
    }
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    Acc4& a = s[f];
    a.g.re = wave_sum(a.g.re); a.g.im = wave_sum(a.g.im);
    a.h.re = wave_sum(a.h.re); a.h.im = wave_sum(a.h.im);
    a.ht.re = wave_sum(a.ht.re); a.ht.im = wave_sum(a.ht.im);
    a.e.re = wave_sum(a.e.re); a.e.im = wave_sum(a.e.im);
    if (lane == 0) fm.A[f][(long long)g.dof[i] * g.nd + g.dof[j]] = bm_coeff(a, g.bc_type[j], fm.ph[f]);
  }
}

// ------------------------------------------------------------------ K3: self terms
// singular_integration_with_params (singular.rs:154-394) for Tri3, QuadratureParams::for_ka (:48-82)
// from ka = k x mean edge length (:730-745). One wavefront per panel; the flattened point list
// (3 edges x [edge-line points | sub-triangle tensor points]) is strided over the lanes.
__device__ __constant__ double c_csi6[6] = {0.0, 1.0, 0.0, 0.5, 0.5, 0.0};
__device__ __constant__ double c_eta6[6] = {0.0, 0.0, 1.0, 0.0, 0.5, 0.5};

template <int MODE>   // 0: write A[i][i]; 1: probe to out[5 e]; 2: diagonal entry (with free term) to out[e]
__global__ __launch_bounds__(256) void tbem_self_kernel(BemGeom g, BemPhys ph, dc* __restrict__ A) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + wave;
  if (e >= g.np) return;
  if (g.nquad > 0 && g.ptype[e] == 4) return;        // Quad4 panel: tbem_self_quad_kernel
  const double cx = g.c[0][e], cy = g.c[1][e], cz = g.c[2][e];
  const double nxx = g.nx[0][e], nxy = g.nx[1][e], nxz = g.nx[2][e];
  const double P[3][3] = {{g.p0[0][e], g.p0[1][e], g.p0[2][e]}, {g.p1[0][e], g.p1[1][e], g.p1[2][e]}, {g.p2[0][e], g.p2[1][e], g.p2[2][e]}};
  const double nyx = g.ny[0][e], nyy = g.ny[1][e], nyz = g.ny[2][e];
  const double jac = g.jac[e];
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k;
  // mean edge length -> quadrature tier
  double el = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int b = (a + 1) % 3;
    double ddx = P[b][0] - P[a][0], ddy = P[b][1] - P[a][1], ddz = P[b][2] - P[a][2];
    el += __builtin_sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
  }
  const double ka = ph.k * (el / 3.0);
  int ngpo1, ngausin, nsec1, nsec2;
  if (ka < 0.3)      { ngpo1 = 3; ngausin = 4; nsec1 = 4;  nsec2 = 2; }
  else if (ka < 1.0) { ngpo1 = 4; ngausin = 5; nsec1 = 6;  nsec2 = 2; }
  else if (ka < 2.0) { ngpo1 = 5; ngausin = 6; nsec1 = 8;  nsec2 = 3; }
  else               { ngpo1 = 6; ngausin = 7; nsec1 = 10; nsec2 = 4; }
  const int eo = c_gl_index[ngpo1][0], so = c_gl_index[ngausin][0];
  const int ne = c_gl_index[ngpo1][1], ns = c_gl_index[ngausin][1];
  const int n_edge_pts = nsec1 * ne;
  const int per_edge = n_edge_pts + nsec2 * ns * ns;
  const int ntask = 3 * per_edge;
  const double m = nxx * nyx + nxy * nyy + nxz * nyz;

  dc sg = dc_make(0, 0), sh = dc_make(0, 0), sht = dc_make(0, 0), se = dc_make(0, 0);
  for (int t = lane; t < ntask; t += 64) {
    const int ieg = t / per_edge;
    const int u = t - ieg * per_edge;
    const int ig1 = (ieg + 1) % 3, ig2 = ieg + 3;
    if (u < n_edge_pts) {
      // ---- hypersingular part as an edge line integral (singular.rs:182-254)
      const int isec = u / ne, ig = u - isec * ne;
      double dpx = P[ig1][0] - P[ieg][0], dpy = P[ig1][1] - P[ieg][1], dpz = P[ig1][2] - P[ieg][2];
      double len = __builtin_sqrt(dpx * dpx + dpy * dpy + dpz * dpz);
      double ox = dpx / len, oy = dpy / len, oz = dpz / len;
      double lens = len / (2.0 * (double)nsec1);
      double delsec = 2.0 / (double)nsec1;
      double secmid = -1.0 - delsec / 2.0;
      for (int q = 0; q <= isec; ++q) secmid += delsec;
      double sga = secmid + c_gl_x[eo + ig] / (double)nsec1;
      double wga = c_gl_w[eo + ig] * lens;
      double f = (sga + 1.0) / 2.0;
      double dx = (P[ieg][0] + dpx * f) - cx, dy = (P[ieg][1] + dpy * f) - cy, dz = (P[ieg][2] + dpz * f) - cz;
      double r2 = dx * dx + dy * dy + dz * dz;
      if (r2 >= 1e-30) {
        double r, ri; sqrt_rsqrt(r2, r, ri);
        double sn, cs; sincos_bounded(k * r, sn, cs);
        double gs = MA_INV4PI * ri;
        double gre = cs * gs, gim = sn * gs;
        double fre = -(gre * ri) - gim * k, fim = gre * k - gim * ri;   // zg * (-1/r + ik)
        double ux = dx * ri, uy = dy * ri, uz = dz * ri;
        // ((grad G) x edge_dir) . n_x  = zg_factor * ((u x o) . n_x)
        double wx = uy * oz - uz * oy, wy = uz * ox - ux * oz, wz = ux * oy - uy * ox;
        double sc = (wx * nxx + wy * nxy + wz * nxz) * wga;
        se.re += fre * sc; se.im += fim * sc;
      }
    } else {
      // ---- G, H, H^T on the collapsed-square sub-triangles (singular.rs:257-357)
      const int v2 = u - n_edge_pts;
      const int isec = v2 / (ns * ns);
      const int ij = v2 - isec * ns * ns;
      const int ii = ij / ns, jj = ij - ii * ns;
      const double aresub = 1.0 / 24.0 / (double)nsec2;
      const double ss0 = 1.0 / 3.0, ts0 = 1.0 / 3.0;
      double ss1, ss2, ts1, ts2;
      if (isec == 0) { ss1 = c_csi6[ieg]; ss2 = c_csi6[ig2]; ts1 = c_eta6[ieg]; ts2 = c_eta6[ig2]; }
      else           { ss1 = c_csi6[ig2]; ss2 = c_csi6[ig1]; ts1 = c_eta6[ig2]; ts2 = c_eta6[ig1]; }
      const double sga = c_gl_x[so + ii], tga = c_gl_x[so + jj];
      const double wei = c_gl_w[so + ii] * c_gl_w[so + jj];
      const double sgg = 0.5 * (1.0 - sga) * ss0 + 0.25 * (1.0 + sga) * ((1.0 - tga) * ss1 + (1.0 + tga) * ss2);
      const double tgg = 0.5 * (1.0 - sga) * ts0 + 0.25 * (1.0 + sga) * ((1.0 - tga) * ts1 + (1.0 + tga) * ts2);
      const double n0 = 1.0 - sgg - tgg;
      double dx = (n0 * P[0][0] + sgg * P[1][0] + tgg * P[2][0]) - cx;
      double dy = (n0 * P[0][1] + sgg * P[1][1] + tgg * P[2][1]) - cy;
      double dz = (n0 * P[0][2] + sgg * P[1][2] + tgg * P[2][2]) - cz;
      const double wga = wei * (1.0 + sga) * aresub * jac;
      double r2 = dx * dx + dy * dy + dz * dz;
      if (r2 >= 1e-30) {
        double r, ri; sqrt_rsqrt(r2, r, ri);
        double sn, cs; sincos_bounded(k * r, sn, cs);
        double gs = wga * MA_INV4PI * ri;
        double gre = cs * gs, gim = sn * gs;
        double bre = -(gre * ri) - gim * k, bim = gre * k - gim * ri;
        double a = (dx * nyx + dy * nyy + dz * nyz) * ri;
        double b = -((dx * nxx + dy * nxy + dz * nxz) * ri);
        sg.re += gre; sg.im += gim;
        sh.re += bre * a; sh.im += bim * a;
        sht.re += bre * b; sht.im += bim * b;
        se.re += gre * k2 * m; se.im += gim * k2 * m;       // E += zg k^2 (n_x . n_y)  (singular.rs:357)
      }
    }
  }
  Acc4 s;
  s.g = dc_make(wave_sum(sg.re), wave_sum(sg.im));
  s.h = dc_make(wave_sum(sh.re), wave_sum(sh.im));
  s.ht = dc_make(wave_sum(sht.re), wave_sum(sht.im));
  s.e = dc_make(wave_sum(se.re), wave_sum(se.im));
  if (MODE == 1) {
    if (lane == 0) {
      dc* o = A + 5 * (long long)e;
      o[0] = dc_make((double)ntask, 0.0); o[1] = s.g; o[2] = s.h; o[3] = s.ht; o[4] = s.e;
    }
    return;
  }
  if (lane == 0) {
    const int bc = g.bc_type[e];
    dc coeff = bm_coeff(s, bc, ph);
    // free term (add_free_terms, tbem.rs:273-304): velocity -gamma/2, pressure -beta tau/2
    dc fr = dc_make(0.0, 0.0);
    if (bc == 0) fr = dc_make(-(ph.gamma * 0.5), 0.0);
    else if (bc == 1) fr = dc_make(-(ph.beta_re * ph.tau * 0.5), -(ph.beta_im * ph.tau * 0.5));
    const long long d = g.dof[e];
    if (MODE == 2) A[e] = dc_make(fr.re + coeff.re, fr.im + coeff.im);
    else A[d * g.nd + d] = dc_make(fr.re + coeff.re, fr.im + coeff.im);
  }
}

// ------------------------------------------------------------------ Quad4 panels
// The same three regimes for bilinear quadrilaterals (ElementType::Quad4): the position, the unit normal and the
// Jacobian vary over the panel (compute_parameters, regular.rs:211-260), the rule of an (un)subdivided piece is the
// n x n Gauss-Legendre tensor rule with n in [4, 7] from the distance criterion (gauss.rs:94-105, singular.rs:663-693),
// pieces are axis-aligned squares (centre, half-width) in the reference square (regular.rs:96-103).
__device__ __constant__ double c_csi8[8] = {1.0, -1.0, -1.0, 1.0, 0.0, -1.0, 0.0, 1.0};
__device__ __constant__ double c_eta8[8] = {1.0, 1.0, -1.0, -1.0, 1.0, 0.0, -1.0, 0.0};

struct QuadPoint { double dx, dy, dz, nyx, nyy, nyz, jac; };
// point (s, t) of the panel with vertices v, relative to the collocation point c
__device__ __forceinline__ QuadPoint quad_point(const double* v, double s, double t, double cx, double cy, double cz) {
  const double s1 = 0.25 * (s + 1.0), s2 = 0.25 * (s - 1.0), t1 = t + 1.0, t2 = t - 1.0;
  const double n0 = s1 * t1, n1 = -s2 * t1, n2 = s2 * t2, n3 = -s1 * t2;
  const double a0 = 0.25 * (t + 1.0), a1 = -0.25 * (t + 1.0), a2 = 0.25 * (t - 1.0), a3 = -0.25 * (t - 1.0);     // dN/ds
  const double b0 = 0.25 * (s + 1.0), b1 = 0.25 * (1.0 - s), b2 = 0.25 * (s - 1.0), b3 = -0.25 * (s + 1.0);      // dN/dt
  double p[3], ds[3], dt[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    p[d] = n0 * v[d] + n1 * v[3 + d] + n2 * v[6 + d] + n3 * v[9 + d];
    ds[d] = a0 * v[d] + a1 * v[3 + d] + a2 * v[6 + d] + a3 * v[9 + d];
    dt[d] = b0 * v[d] + b1 * v[3 + d] + b2 * v[6 + d] + b3 * v[9 + d];
  }
  const double nx = ds[1] * dt[2] - ds[2] * dt[1], ny = ds[2] * dt[0] - ds[0] * dt[2], nz = ds[0] * dt[1] - ds[1] * dt[0];
  const double jac = __builtin_sqrt(nx * nx + ny * ny + nz * nz);
  const double ij = jac > 1e-15 ? 1.0 / jac : 0.0;
  QuadPoint q;
  q.dx = p[0] - cx; q.dy = p[1] - cy; q.dz = p[2] - cz;
  q.nyx = nx * ij; q.nyy = ny * ij; q.nyz = nz * ij; q.jac = jac;
  return q;
}
__device__ __forceinline__ void quad_green(const double* v, double s, double t, double w, double cx, double cy, double cz,
                                           double nxx, double nxy, double nxz, double k, double k2, Acc4& acc) {
  const QuadPoint q = quad_point(v, s, t, cx, cy, cz);
  const double m = nxx * q.nyx + nxy * q.nyy + nxz * q.nyz;
  green_point(q.dx, q.dy, q.dz, w * q.jac * MA_INV4PI, k, k2, q.nyx, q.nyy, q.nyz, nxx, nxy, nxz, m, acc);
}

// un-subdivided coefficient of the pair (row i, quad panel with vertices v): the rule of the order the distance asks for
__device__ __forceinline__ dc quad_far_coeff(const BemGeom& g, const double* v, double sq, int fbc, int i, const BemPhys& ph, double k, double k2) {
  const double cs[4] = {1.0, -1.0, -1.0, 1.0}, ct[4] = {1.0, 1.0, -1.0, -1.0};
  const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];
  const double nxx = g.nx[0][i], nxy = g.nx[1][i], nxz = g.nx[2][i];
  const int order = quad_gauss_order(quad_ratio(cs, ct, v, cx, cy, cz, sq));
  const int off = c_gl_index[order][0], n = c_gl_index[order][1];
  Acc4 s;
  s.g = s.h = s.ht = s.e = dc_make(0.0, 0.0);
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b)
      quad_green(v, c_gl_x[off + a], c_gl_x[off + b], c_gl_w[off + a] * c_gl_w[off + b], cx, cy, cz, nxx, nxy, nxz, k, k2, s);
  return bm_coeff(s, fbc, ph);
}
__device__ __forceinline__ double quad_sq_area(const BemGeom& g, int j) {
#pragma clang fp contract(off)
  double arels = g.area[j] * 1.0 * 1.0;
  return __builtin_sqrt(arels);
}

// K1q: every pair whose field panel is a quad, un-subdivided rule of the order the distance asks for.
// grid.x: strips of 256 quad panels (lane = panel), grid.y: strips of collocation rows.
__global__ __launch_bounds__(256) void tbem_far_quad_kernel(BemGeom g, BemPhys ph, dc* __restrict__ A, int rows_per_block, int row_blk0) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  const bool valid = q < g.nquad;
  const int j = g.quad_ids[valid ? q : 0];
  double v[12]; quad_load(g, j, v);
  const int fbc = g.bc_type[j];
  const long long col = g.dof[j];
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k;
  const double sq = quad_sq_area(g, j);
  const int i0 = ((int)blockIdx.y + row_blk0) * rows_per_block, i1 = min(i0 + rows_per_block, g.np);
  for (int i = i0; i < i1; ++i) {
    const dc a = quad_far_coeff(g, v, sq, fbc, i, ph, k, k2);
    if (valid) A[(long long)g.dof[i] * g.nd + col] = a;
  }
}

// ---- the Quad4 columns of the matrix-free operator (op_kernels.hip streams the Tri3 columns): the loop nest of
// tbem_far_quad_kernel, accumulating a_ij x_j instead of storing a_ij. Lane = quad panel; the 64 lanes' products of a row are
// summed with DPP, a wavefront collects 64 consecutive rows in its lanes, the four wavefronts are added through LDS.
// grid.x: strips of 256 quad panels, grid.y: tiles of rows_per_block rows (a multiple of 64) of [row0, row1);
// partial[strip][row - row0] (the caller passes the first quad strip's address).
__global__ __launch_bounds__(256) void tbem_matvec_quad_kernel(BemGeom g, BemPhys ph, int row0, int row1, int rows_per_block, const dc* __restrict__ x,
                                                               dc* __restrict__ partial) {
  __shared__ double s_re[4][64], s_im[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * 256 + threadIdx.x;
  const bool valid = q < g.nquad;
  const int j = g.quad_ids[valid ? q : 0];
  double v[12]; quad_load(g, j, v);
  const int fbc = g.bc_type[j];
  const dc xj = valid ? x[g.dof[j]] : dc_make(0.0, 0.0);
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k;
  const double sq = quad_sq_area(g, j);
  const int nr = row1 - row0;
  const int t0 = row0 + blockIdx.y * rows_per_block, t1 = min(t0 + rows_per_block, row1);
  for (int base = t0; base < t1; base += 64) {
    double acc_re = 0.0, acc_im = 0.0;
    const int iend = min(base + 64, t1);
    for (int i = base; i < iend; ++i) {
      const dc a = quad_far_coeff(g, v, sq, fbc, i, ph, k, k2);
      const double pr = wave_sum_lane63(a.re * xj.re - a.im * xj.im), pi = wave_sum_lane63(a.re * xj.im + a.im * xj.re);
      const int tl = i - base;
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
// transposed: partial[chunk][j] = sum over the chunk's rows of a_ij x[dof_i] for the quad panels j (overwrites what the Tri3
// kernel left in those slots; same chunking)
__global__ __launch_bounds__(256) void tbem_matvec_quad_t_kernel(BemGeom g, BemPhys ph, int row0, int row1, int chunk_rows, const dc* __restrict__ x,
                                                                 dc* __restrict__ partial) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  const bool valid = q < g.nquad;
  const int j = g.quad_ids[valid ? q : 0];
  double v[12]; quad_load(g, j, v);
  const int fbc = g.bc_type[j];
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k;
  const double sq = quad_sq_area(g, j);
  const int i0 = row0 + blockIdx.y * chunk_rows, i1 = min(i0 + chunk_rows, row1);
  double yr = 0.0, yi = 0.0;
  for (int i = i0; i < i1; ++i) {
    const dc xi = x[g.dof[i]];
    const dc a = quad_far_coeff(g, v, sq, fbc, i, ph, k, k2);
    yr += a.re * xi.re - a.im * xi.im; yi += a.re * xi.im + a.im * xi.re;
  }
  if (valid) partial[(size_t)blockIdx.y * g.np + j] = dc_make(yr, yi);
}
// the streamed coefficient of listed pairs whose field panel is a quad (to form corrections A_true - A_streamed)
__global__ __launch_bounds__(256) void tbem_pairs_quad_far_kernel(BemGeom g, BemPhys ph, const int2* __restrict__ pairs, long long npairs, dc* __restrict__ out) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= npairs) return;
  const int i = pairs[q].x, j = pairs[q].y;
  if (g.ptype[j] != 4) return;
  double v[12]; quad_load(g, j, v);
  out[q] = quad_far_coeff(g, v, quad_sq_area(g, j), g.bc_type[j], i, ph, ph.k * ph.harmonic, ph.k * ph.k);
}

// generate_subelements for nv = 4: leaves {xi centre, eta centre, half-width, Gauss order} in s_leaf[wave][0..nleaf)
__device__ __forceinline__ int quad_build_leaves(const double* v, double area, double cx, double cy, double cz, int wave, int lane,
                                                 double (*s_leaf)[MA_MAX_LEAVES][4], double (*s_next)[MA_MAX_NSE][8]) {
  double sv[4] = {1.0, -1.0, -1.0, 1.0}, tv[4] = {1.0, 1.0, -1.0, -1.0};
  int nsel = 1, nleaf = 0;
  double faclin = 2.0;
  for (;;) {
    faclin *= 0.5;
    double sq;
    {
#pragma clang fp contract(off)
      double arels = area * faclin * faclin;
      sq = __builtin_sqrt(arels);
    }
    const bool active = lane < nsel;
    const double ratio = active ? quad_ratio(sv, tv, v, cx, cy, cz, sq) : 1e300;
    const bool split = active && (ratio < 3.0);
    const unsigned long long smask = __ballot(split);
    const int nsplit = __popcll(smask);
    const int rank = __popcll(smask & lanemask_lt());
    int cut = 64;
    if (nsplit > 15) {
      unsigned long long mm = smask;
      for (int q = 0; q < 15; ++q) mm &= mm - 1;
      cut = __builtin_ctzll(mm);
    }
    const bool isleaf = active && !split && lane < cut;
    const unsigned long long lmask = __ballot(isleaf);
    const int lrank = nleaf + __popcll(lmask & lanemask_lt());
    if (isleaf && lrank < MA_MAX_LEAVES) {
      double* L = s_leaf[wave][lrank];
      double xc, ec;
      {
#pragma clang fp contract(off)
        xc = (((sv[0] + sv[1]) + sv[2]) + sv[3]) / 4.0;
        ec = (((tv[0] + tv[1]) + tv[2]) + tv[3]) / 4.0;
      }
      L[0] = xc; L[1] = ec; L[2] = faclin; L[3] = (double)quad_gauss_order(ratio);
    }
    nleaf += __popcll(lmask);
    if (nleaf >= MA_MAX_LEAVES) { nleaf = MA_MAX_LEAVES; break; }
    if (nsplit == 0) break;
    if (split && rank < 15) {
      double scent, tcent, ms[4], mt[4];
      {
#pragma clang fp contract(off)
        scent = ((((0.0 + sv[0]) + sv[1]) + sv[2]) + sv[3]) / 4.0;
        tcent = ((((0.0 + tv[0]) + tv[1]) + tv[2]) + tv[3]) / 4.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) { ms[a] = (sv[a] + sv[(a + 1) & 3]) / 2.0; mt[a] = (tv[a] + tv[(a + 1) & 3]) / 2.0; }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {                      // child a: v_a, mid(a, a+1), centre, mid(a-1, a)
        double* Cn = s_next[wave][rank * 4 + a];
        const int am = (a + 3) & 3;
        Cn[0] = sv[a]; Cn[1] = ms[a]; Cn[2] = scent; Cn[3] = ms[am];
        Cn[4] = tv[a]; Cn[5] = mt[a]; Cn[6] = tcent; Cn[7] = mt[am];
      }
    }
    wave_lds_sync();
    nsel = (nsplit > 15 ? 15 : nsplit) * 4;
    if (lane < nsel) {
      const double* Cn = s_next[wave][lane];
#pragma unroll
      for (int a = 0; a < 4; ++a) { sv[a] = Cn[a]; tv[a] = Cn[4 + a]; }
    }
    wave_lds_sync();
  }
  wave_lds_sync();

  return nleaf;
}

// K2q: near pairs with a quad field panel. generate_subelements for nv = 4 (singular.rs:497-660): a split yields the
// children [v_j, mid(v_j, v_j+1), centre, mid(v_j-1, v_j)], j = 0..3, each with its own vertex order (which decides the
// order of ITS children, hence which pieces the 15-splits-per-level limit drops); leaves keep centre, half-width and order.
template <int MODE>   // 0: write A[i][j]; 1: probe {leaves, G, H, H^T, E}; 2: coefficient to out[pid]
__global__ __launch_bounds__(256) void tbem_near_quad_kernel(BemGeom g, BemPhys ph, const int2* __restrict__ pairs, long long npairs, dc* __restrict__ A) {
  __shared__ double s_leaf[4][MA_MAX_LEAVES][4];
  __shared__ double s_next[4][MA_MAX_NSE][8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long pid = (long long)blockIdx.x * 4 + wave;
  if (pid >= npairs) return;
  const int2 pr = pairs[pid];
  const int i = pr.x, j = pr.y;
  if (g.ptype[j] != 4) return;
  const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];
  const double nxx = g.nx[0][i], nxy = g.nx[1][i], nxz = g.nx[2][i];
  double v[12]; quad_load(g, j, v);
  const double area = g.area[j];
  const int nleaf = quad_build_leaves(v, area, cx, cy, cz, wave, lane, s_leaf, s_next);

  // ---- integrate leaf by leaf: lane = point of the leaf's n x n rule (n <= 7)
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k;
  Acc4 s;
  s.g = s.h = s.ht = s.e = dc_make(0.0, 0.0);
  for (int lf = 0; lf < nleaf; ++lf) {
    const double* L = s_leaf[wave][lf];
    const double xice = L[0], etce = L[1], fase = L[2];
    const int order = (int)L[3];
    const int off = c_gl_index[order][0], n = c_gl_index[order][1];
    if (lane < n * n) {
      const int a = lane / n, b = lane - a * n;
      const double xio = xice + c_gl_x[off + a] * fase, eto = etce + c_gl_x[off + b] * fase;
      quad_green(v, xio, eto, c_gl_w[off + a] * c_gl_w[off + b] * (fase * fase), cx, cy, cz, nxx, nxy, nxz, k, k2, s);
    }
  }
  s.g.re = wave_sum(s.g.re); s.g.im = wave_sum(s.g.im);
  s.h.re = wave_sum(s.h.re); s.h.im = wave_sum(s.h.im);
  s.ht.re = wave_sum(s.ht.re); s.ht.im = wave_sum(s.ht.im);
  s.e.re = wave_sum(s.e.re); s.e.im = wave_sum(s.e.im);
  if (lane == 0) {
    if (MODE == 1) {
      dc* o = A + 5 * pid;
      o[0] = dc_make((double)nleaf, 0.0); o[1] = s.g; o[2] = s.h; o[3] = s.ht; o[4] = s.e;
    } else if (MODE == 2) A[pid] = bm_coeff(s, g.bc_type[j], ph);
    else A[(long long)g.dof[i] * g.nd + g.dof[j]] = bm_coeff(s, g.bc_type[j], ph);
  }
}

// K3q: singular self term of a quad panel (singular.rs:154-394 with num_nodes = 4): four edges, per edge the line integral
// of the hypersingular part and nsec2 collapsed-square sub-triangles (centre (0,0), edge vertex, edge mid-point; area 1/4 each).
template <int MODE>   // 0: write A[i][i]; 1: probe to out[5 e]; 2: diagonal entry (with free term) to out[e]
__global__ __launch_bounds__(256) void tbem_self_quad_kernel(BemGeom g, BemPhys ph, dc* __restrict__ A) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + wave;
  if (e >= g.np) return;
  if (g.ptype[e] != 4) return;
  const double cx = g.c[0][e], cy = g.c[1][e], cz = g.c[2][e];
  const double nxx = g.nx[0][e], nxy = g.nx[1][e], nxz = g.nx[2][e];
  double v[12]; quad_load(g, e, v);
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k;
  double el = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int b = (a + 1) & 3;
    const double ddx = v[3 * b] - v[3 * a], ddy = v[3 * b + 1] - v[3 * a + 1], ddz = v[3 * b + 2] - v[3 * a + 2];
    el += __builtin_sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
  }
  const double ka = ph.k * (el / 4.0);
  int ngpo1, ngausin, nsec1, nsec2;
  if (ka < 0.3)      { ngpo1 = 3; ngausin = 4; nsec1 = 4;  nsec2 = 2; }
  else if (ka < 1.0) { ngpo1 = 4; ngausin = 5; nsec1 = 6;  nsec2 = 2; }
  else if (ka < 2.0) { ngpo1 = 5; ngausin = 6; nsec1 = 8;  nsec2 = 3; }
  else               { ngpo1 = 6; ngausin = 7; nsec1 = 10; nsec2 = 4; }
  const int eo = c_gl_index[ngpo1][0], so = c_gl_index[ngausin][0];
  const int ne = c_gl_index[ngpo1][1], ns = c_gl_index[ngausin][1];
  const int n_edge_pts = nsec1 * ne;
  const int per_edge = n_edge_pts + nsec2 * ns * ns;
  const int ntask = 4 * per_edge;
  Acc4 s;
  s.g = s.h = s.ht = s.e = dc_make(0.0, 0.0);
  for (int t = lane; t < ntask; t += 64) {
    const int ieg = t / per_edge;
    const int u = t - ieg * per_edge;
    const int ig1 = (ieg + 1) & 3, ig2 = ieg + 4;
    if (u < n_edge_pts) {
      const int isec = u / ne, ig = u - isec * ne;
      const double dpx = v[3 * ig1] - v[3 * ieg], dpy = v[3 * ig1 + 1] - v[3 * ieg + 1], dpz = v[3 * ig1 + 2] - v[3 * ieg + 2];
      const double len = __builtin_sqrt(dpx * dpx + dpy * dpy + dpz * dpz);
      const double ox = dpx / len, oy = dpy / len, oz = dpz / len;
      const double lens = len / (2.0 * (double)nsec1);
      const double delsec = 2.0 / (double)nsec1;
      double secmid = -1.0 - delsec / 2.0;
      for (int q = 0; q <= isec; ++q) secmid += delsec;
      const double sga = secmid + c_gl_x[eo + ig] / (double)nsec1;
      const double wga = c_gl_w[eo + ig] * lens;
      const double f = (sga + 1.0) / 2.0;
      const double dx = (v[3 * ieg] + dpx * f) - cx, dy = (v[3 * ieg + 1] + dpy * f) - cy, dz = (v[3 * ieg + 2] + dpz * f) - cz;
      const double r2 = dx * dx + dy * dy + dz * dz;
      if (r2 >= 1e-30) {
        double r, ri; sqrt_rsqrt(r2, r, ri);
        double sn, cs; sincos_bounded(k * r, sn, cs);
        const double gs = MA_INV4PI * ri;
        const double gre = cs * gs, gim = sn * gs;
        const double fre = -(gre * ri) - gim * k, fim = gre * k - gim * ri;
        const double ux = dx * ri, uy = dy * ri, uz = dz * ri;
        const double wx = uy * oz - uz * oy, wy = uz * ox - ux * oz, wz = ux * oy - uy * ox;
        const double sc = (wx * nxx + wy * nxy + wz * nxz) * wga;
        s.e.re += fre * sc; s.e.im += fim * sc;
      }
    } else {
      const int v2 = u - n_edge_pts;
      const int isec = v2 / (ns * ns);
      const int ij = v2 - isec * ns * ns;
      const int ii = ij / ns, jj = ij - ii * ns;
      const double aresub = 0.25 / (double)nsec2;
      double ss1, ss2, ts1, ts2;
      if (isec == 0) { ss1 = c_csi8[ieg]; ss2 = c_csi8[ig2]; ts1 = c_eta8[ieg]; ts2 = c_eta8[ig2]; }
      else           { ss1 = c_csi8[ig2]; ss2 = c_csi8[ig1]; ts1 = c_eta8[ig2]; ts2 = c_eta8[ig1]; }
      const double sga = c_gl_x[so + ii], tga = c_gl_x[so + jj];
      const double wei = c_gl_w[so + ii] * c_gl_w[so + jj];
      const double sgg = 0.25 * (1.0 + sga) * ((1.0 - tga) * ss1 + (1.0 + tga) * ss2);       // centre (0, 0) drops out
      const double tgg = 0.25 * (1.0 + sga) * ((1.0 - tga) * ts1 + (1.0 + tga) * ts2);
      const QuadPoint q = quad_point(v, sgg, tgg, cx, cy, cz);
      const double wga = wei * (1.0 + sga) * aresub * q.jac;
      const double r2 = q.dx * q.dx + q.dy * q.dy + q.dz * q.dz;
      if (r2 >= 1e-30) {
        double r, ri; sqrt_rsqrt(r2, r, ri);
        double sn, cs; sincos_bounded(k * r, sn, cs);
        const double gs = wga * MA_INV4PI * ri;
        const double gre = cs * gs, gim = sn * gs;
        const double bre = -(gre * ri) - gim * k, bim = gre * k - gim * ri;
        const double a = (q.dx * q.nyx + q.dy * q.nyy + q.dz * q.nyz) * ri;
        const double b = -((q.dx * nxx + q.dy * nxy + q.dz * nxz) * ri);
        const double m = nxx * q.nyx + nxy * q.nyy + nxz * q.nyz;
        s.g.re += gre; s.g.im += gim;
        s.h.re += bre * a; s.h.im += bim * a;
        s.ht.re += bre * b; s.ht.im += bim * b;
        s.e.re += gre * k2 * m; s.e.im += gim * k2 * m;
      }
    }
  }
  s.g = dc_make(wave_sum(s.g.re), wave_sum(s.g.im));
  s.h = dc_make(wave_sum(s.h.re), wave_sum(s.h.im));
  s.ht = dc_make(wave_sum(s.ht.re), wave_sum(s.ht.im));
  s.e = dc_make(wave_sum(s.e.re), wave_sum(s.e.im));
  if (lane != 0) return;
  if (MODE == 1) {
    dc* o = A + 5 * (long long)e;
    o[0] = dc_make((double)ntask, 0.0); o[1] = s.g; o[2] = s.h; o[3] = s.ht; o[4] = s.e;
    return;
  }
  const int bc = g.bc_type[e];
  const dc coeff = bm_coeff(s, bc, ph);
  dc fr = dc_make(0.0, 0.0);
  if (bc == 0) fr = dc_make(-(ph.gamma * 0.5), 0.0);
  else if (bc == 1) fr = dc_make(-(ph.beta_re * ph.tau * 0.5), -(ph.beta_im * ph.tau * 0.5));
  const long long d = g.dof[e];
  if (MODE == 2) A[e] = dc_make(fr.re + coeff.re, fr.im + coeff.im);
  else A[d * g.nd + d] = dc_make(fr.re + coeff.re, fr.im + coeff.im);
}

// ------------------------------------------------------------------ boundary values on the right-hand side
// rhs_contribution of regular.rs:157-177 / singular.rs:360-392 and the free-term share of add_free_terms
// (tbem.rs:273-304), for panels that carry non-zero boundary values (radiation problems). Per quadrature point the
// reference adds  K(point) * zb(point),  K = gamma tau G + beta_p dG/dn_x  (velocity) or -(gamma tau dG/dn_y + beta_p E)
// (pressure),  zb = sum_{a < min(3, len)} bc[a] N_a(point)  -- with a single value only N_0 weighs it -- and
// beta_p = i h / k, the PhysicsParams' own coupling, not the beta the system was built with (regular.rs:168).
// Three passes mirror the matrix kernels (far 13-point rule, subdivided near pairs, singular self term), each leaves a
// deterministic per-row / per-pair partial, and tbem_rhs_finish_kernel adds them up in a fixed order.
__device__ __forceinline__ dc green_point_k(double dx, double dy, double dz, double w4pi, double k, double k2,
                                            double nyx, double nyy, double nyz, double nxx, double nxy, double nxz,
                                            double m, int fbc, double gt, double bp) {
  const double r2 = __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
  if (!(r2 >= 1e-30)) return dc_make(0.0, 0.0);
  double r, ri; sqrt_rsqrt(r2, r, ri);
  double sn, cs; sincos_bounded(k * r, sn, cs);
  const double gsc = w4pi * ri;
  const double gre = cs * gsc, gim = sn * gsc;
  const double bre = -(gre * ri) - gim * k, bim = gre * k - gim * ri;
  const double a = (dx * nyx + dy * nyy + dz * nyz) * ri;
  const double b = -((dx * nxx + dy * nxy + dz * nxz) * ri);
  if (fbc == 0) {                                       // gamma tau zg + (i bp) zht
    const double htre = bre * b, htim = bim * b;
    return dc_make(gt * gre - bp * htim, gt * gim + bp * htre);
  }
  const double rq = a * b, ri2 = ri * ri;               // -(gamma tau zhh + (i bp) ze)
  const double fr = (3.0 * ri2 - k2) * rq + m * ri2, fi = -(k * ri) * (3.0 * rq + m);
  const double ere = gre * fr - gim * fi, eim = gre * fi + gim * fr;
  const double hre = bre * a, him = bim * a;
  return dc_make(-(gt * hre - bp * eim), -(gt * him + bp * ere));
}
__device__ __forceinline__ dc bc_combine(const BemBc& bc, int j, const dc* acc, int nn = 3) {
  const int len = min(bc.len[j], nn);
  dc t = dc_make(0.0, 0.0);
  for (int a = 0; a < len; ++a) { const dc v = bc.val[4 * j + a]; t.re += v.re * acc[a].re - v.im * acc[a].im; t.im += v.re * acc[a].im + v.im * acc[a].re; }
  return t;
}

// far pairs: one wavefront per collocation row, lanes stride over the field panels that carry values
__global__ __launch_bounds__(256) void tbem_rhs_far_kernel(BemGeom g, BemPhys ph, BemBc bc, dc* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= g.np) return;
  const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];
  const double nxx = g.nx[0][i], nxy = g.nx[1][i], nxz = g.nx[2][i];
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k, gt = ph.gamma * ph.tau, bp = ph.tau > 0.0 ? ph.harmonic / ph.k : 0.0;
  double sre = 0.0, sim = 0.0;
  for (int j = lane; j < g.np; j += 64) {
    if (!bc.nz[j] || j == i || (g.nquad > 0 && g.ptype[j] == 4) || pair_is_near(g, i, j)) continue;
    const int fbc = g.bc_type[j];
    const double nyx = g.ny[0][j], nyy = g.ny[1][j], nyz = g.ny[2][j];
    const double e1x = g.e1[0][j], e1y = g.e1[1][j], e1z = g.e1[2][j];
    const double e2x = g.e2[0][j], e2y = g.e2[1][j], e2z = g.e2[2][j];
    const double d0x = g.p0[0][j] - cx, d0y = g.p0[1][j] - cy, d0z = g.p0[2][j] - cz;
    const double jw = g.jac[j] * MA_INV4PI;
    const double m = nxx * nyx + nxy * nyy + nxz * nyz;
    dc acc[3] = {dc_make(0, 0), dc_make(0, 0), dc_make(0, 0)};
    for (int q = 0; q < 13; ++q) {
      const double xi = c_tri13[q][0], eta = c_tri13[q][1], w = c_tri13[q][2];
      const double dx = __builtin_fma(eta, e2x, __builtin_fma(xi, e1x, d0x));
      const double dy = __builtin_fma(eta, e2y, __builtin_fma(xi, e1y, d0y));
      const double dz = __builtin_fma(eta, e2z, __builtin_fma(xi, e1z, d0z));
      const dc kv = green_point_k(dx, dy, dz, w * jw, k, k2, nyx, nyy, nyz, nxx, nxy, nxz, m, fbc, gt, bp);
      const double n0 = 1.0 - xi - eta;
      acc[0].re += n0 * kv.re; acc[0].im += n0 * kv.im;
      acc[1].re += xi * kv.re; acc[1].im += xi * kv.im;
      acc[2].re += eta * kv.re; acc[2].im += eta * kv.im;
    }
    const dc t = bc_combine(bc, j, acc);
    sre += t.re; sim += t.im;
  }
  sre = wave_sum(sre); sim = wave_sum(sim);
  if (lane == 0) out[i] = dc_make(sre, sim);
}

// near pairs: the leaves of tbem_near_kernel, weighted
__global__ __launch_bounds__(256) void tbem_rhs_near_kernel(BemGeom g, BemPhys ph, BemBc bc, const int2* __restrict__ pairs, long long npairs,
                                                            dc* __restrict__ out) {
  __shared__ double s_leaf[4][MA_MAX_LEAVES][6];
  __shared__ double s_next[4][MA_MAX_NSE][6];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long pid = (long long)blockIdx.x * 4 + wave;
  if (pid >= npairs) return;
  const int2 pr = pairs[pid];
  const int i = pr.x, j = pr.y;
  if (!bc.nz[j] || (g.nquad > 0 && g.ptype[j] == 4)) { if (lane == 0) out[pid] = dc_make(0.0, 0.0); return; }   // quad pairs: tbem_rhs_near_quad_kernel overwrites
  const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];
  const double nxx = g.nx[0][i], nxy = g.nx[1][i], nxz = g.nx[2][i];
  double v[9] = {g.p0[0][j], g.p0[1][j], g.p0[2][j], g.p1[0][j], g.p1[1][j], g.p1[2][j], g.p2[0][j], g.p2[1][j], g.p2[2][j]};
  const int nleaf = near_build_leaves(v, g.area[j], cx, cy, cz, wave, lane, s_leaf, s_next);
  const double e1x = v[3] - v[0], e1y = v[4] - v[1], e1z = v[5] - v[2];
  const double e2x = v[6] - v[0], e2y = v[7] - v[1], e2z = v[8] - v[2];
  const double nyx = g.ny[0][j], nyy = g.ny[1][j], nyz = g.ny[2][j];
  const double jw = g.jac[j] * MA_INV4PI;
  const double d0x = v[0] - cx, d0y = v[1] - cy, d0z = v[2] - cz;
  const double m = nxx * nyx + nxy * nyy + nxz * nyz;
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k, gt = ph.gamma * ph.tau, bp = ph.tau > 0.0 ? ph.harmonic / ph.k : 0.0;
  const int fbc = g.bc_type[j];
  dc acc[3] = {dc_make(0, 0), dc_make(0, 0), dc_make(0, 0)};
  const int ntask = nleaf * 13;
  for (int t = lane; t < ntask; t += 64) {
    const int lf = t / 13, q = t - lf * 13;
    const double* L = s_leaf[wave][lf];
    const double a0 = L[0], b0 = L[1], a1 = L[2], b1 = L[3], a2 = L[4], b2 = L[5];
    const double xi = c_tri13[q][0], eta = c_tri13[q][1], w = c_tri13[q][2];
    const double l0 = 1.0 - xi - eta;
    const double xio = a0 * l0 + a1 * xi + a2 * eta;
    const double eto = b0 * l0 + b1 * xi + b2 * eta;
    const double det = __builtin_fabs((a1 - a0) * (b2 - b0) - (a2 - a0) * (b1 - b0));
    const double dx = __builtin_fma(eto, e2x, __builtin_fma(xio, e1x, d0x));
    const double dy = __builtin_fma(eto, e2y, __builtin_fma(xio, e1y, d0y));
    const double dz = __builtin_fma(eto, e2z, __builtin_fma(xio, e1z, d0z));
    const dc kv = green_point_k(dx, dy, dz, w * det * jw, k, k2, nyx, nyy, nyz, nxx, nxy, nxz, m, fbc, gt, bp);
    const double n0 = 1.0 - xio - eto;
    acc[0].re += n0 * kv.re; acc[0].im += n0 * kv.im;
    acc[1].re += xio * kv.re; acc[1].im += xio * kv.im;
    acc[2].re += eto * kv.re; acc[2].im += eto * kv.im;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { acc[a].re = wave_sum(acc[a].re); acc[a].im = wave_sum(acc[a].im); }
  if (lane == 0) out[pid] = bc_combine(bc, j, acc);
}

// self term, velocity values: the collapsed-square sub-triangle points of singular.rs:257-357, weighted (the edge line
// integral carries no boundary value). Pressure values use the finished integrals (singular.rs:380-392): finish kernel.
__global__ __launch_bounds__(256) void tbem_rhs_self_kernel(BemGeom g, BemPhys ph, BemBc bc, dc* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + wave;
  if (e >= g.np) return;
  if (!bc.nz[e] || g.bc_type[e] != 0 || (g.nquad > 0 && g.ptype[e] == 4)) { if (lane == 0) out[e] = dc_make(0.0, 0.0); return; }   // quads: tbem_rhs_self_quad_kernel overwrites
  const double cx = g.c[0][e], cy = g.c[1][e], cz = g.c[2][e];
  const double nxx = g.nx[0][e], nxy = g.nx[1][e], nxz = g.nx[2][e];
  const double P[3][3] = {{g.p0[0][e], g.p0[1][e], g.p0[2][e]}, {g.p1[0][e], g.p1[1][e], g.p1[2][e]}, {g.p2[0][e], g.p2[1][e], g.p2[2][e]}};
  const double nyx = g.ny[0][e], nyy = g.ny[1][e], nyz = g.ny[2][e];
  const double jac = g.jac[e];
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k, gt = ph.gamma * ph.tau, bp = ph.tau > 0.0 ? ph.harmonic / ph.k : 0.0;
  double el = 0.0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int b = (a + 1) % 3;
    double ddx = P[b][0] - P[a][0], ddy = P[b][1] - P[a][1], ddz = P[b][2] - P[a][2];
    el += __builtin_sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
  }
  const double ka = ph.k * (el / 3.0);
  int ngausin, nsec2;
  if (ka < 0.3)      { ngausin = 4; nsec2 = 2; }
  else if (ka < 1.0) { ngausin = 5; nsec2 = 2; }
  else if (ka < 2.0) { ngausin = 6; nsec2 = 3; }
  else               { ngausin = 7; nsec2 = 4; }
  const int so = c_gl_index[ngausin][0], ns = c_gl_index[ngausin][1];
  const int per_edge = nsec2 * ns * ns;
  const double m = nxx * nyx + nxy * nyy + nxz * nyz;
  dc acc[3] = {dc_make(0, 0), dc_make(0, 0), dc_make(0, 0)};
  for (int t = lane; t < 3 * per_edge; t += 64) {
    const int ieg = t / per_edge, v2 = t - ieg * per_edge;
    const int ig1 = (ieg + 1) % 3, ig2 = ieg + 3;
    const int isec = v2 / (ns * ns);
    const int ij = v2 - isec * ns * ns;
    const int ii = ij / ns, jj = ij - ii * ns;
    const double aresub = 1.0 / 24.0 / (double)nsec2;
    const double ss0 = 1.0 / 3.0, ts0 = 1.0 / 3.0;
    double ss1, ss2, ts1, ts2;
    if (isec == 0) { ss1 = c_csi6[ieg]; ss2 = c_csi6[ig2]; ts1 = c_eta6[ieg]; ts2 = c_eta6[ig2]; }
    else           { ss1 = c_csi6[ig2]; ss2 = c_csi6[ig1]; ts1 = c_eta6[ig2]; ts2 = c_eta6[ig1]; }
    const double sga = c_gl_x[so + ii], tga = c_gl_x[so + jj];
    const double wei = c_gl_w[so + ii] * c_gl_w[so + jj];
    const double sgg = 0.5 * (1.0 - sga) * ss0 + 0.25 * (1.0 + sga) * ((1.0 - tga) * ss1 + (1.0 + tga) * ss2);
    const double tgg = 0.5 * (1.0 - sga) * ts0 + 0.25 * (1.0 + sga) * ((1.0 - tga) * ts1 + (1.0 + tga) * ts2);
    const double n0 = 1.0 - sgg - tgg;
    const double dx = (n0 * P[0][0] + sgg * P[1][0] + tgg * P[2][0]) - cx;
    const double dy = (n0 * P[0][1] + sgg * P[1][1] + tgg * P[2][1]) - cy;
    const double dz = (n0 * P[0][2] + sgg * P[1][2] + tgg * P[2][2]) - cz;
    const double wga = wei * (1.0 + sga) * aresub * jac;
    const dc kv = green_point_k(dx, dy, dz, wga * MA_INV4PI, k, k2, nyx, nyy, nyz, nxx, nxy, nxz, m, 0, gt, bp);
    acc[0].re += n0 * kv.re; acc[0].im += n0 * kv.im;
    acc[1].re += sgg * kv.re; acc[1].im += sgg * kv.im;
    acc[2].re += tgg * kv.re; acc[2].im += tgg * kv.im;
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { acc[a].re = wave_sum(acc[a].re); acc[a].im = wave_sum(acc[a].im); }
  if (lane == 0) out[e] = bc_combine(bc, e, acc);
}

// ---- the same three passes for Quad4 field panels that carry values (bilinear N_0..N_3, up to 4 values)
__device__ __forceinline__ void quad_weights(double s, double t, double* n) {
  const double s1 = 0.25 * (s + 1.0), s2 = 0.25 * (s - 1.0), t1 = t + 1.0, t2 = t - 1.0;
  n[0] = s1 * t1; n[1] = -s2 * t1; n[2] = s2 * t2; n[3] = -s1 * t2;
}
__device__ __forceinline__ void quad_point_k(const double* v, double s, double t, double w, double cx, double cy, double cz, double nxx, double nxy,
                                             double nxz, double k, double k2, int fbc, double gt, double bp, dc* acc) {
  const QuadPoint q = quad_point(v, s, t, cx, cy, cz);
  const double m = nxx * q.nyx + nxy * q.nyy + nxz * q.nyz;
  const dc kv = green_point_k(q.dx, q.dy, q.dz, w * q.jac * MA_INV4PI, k, k2, q.nyx, q.nyy, q.nyz, nxx, nxy, nxz, m, fbc, gt, bp);
  double n[4]; quad_weights(s, t, n);
#pragma unroll
  for (int a = 0; a < 4; ++a) { acc[a].re += n[a] * kv.re; acc[a].im += n[a] * kv.im; }
}

// far quad pairs: one wavefront per row, lanes stride over the quad panels; ADDS to out[i] (after tbem_rhs_far_kernel)
__global__ __launch_bounds__(256) void tbem_rhs_far_quad_kernel(BemGeom g, BemPhys ph, BemBc bc, dc* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + wave;
  if (i >= g.np) return;
  const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];
  const double nxx = g.nx[0][i], nxy = g.nx[1][i], nxz = g.nx[2][i];
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k, gt = ph.gamma * ph.tau, bp = ph.tau > 0.0 ? ph.harmonic / ph.k : 0.0;
  const double cs[4] = {1.0, -1.0, -1.0, 1.0}, ct[4] = {1.0, 1.0, -1.0, -1.0};
  double sre = 0.0, sim = 0.0;
  for (int q = lane; q < g.nquad; q += 64) {
    const int j = g.quad_ids[q];
    if (!bc.nz[j] || j == i) continue;
    double v[12]; quad_load(g, j, v);
    double sq;
    {
#pragma clang fp contract(off)
      double arels = g.area[j] * 1.0 * 1.0;
      sq = __builtin_sqrt(arels);
    }
    const double ratio = quad_ratio(cs, ct, v, cx, cy, cz, sq);
    if (ratio < 3.0) continue;                            // near pair: tbem_rhs_near_quad_kernel
    const int order = quad_gauss_order(ratio);
    const int off = c_gl_index[order][0], n = c_gl_index[order][1];
    dc acc[4] = {dc_make(0, 0), dc_make(0, 0), dc_make(0, 0), dc_make(0, 0)};
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b)
        quad_point_k(v, c_gl_x[off + a], c_gl_x[off + b], c_gl_w[off + a] * c_gl_w[off + b], cx, cy, cz, nxx, nxy, nxz, k, k2, g.bc_type[j], gt, bp, acc);
    const dc t = bc_combine(bc, j, acc, 4);
    sre += t.re; sim += t.im;
  }
  sre = wave_sum(sre); sim = wave_sum(sim);
  if (lane == 0) { out[i].re += sre; out[i].im += sim; }
}

__global__ __launch_bounds__(256) void tbem_rhs_near_quad_kernel(BemGeom g, BemPhys ph, BemBc bc, const int2* __restrict__ pairs, long long npairs,
                                                                 dc* __restrict__ out) {
  __shared__ double s_leaf[4][MA_MAX_LEAVES][4];
  __shared__ double s_next[4][MA_MAX_NSE][8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long pid = (long long)blockIdx.x * 4 + wave;
  if (pid >= npairs) return;
  const int2 pr = pairs[pid];
  const int i = pr.x, j = pr.y;
  if (g.ptype[j] != 4 || !bc.nz[j]) return;               // tri pairs / value-free pairs were written by tbem_rhs_near_kernel
  const double cx = g.c[0][i], cy = g.c[1][i], cz = g.c[2][i];
  const double nxx = g.nx[0][i], nxy = g.nx[1][i], nxz = g.nx[2][i];
  double v[12]; quad_load(g, j, v);
  const int nleaf = quad_build_leaves(v, g.area[j], cx, cy, cz, wave, lane, s_leaf, s_next);
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k, gt = ph.gamma * ph.tau, bp = ph.tau > 0.0 ? ph.harmonic / ph.k : 0.0;
  const int fbc = g.bc_type[j];
  dc acc[4] = {dc_make(0, 0), dc_make(0, 0), dc_make(0, 0), dc_make(0, 0)};
  for (int lf = 0; lf < nleaf; ++lf) {
    const double* L = s_leaf[wave][lf];
    const double xice = L[0], etce = L[1], fase = L[2];
    const int order = (int)L[3];
    const int off = c_gl_index[order][0], n = c_gl_index[order][1];
    if (lane < n * n) {
      const int a = lane / n, b = lane - a * n;
      quad_point_k(v, xice + c_gl_x[off + a] * fase, etce + c_gl_x[off + b] * fase, c_gl_w[off + a] * c_gl_w[off + b] * (fase * fase), cx, cy, cz,
                   nxx, nxy, nxz, k, k2, fbc, gt, bp, acc);
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) { acc[a].re = wave_sum(acc[a].re); acc[a].im = wave_sum(acc[a].im); }
  if (lane == 0) out[pid] = bc_combine(bc, j, acc, 4);
}

// self term of a quad with velocity values: the weighted collapsed-square points of tbem_self_quad_kernel
__global__ __launch_bounds__(256) void tbem_rhs_self_quad_kernel(BemGeom g, BemPhys ph, BemBc bc, dc* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + wave;
  if (e >= g.np) return;
  if (g.ptype[e] != 4 || !bc.nz[e] || g.bc_type[e] != 0) return;      // others: tbem_rhs_self_kernel wrote the slot
  const double cx = g.c[0][e], cy = g.c[1][e], cz = g.c[2][e];
  const double nxx = g.nx[0][e], nxy = g.nx[1][e], nxz = g.nx[2][e];
  double v[12]; quad_load(g, e, v);
  const double k = ph.k * ph.harmonic, k2 = ph.k * ph.k, gt = ph.gamma * ph.tau, bp = ph.tau > 0.0 ? ph.harmonic / ph.k : 0.0;
  double el = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int b = (a + 1) & 3;
    const double ddx = v[3 * b] - v[3 * a], ddy = v[3 * b + 1] - v[3 * a + 1], ddz = v[3 * b + 2] - v[3 * a + 2];
    el += __builtin_sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
  }
  const double ka = ph.k * (el / 4.0);
  int ngausin, nsec2;
  if (ka < 0.3)      { ngausin = 4; nsec2 = 2; }
  else if (ka < 1.0) { ngausin = 5; nsec2 = 2; }
  else if (ka < 2.0) { ngausin = 6; nsec2 = 3; }
  else               { ngausin = 7; nsec2 = 4; }
  const int so = c_gl_index[ngausin][0], ns = c_gl_index[ngausin][1];
  const int per_edge = nsec2 * ns * ns;
  dc acc[4] = {dc_make(0, 0), dc_make(0, 0), dc_make(0, 0), dc_make(0, 0)};
  for (int t = lane; t < 4 * per_edge; t += 64) {
    const int ieg = t / per_edge, v2 = t - ieg * per_edge;
    const int ig1 = (ieg + 1) & 3, ig2 = ieg + 4;
    const int isec = v2 / (ns * ns);
    const int ij = v2 - isec * ns * ns;
    const int ii = ij / ns, jj = ij - ii * ns;
    const double aresub = 0.25 / (double)nsec2;
    double ss1, ss2, ts1, ts2;
    if (isec == 0) { ss1 = c_csi8[ieg]; ss2 = c_csi8[ig2]; ts1 = c_eta8[ieg]; ts2 = c_eta8[ig2]; }
    else           { ss1 = c_csi8[ig2]; ss2 = c_csi8[ig1]; ts1 = c_eta8[ig2]; ts2 = c_eta8[ig1]; }
    const double sga = c_gl_x[so + ii], tga = c_gl_x[so + jj];
    const double wei = c_gl_w[so + ii] * c_gl_w[so + jj];
    const double sgg = 0.25 * (1.0 + sga) * ((1.0 - tga) * ss1 + (1.0 + tga) * ss2);
    const double tgg = 0.25 * (1.0 + sga) * ((1.0 - tga) * ts1 + (1.0 + tga) * ts2);
    quad_point_k(v, sgg, tgg, wei * (1.0 + sga) * aresub, cx, cy, cz, nxx, nxy, nxz, k, k2, 0, gt, bp, acc);
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) { acc[a].re = wave_sum(acc[a].re); acc[a].im = wave_sum(acc[a].im); }
  if (lane == 0) out[e] = bc_combine(bc, e, acc, 4);
}

// rhs[dof_i] = free-term share + far[i] + sum of the row's near pairs (in list order) + self[i]; `self5` holds the raw
// self integrals {count, G, H, H^T, E} of every panel (tbem_self_kernel<1>) for the pressure-value self term.
__global__ __launch_bounds__(256) void tbem_rhs_finish_kernel(BemGeom g, BemPhys ph, BemBc bc, const dc* __restrict__ far, const dc* __restrict__ near,
                                                              const long long* __restrict__ pair_off, const dc* __restrict__ selfv,
                                                              const dc* __restrict__ self5, dc* __restrict__ rhs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.np) return;
  const int bt = g.bc_type[i];
  dc avg = dc_make(0.0, 0.0);
  if (bt <= 1) {
    const int len = bc.len[i];
    for (int a = 0; a < len; ++a) { avg.re += bc.val[4 * i + a].re; avg.im += bc.val[4 * i + a].im; }
    avg.re /= (double)len; avg.im /= (double)len;
  }
  dc t = dc_make(0.0, 0.0);
  if (bt == 0) {                                          // avg * beta * tau / 2 (tbem.rs:288)
    const dc ab = avg * dc_make(ph.beta_re, ph.beta_im);
    t = dc_make(ab.re * ph.tau * 0.5, ab.im * ph.tau * 0.5);
  } else if (bt == 1) t = dc_make(avg.re * ph.tau * 0.5, avg.im * ph.tau * 0.5);   // avg * tau / 2 (tbem.rs:297)
  t.re += far[i].re; t.im += far[i].im;
  for (long long p = pair_off[i]; p < pair_off[i + 1]; ++p) { t.re += near[p].re; t.im += near[p].im; }
  if (bc.nz[i]) {
    if (bt == 0) { t.re += selfv[i].re; t.im += selfv[i].im; }
    else if (bt == 1) {                                   // -(gamma tau dG/dn_y + beta_p E) * avg with the finished integrals
      const double gt = ph.gamma * ph.tau, bp = ph.tau > 0.0 ? ph.harmonic / ph.k : 0.0;
      const dc h = self5[5 * (long long)i + 2], ee = self5[5 * (long long)i + 4];
      const dc kk = dc_make(-(gt * h.re - bp * ee.im), -(gt * h.im + bp * ee.re));
      const dc c = kk * avg;
      t.re += c.re; t.im += c.im;
    }
  }
  rhs[g.dof[i]] = t;
}

// ------------------------------------------------------------------ right-hand sides
__global__ void fill_zero_kernel(dc* __restrict__ v, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = dc_make(0.0, 0.0);
}

// IncidentField::compute_rhs_with_beta (incident.rs:317-342), plane wave (:103-117,188-207) or
// point source (:119-133,209-233); rhs = -(gamma p + beta tau dp/dn)
__global__ void incident_rhs_kernel(BemGeom g, BemPhys ph, int kind, double vx, double vy, double vz,
                                    double are, double aim, int accumulate, dc* __restrict__ rhs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.np) return;
  const double px = g.c[0][i], py = g.c[1][i], pz = g.c[2][i];
  const double nx = g.nx[0][i], ny = g.nx[1][i], nz = g.nx[2][i];
  const double k = ph.k;
  dc p = dc_make(0.0, 0.0), d = dc_make(0.0, 0.0);
  if (kind == 0) {
    double kdx = k * (vx * px + vy * py + vz * pz);
    double kdn = k * (vx * nx + vy * ny + vz * nz);
    double sn, cs; sincos(kdx, &sn, &cs);
    p = dc_make(are, aim) * dc_make(cs, sn);
    d = dc_make(0.0, kdn) * p;
  } else {
    double dx = px - vx, dy = py - vy, dz = pz - vz;
    double r = __builtin_sqrt(dx * dx + dy * dy + dz * dz);
    if (r > 1e-10) {
      double kr = k * r;
      double sn, cs; sincos(kr, &sn, &cs);
      double den = 4.0 * 3.14159265358979323846 * r;
      dc gg = dc_make(cs / den, sn / den);
      p = dc_make(are, aim) * gg;
      dc dgdr = dc_make(-1.0 / r, k) * gg;
      double drdn = (dx * nx + dy * ny + dz * nz) / r;
      d = (dc_make(are, aim) * dgdr) * drdn;
    }
  }
  dc bt = dc_make(ph.beta_re * ph.tau, ph.beta_im * ph.tau);
  dc v = dc_neg(p * ph.gamma + bt * d);
  const long long di = g.dof[i];
  if (accumulate) v = v + rhs[di];
  rhs[di] = v;
}

// ------------------------------------------------------------------ launchers
int bem_launch_near_list(const BemGeom& g, int pass, int* counts, const long long* offsets, int2* pairs, hipStream_t st) {
  dim3 grid((g.np + 3) / 4), block(256);
  hipLaunchKernelGGL(near_list_kernel, grid, block, 0, st, g, pass, counts, offsets, pairs);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int bem_far_row_strips(const BemGeom& g) { return (g.np + 31) / 32; }
// strips [blk0, blk0 + nblk) of 32 collocation rows (nblk < 0: all of them)
int bem_launch_far_multi(const BemGeom& g, int nf, const BemPhys* phs, c64* const* As, hipStream_t st, int blk0, int nblk) {
  MA_REQUIRE(nf >= 1 && nf <= 3, MA_ERR_INVALID, "1..3 systems per far pass");
  const int rpb = 32;
  const int all = (g.np + rpb - 1) / rpb;
  if (nblk < 0) { blk0 = 0; nblk = all; }
  MA_REQUIRE(blk0 >= 0 && blk0 + nblk <= all, MA_ERR_INVALID, "row strips [%d, %d) of %d", blk0, blk0 + nblk, all);
  if (nblk == 0) return MA_OK;
  dim3 grid((g.np + 255) / 256, nblk), block(256);
  FarMulti fm{};
  for (int f = 0; f < 3; ++f) { fm.ph[f] = phs[f < nf ? f : 0]; fm.A[f] = reinterpret_cast<dc*>(As[f < nf ? f : 0]); }
  const bool vel = g.all_velocity != 0;
  if (nf == 1) { if (vel) hipLaunchKernelGGL((tbem_far_kernel<1, true>), grid, block, 0, st, g, fm, rpb, blk0); else hipLaunchKernelGGL((tbem_far_kernel<1, false>), grid, block, 0, st, g, fm, rpb, blk0); }
  else if (nf == 2) { if (vel) hipLaunchKernelGGL((tbem_far_kernel<2, true>), grid, block, 0, st, g, fm, rpb, blk0); else hipLaunchKernelGGL((tbem_far_kernel<2, false>), grid, block, 0, st, g, fm, rpb, blk0); }
  else { if (vel) hipLaunchKernelGGL((tbem_far_kernel<3, true>), grid, block, 0, st, g, fm, rpb, blk0); else hipLaunchKernelGGL((tbem_far_kernel<3, false>), grid, block, 0, st, g, fm, rpb, blk0); }
  if (g.nquad > 0) for (int f = 0; f < nf; ++f)
    hipLaunchKernelGGL(tbem_far_quad_kernel, dim3((g.nquad + 255) / 256, nblk), block, 0, st, g, phs[f], reinterpret_cast<dc*>(As[f]), rpb, blk0);
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int bem_launch_far(const BemGeom& g, const BemPhys& ph, c64* A, hipStream_t st) { return bem_launch_far_multi(g, 1, &ph, &A, st, 0, -1); }

int bem_launch_near(const BemGeom& g, const BemPhys& ph, const int2* pairs, long long npairs, c64* A, hipStream_t st) {
  if (npairs <= 0) return MA_OK;
  dim3 grid((unsigned)((npairs + 3) / 4)), block(256);
  hipLaunchKernelGGL(tbem_near_kernel<0>, grid, block, 0, st, g, ph, pairs, npairs, reinterpret_cast<dc*>(A));
  if (g.nquad > 0) hipLaunchKernelGGL(tbem_near_quad_kernel<0>, grid, block, 0, st, g, ph, pairs, npairs, reinterpret_cast<dc*>(A));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// the near pairs of cnt <= 3 systems of one mesh: one pass over the leaves and the quadrature points' geometry (Tri3 field panels);
// Quad4 field panels per system as before
int bem_launch_near_multi(const BemGeom& g, int cnt, const BemPhys* ph, c64* const* As, const int2* pairs, long long npairs, hipStream_t st) {
  if (npairs <= 0 || cnt <= 0) return MA_OK;
  MA_REQUIRE(cnt <= 3, MA_ERR_INVALID, "at most three systems per pass");
  if (cnt == 1) return bem_launch_near(g, ph[0], pairs, npairs, As[0], st);
  FarMulti fm;
  for (int f = 0; f < 3; ++f) { fm.ph[f] = ph[f < cnt ? f : 0]; fm.A[f] = reinterpret_cast<dc*>(As[f < cnt ? f : 0]); }
  dim3 grid((unsigned)((npairs + 3) / 4)), block(256);
  if (cnt == 2) hipLaunchKernelGGL(tbem_near_multi_kernel<2>, grid, block, 0, st, g, fm, pairs, npairs);
  else hipLaunchKernelGGL(tbem_near_multi_kernel<3>, grid, block, 0, st, g, fm, pairs, npairs);
  if (g.nquad > 0) for (int f = 0; f < cnt; ++f) hipLaunchKernelGGL(tbem_near_quad_kernel<0>, grid, block, 0, st, g, ph[f], pairs, npairs, reinterpret_cast<dc*>(As[f]));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int bem_launch_probe_pairs(const BemGeom& g, const BemPhys& ph, const int2* pairs, long long npairs, c64* out5, hipStream_t st) {
  if (npairs <= 0) return MA_OK;
  dim3 grid((unsigned)((npairs + 3) / 4)), block(256);
  hipLaunchKernelGGL(tbem_near_kernel<1>, grid, block, 0, st, g, ph, pairs, npairs, reinterpret_cast<dc*>(out5));
  if (g.nquad > 0) hipLaunchKernelGGL(tbem_near_quad_kernel<1>, grid, block, 0, st, g, ph, pairs, npairs, reinterpret_cast<dc*>(out5));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int bem_launch_probe_self(const BemGeom& g, const BemPhys& ph, c64* out5, hipStream_t st) {
  dim3 grid((g.np + 3) / 4), block(256);
  hipLaunchKernelGGL(tbem_self_kernel<1>, grid, block, 0, st, g, ph, reinterpret_cast<dc*>(out5));
  if (g.nquad > 0) hipLaunchKernelGGL(tbem_self_quad_kernel<1>, grid, block, 0, st, g, ph, reinterpret_cast<dc*>(out5));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int bem_launch_self(const BemGeom& g, const BemPhys& ph, c64* A, hipStream_t st) {
  dim3 grid((g.np + 3) / 4), block(256);
  hipLaunchKernelGGL(tbem_self_kernel<0>, grid, block, 0, st, g, ph, reinterpret_cast<dc*>(A));
  if (g.nquad > 0) hipLaunchKernelGGL(tbem_self_quad_kernel<0>, grid, block, 0, st, g, ph, reinterpret_cast<dc*>(A));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// Quad4 columns of the matrix-free operator (no-ops on a Tri3 mesh)
int bem_quad_strips(const BemGeom& g) { return (g.nquad + 255) / 256; }
int bem_launch_quad_matvec(const BemGeom& g, const BemPhys& ph, int row0, int row1, int rows_per_block, const c64* x, c64* partial_quad_strips, hipStream_t st) {
  if (g.nquad <= 0 || row1 <= row0) return MA_OK;
  dim3 grid(bem_quad_strips(g), (row1 - row0 + rows_per_block - 1) / rows_per_block), block(256);
  hipLaunchKernelGGL(tbem_matvec_quad_kernel, grid, block, 0, st, g, ph, row0, row1, rows_per_block, reinterpret_cast<const dc*>(x), reinterpret_cast<dc*>(partial_quad_strips));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int bem_launch_quad_matvec_t(const BemGeom& g, const BemPhys& ph, int row0, int row1, int nchunks, int chunk_rows, const c64* x, c64* partial, hipStream_t st) {
  if (g.nquad <= 0 || row1 <= row0) return MA_OK;
  dim3 grid(bem_quad_strips(g), nchunks), block(256);
  hipLaunchKernelGGL(tbem_matvec_quad_t_kernel, grid, block, 0, st, g, ph, row0, row1, chunk_rows, reinterpret_cast<const dc*>(x), reinterpret_cast<dc*>(partial));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int bem_launch_quad_pairs_far(const BemGeom& g, const BemPhys& ph, const int2* pairs, long long npairs, c64* out, hipStream_t st) {
  if (g.nquad <= 0 || npairs <= 0) return MA_OK;
  hipLaunchKernelGGL(tbem_pairs_quad_far_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, g, ph, pairs, npairs, reinterpret_cast<dc*>(out));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// list forms for the on-the-fly operator: coefficient of every listed near pair / of every diagonal entry
int bem_launch_near_list_values(const BemGeom& g, const BemPhys& ph, const int2* pairs, long long npairs, c64* out, hipStream_t st) {
  if (npairs <= 0) return MA_OK;
  dim3 grid((unsigned)((npairs + 3) / 4)), block(256);
  hipLaunchKernelGGL(tbem_near_kernel<2>, grid, block, 0, st, g, ph, pairs, npairs, reinterpret_cast<dc*>(out));
  if (g.nquad > 0) hipLaunchKernelGGL(tbem_near_quad_kernel<2>, grid, block, 0, st, g, ph, pairs, npairs, reinterpret_cast<dc*>(out));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
int bem_launch_self_list_values(const BemGeom& g, const BemPhys& ph, c64* out, hipStream_t st) {
  dim3 grid((g.np + 3) / 4), block(256);
  hipLaunchKernelGGL(tbem_self_kernel<2>, grid, block, 0, st, g, ph, reinterpret_cast<dc*>(out));
  if (g.nquad > 0) hipLaunchKernelGGL(tbem_self_quad_kernel<2>, grid, block, 0, st, g, ph, reinterpret_cast<dc*>(out));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// right-hand side from the panels' boundary values: scratch = far[np] | self[np] | self5[5 np] | near[npairs]
int bem_launch_rhs_bc(const BemGeom& g, const BemPhys& ph, const BemBc& bc, const int2* pairs, const long long* pair_off, long long npairs,
                      c64* scratch, c64* rhs, hipStream_t st) {
  dc* far = reinterpret_cast<dc*>(scratch); dc* selfv = far + g.np; dc* self5 = selfv + g.np; dc* near = self5 + 5 * (size_t)g.np;
  dim3 rows((g.np + 3) / 4), block(256);
  hipLaunchKernelGGL(tbem_rhs_far_kernel, rows, block, 0, st, g, ph, bc, far);
  if (npairs > 0) hipLaunchKernelGGL(tbem_rhs_near_kernel, dim3((unsigned)((npairs + 3) / 4)), block, 0, st, g, ph, bc, pairs, npairs, near);
  hipLaunchKernelGGL(tbem_rhs_self_kernel, rows, block, 0, st, g, ph, bc, selfv);
  hipLaunchKernelGGL(tbem_self_kernel<1>, rows, block, 0, st, g, ph, self5);
  if (g.nquad > 0) {
    hipLaunchKernelGGL(tbem_rhs_far_quad_kernel, rows, block, 0, st, g, ph, bc, far);
    if (npairs > 0) hipLaunchKernelGGL(tbem_rhs_near_quad_kernel, dim3((unsigned)((npairs + 3) / 4)), block, 0, st, g, ph, bc, pairs, npairs, near);
    hipLaunchKernelGGL(tbem_rhs_self_quad_kernel, rows, block, 0, st, g, ph, bc, selfv);
    hipLaunchKernelGGL(tbem_self_quad_kernel<1>, rows, block, 0, st, g, ph, self5);
  }
  hipLaunchKernelGGL(tbem_rhs_finish_kernel, dim3((g.np + 255) / 256), block, 0, st, g, ph, bc, far, near, pair_off, selfv, self5, reinterpret_cast<dc*>(rhs));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int bem_launch_zero(c64* v, int n, hipStream_t st) {
  if (n <= 0) return MA_OK;
  hipLaunchKernelGGL(fill_zero_kernel, dim3((n + 255) / 256), dim3(256), 0, st, reinterpret_cast<dc*>(v), n);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int bem_launch_incident(const BemGeom& g, const BemPhys& ph, int kind, const double* v, double are, double aim,
                        int accumulate, c64* rhs, hipStream_t st) {
  hipLaunchKernelGGL(incident_rhs_kernel, dim3((g.np + 255) / 256), dim3(256), 0, st, g, ph, kind, v[0], v[1], v[2],
                     are, aim, accumulate, reinterpret_cast<dc*>(rhs));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

}  // namespace ma
