// fmm_plan.hip — SlfmmSystem (math-bem/src/core/assembly/slfmm.rs) on the device: A = [N] + [S][D][T].
//
//   build_slfmm_system (:417-470)   slfmm_create: the near blocks come from the TBEM near / self kernels (one wavefront per
//                                   element pair, bem_kernels.hip) with the coefficient of compute_near_block (:583-584:
//                                   dg_dn gamma tau + d2g beta, beta = PhysicsParams::burton_miller_beta(), no sign switch)
//                                   and the free terms of :514-533; the D entries (h_0(k r) i k per far pair, :693-714) on
//                                   the host with the reference's spherical_hankel_first_kind; T and S (:615-656, :724-765)
//                                   are NOT stored: their entries w_p exp(-/+ i k s_p . (x_j - C)) are two FMAs and a sincos
//                                   from data that is resident anyway, recomputed inside the apply kernels.
//   matvec / matvec_transpose       slfmm_apply: four launches -- near blocks (one workgroup per cluster, a wavefront per
//   (:150-257, :262-376)            row, every contribution of a row summed in one fixed order: no atomics), multipoles
//                                   (T x, or S^T x), translation over the far pairs, evaluation (S l, or T^T m).
// Meshes without evaluation elements, velocity-type boundary conditions (the coefficient of compute_near_block is the
// velocity one whatever the panel's type), every panel in at most one cluster; anything else is MA_ERR_UNSUPPORTED.
#include "fmm_plan.hpp"
#include "ma_device_math.hpp"
#include "ma_tables.h"
#include <vector>
#include <cmath>
#include <algorithm>
#include <new>

using namespace ma;

struct SlfmmEntry { long long boff; int other; int tflag; };   // a near block seen from one of its two clusters

struct ma_slfmm {
  int device = 0; ma_bem_plan* plan = nullptr;
  long long n = 0; int nc = 0, P = 0; double k = 0.0;
  int* d_eptr = nullptr; int* d_eidx = nullptr; int* d_edof = nullptr;
  double* d_cc = nullptr; double* d_sc = nullptr; double* d_sw = nullptr;
  c64* d_bval = nullptr; long long nbval = 0;
  int* d_cptr = nullptr; SlfmmEntry* d_cent = nullptr;
  int* d_fptr = nullptr; int* d_foth = nullptr; c64* d_fval = nullptr;      // far pairs grouped by field cluster (forward)
  int* d_tptr = nullptr; int* d_toth = nullptr; c64* d_tval = nullptr;      // grouped by source cluster (transpose)
  c64* d_up = nullptr; c64* d_tr = nullptr;
  // host copies for extract_near_field_matrix
  std::vector<int> h_eptr, h_edof, h_bsrc, h_bfld; std::vector<long long> h_boff;
};

namespace {

// ---- reference arithmetic on the host: spherical_hankel_first_kind (math-wave/src/special/spherical.rs:165-246), order >= 2
void spherical_hankel_first_kind(int order, double x, double harmonic, std::vector<double>& re, std::vector<double>& im) {
  re.assign((size_t)order, 0.0); im.assign((size_t)order, 0.0);
  const double cos_x = std::cos(x), sin_x = std::sin(x);
  im[0] = -cos_x / x; im[1] = -(cos_x / x + sin_x) / x;
  for (int n = 2; n < order; ++n) im[(size_t)n] = (double)(2 * n - 1) / x * im[(size_t)n - 1] - im[(size_t)n - 2];
  const double nu = (double)(order - 1);
  double di = (2.0 * (nu + 1.0) + 1.0) / x, cj = di, dj = 0.0, err = 1.0;
  for (int j = 1; err > 1e-9; ++j) {
    const double bj = (2.0 * (nu + (double)j + 1.0) + 1.0) / x;
    dj = bj - dj; if (dj == 0.0) dj = 1e-30; dj = 1.0 / dj;
    cj = bj - 1.0 / cj; if (cj == 0.0) cj = 1e-30;
    di = di * cj * dj;
    err = std::fabs(cj * dj - 1.0);
    if (j + 1 > 1000) break;
  }
  const double gnu = nu / x - 1.0 / di;
  std::vector<double> gg((size_t)order, 0.0), dg((size_t)order, 0.0);
  gg[(size_t)order - 1] = 1.0; dg[(size_t)order - 1] = gnu;
  for (int i = order - 2; i >= 0; --i) {
    gg[(size_t)i] = ((double)i + 2.0) / x * gg[(size_t)i + 1] + dg[(size_t)i + 1];
    dg[(size_t)i] = (double)i / x * gg[(size_t)i] - gg[(size_t)i + 1];
  }
  const double dp = std::fabs(gg[0]) > 1e-5 ? sin_x / x / gg[0] : (cos_x - sin_x / x) / x / dg[0];
  for (int n = 0; n < order; ++n) { re[(size_t)n] = dp * gg[(size_t)n]; im[(size_t)n] *= harmonic; }
}

// near blocks: every row of cluster c sums all its contributions in the order of the cluster's entry list
__global__ __launch_bounds__(256) void slfmm_near_kernel(const int* __restrict__ eptr, const int* __restrict__ edof, const int* __restrict__ cptr,
                                                         const SlfmmEntry* __restrict__ cent, const dc* __restrict__ bval, const dc* __restrict__ x,
                                                         dc* __restrict__ y, int tmode) {
  const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e0 = eptr[c], nc_ = eptr[c + 1] - e0;
  for (int i = wave; i < nc_; i += 4) {
    double sr = 0.0, si = 0.0;
    for (int q = cptr[c]; q < cptr[c + 1]; ++q) {
      const SlfmmEntry en = cent[q];
      const int o0 = eptr[en.other], no = eptr[en.other + 1] - o0;
      const bool self = en.other == c;
      const bool tr = (en.tflag != 0) != (self && tmode != 0);
      const dc* B = bval + en.boff;
      for (int j = lane; j < no; j += 64) {
        const dc b = tr ? B[(long long)j * nc_ + i] : B[(long long)i * no + j];
        const dc xv = x[edof[o0 + j]];
        sr += b.re * xv.re - b.im * xv.im; si += b.re * xv.im + b.im * xv.re;
      }
    }
    sr = wave_sum(sr); si = wave_sum(si);
    if (lane == 0) y[edof[e0 + i]] = dc_make(sr, si);
  }
}
// up[c][p] = sum_j w_p exp(i sgn k s_p . (x_j - C_c)) x[dof_j]     (sgn -1: T x; +1: S^T x)
__global__ __launch_bounds__(256) void slfmm_up_kernel(BemGeom g, const int* __restrict__ eptr, const int* __restrict__ eidx, const int* __restrict__ edof,
                                                       const double* __restrict__ cc, const double* __restrict__ sc, const double* __restrict__ sw,
                                                       int P, double k, double sgn, const dc* __restrict__ x, dc* __restrict__ up) {
  const int c = blockIdx.x;
  const int e0 = eptr[c], n = eptr[c + 1] - e0;
  const double Cx = cc[3 * c], Cy = cc[3 * c + 1], Cz = cc[3 * c + 2];
  for (int p = threadIdx.x; p < P; p += 256) {
    const double sx = sc[3 * p], sy = sc[3 * p + 1], sz = sc[3 * p + 2], w = sw[p];
    double sr = 0.0, si = 0.0;
    for (int j = 0; j < n; ++j) {
      const int e = eidx[e0 + j];
      const double sd = sx * (g.c[0][e] - Cx) + sy * (g.c[1][e] - Cy) + sz * (g.c[2][e] - Cz);
      double sn, cs; sincos(k * sd, &sn, &cs);
      const double er = cs * w, ei = sgn * sn * w;
      const dc xv = x[edof[e0 + j]];
      sr += er * xv.re - ei * xv.im; si += er * xv.im + ei * xv.re;
    }
    up[(long long)c * P + p] = dc_make(sr, si);
  }
}
// tr[c][p] = sum over the cluster's far partners of d * up[other][p]   (the diagonal D entry is the same for every p)
__global__ __launch_bounds__(256) void slfmm_translate_kernel(const int* __restrict__ fptr, const int* __restrict__ foth, const dc* __restrict__ fval, int P,
                                                              const dc* __restrict__ up, dc* __restrict__ tr) {
  const int c = blockIdx.x;
  for (int p = threadIdx.x; p < P; p += 256) {
    double sr = 0.0, si = 0.0;
    for (int q = fptr[c]; q < fptr[c + 1]; ++q) {
      const dc d = fval[q]; const dc m = up[(long long)foth[q] * P + p];
      sr += d.re * m.re - d.im * m.im; si += d.re * m.im + d.im * m.re;
    }
    tr[(long long)c * P + p] = dc_make(sr, si);
  }
}
// y[dof_j] += sum_p w_p exp(i sgn k s_p . (x_j - C_c)) tr[c][p]     (sgn +1: S l; -1: T^T m)
__global__ __launch_bounds__(256) void slfmm_down_kernel(BemGeom g, const int* __restrict__ eptr, const int* __restrict__ eidx, const int* __restrict__ edof,
                                                         const double* __restrict__ cc, const double* __restrict__ sc, const double* __restrict__ sw,
                                                         int P, double k, double sgn, const dc* __restrict__ tr, dc* __restrict__ y) {
  const int c = blockIdx.x;
  const int e0 = eptr[c], n = eptr[c + 1] - e0;
  const double Cx = cc[3 * c], Cy = cc[3 * c + 1], Cz = cc[3 * c + 2];
  for (int j = threadIdx.x; j < n; j += 256) {
    const int e = eidx[e0 + j];
    const double dx = g.c[0][e] - Cx, dy = g.c[1][e] - Cy, dz = g.c[2][e] - Cz;
    double sr = 0.0, si = 0.0;
    for (int p = 0; p < P; ++p) {
      const double sd = sc[3 * p] * dx + sc[3 * p + 1] * dy + sc[3 * p + 2] * dz;
      double sn, cs; sincos(k * sd, &sn, &cs);
      const double w = sw[p], er = cs * w, ei = sgn * sn * w;
      const dc l = tr[(long long)c * P + p];
      sr += er * l.re - ei * l.im; si += er * l.im + ei * l.re;
    }
    dc* o = y + edof[e0 + j];
    o->re += sr; o->im += si;
  }
}
// the self blocks' diagonal: singular integral + SLFMM free term. self_vals = coefficient - gamma/2 (TBEM's free term): + gamma
__global__ void slfmm_fix_diag_kernel(int n, const long long* __restrict__ pos, const int* __restrict__ panel, const dc* __restrict__ selfv, double gamma,
                                      dc* __restrict__ bval) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  const dc v = selfv[panel[q]];
  bval[pos[q]] = dc_make(v.re + gamma, v.im);
}
__global__ void slfmm_scatter_block_kernel(const dc* __restrict__ B, int ns, int nf, const int* __restrict__ sd, const int* __restrict__ fd, long long n, int sym,
                                           dc* __restrict__ A) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)ns * nf) return;
  const int i = (int)(t / nf), j = (int)(t - (long long)i * nf);
  const dc b = B[t];
  dc* a = A + (long long)sd[i] * n + fd[j];
  a->re += b.re; a->im += b.im;
  if (sym) { dc* a2 = A + (long long)fd[j] * n + sd[i]; a2->re += b.re; a2->im += b.im; }
}

template <typename T> int upload(T** d, const std::vector<T>& h) {
  const size_t n = h.empty() ? 1 : h.size();
  MA_HIP(hipMalloc(d, sizeof(T) * n));
  if (!h.empty()) MA_HIP(hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  return MA_OK;
}

}  // namespace

long long slfmm_num_dofs(const ma_slfmm* S) { return S->n; }
int slfmm_device(const ma_slfmm* S) { return S->device; }

void slfmm_destroy(ma_slfmm* S) {
  if (!S) return;
  (void)hipSetDevice(S->device);
  void* p[] = {S->d_eptr, S->d_eidx, S->d_edof, S->d_cc, S->d_sc, S->d_sw, S->d_bval, S->d_cptr, S->d_cent, S->d_fptr, S->d_foth, S->d_fval, S->d_tptr, S->d_toth,
               S->d_tval, S->d_up, S->d_tr};
  for (void* q : p) if (q) (void)hipFree(q);
  delete S;
}

int slfmm_create(ma_bem_plan* plan, const ma_clusters_t* cl, const ma_physics_t* physics, int n_theta, int n_phi, int n_terms, ma_slfmm** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(plan && cl && physics, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(cl->n_clusters >= 1 && cl->center && cl->elem_ptr && cl->elem_idx && cl->near_ptr && cl->far_ptr, MA_ERR_INVALID, "incomplete cluster lists");
  MA_REQUIRE(n_theta >= 1 && n_theta <= 20 && mat_gl_index[n_theta][1] == n_theta, MA_ERR_INVALID,
             "n_theta = %d is not a tabulated Gauss-Legendre order (gauss.rs:27-60 would hand back another rule and the reference's sphere-point count would not match)", n_theta);
  MA_REQUIRE(n_phi >= 1 && n_phi <= 4096 && n_terms >= 0, MA_ERR_INVALID, "bad n_phi / n_terms");
  const int nc = cl->n_clusters, np = plan->np;
  MA_REQUIRE(plan->geom.nquad >= 0, MA_ERR_INVALID, "bad plan");
  MA_HIP(hipSetDevice(plan->device));
  // every panel in at most one cluster; velocity-type boundary conditions
  std::vector<unsigned char> bct((size_t)np), seen((size_t)np, 0);
  MA_HIP(hipMemcpy(bct.data(), plan->geom.bc_type, (size_t)np, hipMemcpyDeviceToHost));
  std::vector<int> hdof((size_t)np);
  MA_HIP(hipMemcpy(hdof.data(), plan->geom.dof, sizeof(int) * (size_t)np, hipMemcpyDeviceToHost));
  MA_REQUIRE(cl->elem_ptr[0] == 0 && cl->near_ptr[0] == 0 && cl->far_ptr[0] == 0, MA_ERR_INVALID, "list offsets must start at 0");
  for (int c = 0; c < nc; ++c) {
    MA_REQUIRE(cl->elem_ptr[c + 1] >= cl->elem_ptr[c] && cl->near_ptr[c + 1] >= cl->near_ptr[c] && cl->far_ptr[c + 1] >= cl->far_ptr[c], MA_ERR_INVALID, "cluster %d: decreasing offsets", c);
    for (int q = cl->elem_ptr[c]; q < cl->elem_ptr[c + 1]; ++q) {
      const int e = cl->elem_idx[q];
      MA_REQUIRE(e >= 0 && e < np, MA_ERR_INVALID, "cluster %d lists element %d outside 0..%d (meshes with evaluation elements are not supported here)", c, e, np - 1);
      MA_REQUIRE(!seen[(size_t)e], MA_ERR_UNSUPPORTED, "element %d belongs to more than one cluster", e);
      MA_REQUIRE(bct[(size_t)e] == 0, MA_ERR_UNSUPPORTED, "element %d: only velocity-type boundary conditions (compute_near_block's coefficient, slfmm.rs:583-584)", e);
      seen[(size_t)e] = 1;
    }
    for (int q = cl->near_ptr[c]; q < cl->near_ptr[c + 1]; ++q) MA_REQUIRE(cl->near_idx && cl->near_idx[q] >= 0 && cl->near_idx[q] < nc, MA_ERR_INVALID, "near cluster index out of range");
    for (int q = cl->far_ptr[c]; q < cl->far_ptr[c + 1]; ++q) MA_REQUIRE(cl->far_idx && cl->far_idx[q] >= 0 && cl->far_idx[q] < nc, MA_ERR_INVALID, "far cluster index out of range");
  }
  ma_slfmm* S = new (std::nothrow) ma_slfmm(); MA_REQUIRE(S, MA_ERR_NOMEM, "host allocation failed");
  S->device = plan->device; S->plan = plan; S->n = plan->nd; S->nc = nc; S->P = n_theta * n_phi; S->k = physics->wave_number;
  auto fail = [&](int code) { slfmm_destroy(S); return code; };
  const int P = S->P;
  // ---- unit_sphere_quadrature (gauss.rs:110-130)
  std::vector<double> sc((size_t)P * 3), sw((size_t)P);
  {
    const int off = mat_gl_index[n_theta][0];
    const double pi = 3.14159265358979323846, dphi = 2.0 * pi / (double)n_phi;
    int q = 0;
    for (int i = 0; i < n_theta; ++i) {
      const double ct = mat_gl_x[off + i], st = std::sqrt(1.0 - ct * ct);
      for (int j = 0; j < n_phi; ++j, ++q) {
        const double phi = dphi * (double)j;
        sc[(size_t)3 * q] = st * std::cos(phi); sc[(size_t)3 * q + 1] = st * std::sin(phi); sc[(size_t)3 * q + 2] = ct;
        sw[(size_t)q] = mat_gl_w[off + i] * dphi / (4.0 * pi);
      }
    }
  }
  // ---- cluster lists
  std::vector<int> eptr(cl->elem_ptr, cl->elem_ptr + nc + 1), eidx(cl->elem_idx, cl->elem_idx + cl->elem_ptr[nc]), edof(eidx.size());
  for (size_t q = 0; q < eidx.size(); ++q) edof[q] = hdof[(size_t)eidx[q]];
  std::vector<double> cc(cl->center, cl->center + 3 * (size_t)nc);
  S->h_eptr = eptr; S->h_edof = edof;
  // ---- near blocks: (i, i) and (i, j > i in near_clusters[i])   (slfmm.rs:484-497)
  std::vector<int>& bsrc = S->h_bsrc; std::vector<int>& bfld = S->h_bfld; std::vector<long long>& boff = S->h_boff;
  long long tot = 0;
  for (int i = 0; i < nc; ++i) {
    const long long ni = eptr[(size_t)i + 1] - eptr[(size_t)i];
    bsrc.push_back(i); bfld.push_back(i); boff.push_back(tot); tot += ni * ni;
    for (int q = cl->near_ptr[i]; q < cl->near_ptr[i + 1]; ++q) {
      const int j = cl->near_idx[q];
      if (j > i) { bsrc.push_back(i); bfld.push_back(j); boff.push_back(tot); tot += ni * (long long)(eptr[(size_t)j + 1] - eptr[(size_t)j]); }
    }
  }
  boff.push_back(tot);
  S->nbval = tot;
  if (tot >= 2000000000LL) { set_error("near field of %lld entries", tot); return fail(MA_ERR_UNSUPPORTED); }
  std::vector<int2> pairs((size_t)std::max<long long>(tot, 1));
  std::vector<long long> dpos; std::vector<int> dpanel;
  for (size_t b = 0; b < bsrc.size(); ++b) {
    const int ci = bsrc[b], cj = bfld[b];
    const int ns = eptr[(size_t)ci + 1] - eptr[(size_t)ci], nf = eptr[(size_t)cj + 1] - eptr[(size_t)cj];
    for (int i = 0; i < ns; ++i)
      for (int j = 0; j < nf; ++j) {
        const int se = eidx[(size_t)eptr[(size_t)ci] + i], fe = eidx[(size_t)eptr[(size_t)cj] + j];
        const long long pos = boff[b] + (long long)i * nf + j;
        if (ci == cj && se == fe) { dpos.push_back(pos); dpanel.push_back(se); pairs[(size_t)pos] = make_int2(se, (se + 1) % np == se ? se : (se + 1) % np); }
        else pairs[(size_t)pos] = make_int2(se, fe);
      }
  }
  // ---- D entries (:663-721)
  std::vector<std::vector<std::pair<int, c64>>> by_field((size_t)nc), by_source((size_t)nc);
  {
    const int order = std::max(n_terms, 2);
    std::vector<double> hr, hi;
    for (int i = 0; i < nc; ++i)
      for (int q = cl->far_ptr[i]; q < cl->far_ptr[i + 1]; ++q) {
        const int j = cl->far_idx[q];
        const double dx = cc[(size_t)3 * i] - cc[(size_t)3 * j], dy = cc[(size_t)3 * i + 1] - cc[(size_t)3 * j + 1], dz = cc[(size_t)3 * i + 2] - cc[(size_t)3 * j + 2];
        const double r = std::sqrt(dx * dx + dy * dy + dz * dz);
        if (!(r > 0.0)) { set_error("far clusters %d and %d share their centre", i, j); return fail(MA_ERR_INVALID); }
        spherical_hankel_first_kind(order, S->k * r, 1.0, hr, hi);
        const c64 d{-hi[0] * S->k, hr[0] * S->k};                      // h_0 * (i k)
        by_field[(size_t)j].push_back({i, d}); by_source[(size_t)i].push_back({j, d});
      }
  }
  auto flatten = [&](const std::vector<std::vector<std::pair<int, c64>>>& L, std::vector<int>& ptr, std::vector<int>& oth, std::vector<c64>& val) {
    ptr.assign(1, 0);
    for (const auto& l : L) { for (const auto& e : l) { oth.push_back(e.first); val.push_back(e.second); } ptr.push_back((int)oth.size()); }
  };
  std::vector<int> fptr, foth, tptr, toth; std::vector<c64> fval, tval;
  flatten(by_field, fptr, foth, fval); flatten(by_source, tptr, toth, tval);
  // ---- per-cluster views of the blocks: as source (rows of the block) or as field of an off-diagonal block (its transpose)
  std::vector<std::vector<SlfmmEntry>> views((size_t)nc);
  for (size_t b = 0; b < bsrc.size(); ++b) {
    views[(size_t)bsrc[b]].push_back({boff[b], bfld[b], 0});
    if (bsrc[b] != bfld[b]) views[(size_t)bfld[b]].push_back({boff[b], bsrc[b], 1});
  }
  std::vector<int> cptr(1, 0); std::vector<SlfmmEntry> cent;
  for (const auto& v : views) { cent.insert(cent.end(), v.begin(), v.end()); cptr.push_back((int)cent.size()); }
  int rc = MA_OK;
#define UP(dst, src) if (!rc) rc = upload(&S->dst, src)
  UP(d_eptr, eptr); UP(d_eidx, eidx); UP(d_edof, edof); UP(d_cc, cc); UP(d_sc, sc); UP(d_sw, sw); UP(d_cptr, cptr); UP(d_cent, cent);
  UP(d_fptr, fptr); UP(d_foth, foth); UP(d_fval, fval); UP(d_tptr, tptr); UP(d_toth, toth); UP(d_tval, tval);
#undef UP
  if (rc) return fail(rc);
  hipError_t e = hipMalloc(&S->d_bval, sizeof(c64) * (size_t)std::max<long long>(tot, 1));
  if (e == hipSuccess) e = hipMalloc(&S->d_up, sizeof(c64) * (size_t)nc * (size_t)P);
  if (e == hipSuccess) e = hipMalloc(&S->d_tr, sizeof(c64) * (size_t)nc * (size_t)P);
  int2* d_pairs = nullptr; long long* d_dpos = nullptr; int* d_dpanel = nullptr; c64* d_self = nullptr;
  if (e == hipSuccess) e = hipMalloc(&d_pairs, sizeof(int2) * pairs.size());
  if (e == hipSuccess) e = hipMemcpy(d_pairs, pairs.data(), sizeof(int2) * pairs.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&d_self, sizeof(c64) * (size_t)np);
  auto drop = [&]() { if (d_pairs) (void)hipFree(d_pairs); if (d_dpos) (void)hipFree(d_dpos); if (d_dpanel) (void)hipFree(d_dpanel); if (d_self) (void)hipFree(d_self); };
  if (e != hipSuccess) { set_error("SLFMM workspace: %s", hipGetErrorString(e)); drop(); return fail(MA_ERR_NOMEM); }
  rc = upload(&d_dpos, dpos); if (!rc) rc = upload(&d_dpanel, dpanel);
  // coefficient of compute_near_block (:583-584): dg_dn gamma tau + d2g beta with beta = i h / k (types.rs:64-70), sign +1
  BemPhys ph;
  const double bim = physics->tau > 0.0 ? physics->harmonic_factor / physics->wave_number : 0.0;
  if (!rc) rc = ma_bem_make_phys(plan, physics, 0.0, bim, &ph);
  ph.sign = 1.0;
  if (!rc) rc = bem_launch_near_list_values(plan->geom, ph, d_pairs, tot, S->d_bval, nullptr);
  if (!rc) rc = bem_launch_self_list_values(plan->geom, ph, d_self, nullptr);
  if (!rc && !dpos.empty()) {
    hipLaunchKernelGGL(slfmm_fix_diag_kernel, dim3((unsigned)((dpos.size() + 255) / 256)), dim3(256), 0, nullptr, (int)dpos.size(), d_dpos, d_dpanel,
                       reinterpret_cast<const dc*>(d_self), physics->gamma, reinterpret_cast<dc*>(S->d_bval));
    if (hipGetLastError() != hipSuccess) { set_error("SLFMM diagonal kernel failed"); rc = MA_ERR_HIP; }
  }
  if (!rc && hipDeviceSynchronize() != hipSuccess) { set_error("SLFMM near-field kernels failed"); rc = MA_ERR_HIP; }
  drop();
  if (rc) return fail(rc);
  *out = S;
  return MA_OK;
}

int slfmm_apply(ma_slfmm* S, const c64* d_x, c64* d_y, int transpose, hipStream_t st) {
  MA_HIP(hipSetDevice(S->device));
  const BemGeom& g = S->plan->geom;
  MA_HIP(hipMemsetAsync(d_y, 0, sizeof(c64) * (size_t)S->n, st));        // dofs outside every cluster receive nothing
  const dc* x = reinterpret_cast<const dc*>(d_x); dc* y = reinterpret_cast<dc*>(d_y);
  hipLaunchKernelGGL(slfmm_near_kernel, dim3(S->nc), dim3(256), 0, st, S->d_eptr, S->d_edof, S->d_cptr, S->d_cent, reinterpret_cast<const dc*>(S->d_bval), x, y, transpose);
  MA_HIP(hipGetLastError());
  // far field: forward T (e^-), D grouped by field, S (e^+); transpose S^T (e^+), D grouped by source, T^T (e^-)
  const double s_up = transpose ? 1.0 : -1.0, s_dn = transpose ? -1.0 : 1.0;
  hipLaunchKernelGGL(slfmm_up_kernel, dim3(S->nc), dim3(256), 0, st, g, S->d_eptr, S->d_eidx, S->d_edof, S->d_cc, S->d_sc, S->d_sw, S->P, S->k, s_up, x,
                     reinterpret_cast<dc*>(S->d_up));
  MA_HIP(hipGetLastError());
  hipLaunchKernelGGL(slfmm_translate_kernel, dim3(S->nc), dim3(256), 0, st, transpose ? S->d_tptr : S->d_fptr, transpose ? S->d_toth : S->d_foth,
                     reinterpret_cast<const dc*>(transpose ? S->d_tval : S->d_fval), S->P, reinterpret_cast<const dc*>(S->d_up), reinterpret_cast<dc*>(S->d_tr));
  MA_HIP(hipGetLastError());
  hipLaunchKernelGGL(slfmm_down_kernel, dim3(S->nc), dim3(256), 0, st, g, S->d_eptr, S->d_eidx, S->d_edof, S->d_cc, S->d_sc, S->d_sw, S->P, S->k, s_dn,
                     reinterpret_cast<const dc*>(S->d_tr), y);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int slfmm_near_matrix(ma_slfmm* S, c64* d_A, hipStream_t st) {
  MA_HIP(hipSetDevice(S->device));
  MA_HIP(hipMemsetAsync(d_A, 0, sizeof(c64) * (size_t)S->n * (size_t)S->n, st));
  for (size_t b = 0; b < S->h_bsrc.size(); ++b) {
    const int ci = S->h_bsrc[b], cj = S->h_bfld[b];
    const int ns = S->h_eptr[(size_t)ci + 1] - S->h_eptr[(size_t)ci], nf = S->h_eptr[(size_t)cj + 1] - S->h_eptr[(size_t)cj];
    if (ns == 0 || nf == 0) continue;
    hipLaunchKernelGGL(slfmm_scatter_block_kernel, dim3((unsigned)(((long long)ns * nf + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const dc*>(S->d_bval + S->h_boff[b]), ns, nf,
                       S->d_edof + S->h_eptr[(size_t)ci], S->d_edof + S->h_eptr[(size_t)cj], S->n, ci != cj ? 1 : 0, reinterpret_cast<dc*>(d_A));
    MA_HIP(hipGetLastError());
  }
  return MA_OK;
}
