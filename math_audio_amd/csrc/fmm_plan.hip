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
#include <chrono>
#include "ma_device_math.hpp"
#include "ma_tables.h"
#include <vector>
#include <cmath>
#include <algorithm>
#include <new>

using namespace ma;

struct SlfmmEntry { long long boff; long long poff; int other; int tflag; };
// round 5: a leaf-sized near block as ONE record (the leaf kernel reads it with one scalar load instead of walking bsrc / bfld -> eptr -> boff ...)
struct NearBlockDesc { long long boff; long long broff; long long bcoff; int a0; int f0; int ns; int nf; int both; int pad; };   // a near block seen from one of its two clusters; poff: its partial sums for this side

struct ma_slfmm {
  int device = 0; ma_bem_plan* plan = nullptr;
  long long n = 0; int nc = 0, P = 0; double k = 0.0;
  int* d_eptr = nullptr; int* d_eidx = nullptr; int* d_edof = nullptr;
  double* d_cc = nullptr; double* d_sc = nullptr; double* d_sw = nullptr;
  c64* d_bval = nullptr; long long nbval = 0;
  int* d_cptr = nullptr; SlfmmEntry* d_cent = nullptr;
  int* d_fptr = nullptr; int* d_foth = nullptr; c64* d_fval = nullptr;      // far pairs grouped by field cluster (forward)
  int* d_tptr = nullptr; int* d_toth = nullptr; c64* d_tval = nullptr;      // grouped by source cluster (transpose)
  c64* d_fdense = nullptr; c64* d_tdense = nullptr;                         // the same two as dense nc x nc matrices, when the lists are nearly full
  int* d_bsrc = nullptr; int* d_bfld = nullptr; long long* d_boff = nullptr; int nblocks = 0;   // the near blocks one by one
  long long* d_broff = nullptr; long long* d_bcoff = nullptr; c64* d_part = nullptr; long long max_block = 0, max_width = 0, max_rows = 0;   // their partial sums (rows, columns)
  NearBlockDesc* d_bdesc = nullptr; c64* d_xp = nullptr; long long listed = 0;   // round 5 (leaf-sized blocks): one record per block, x in cluster order
  c64* d_up = nullptr; c64* d_tr = nullptr;
  c64* d_phase = nullptr;          // w_p e^{i k s_p.(x_j - C_c)} per listed element and sphere point, when stored (MA_FMM_STORE_PHASES=1; 0: recomputed with libm)
  bool fast_phases = false;        // round 4 (the default, MA_FMM_STORE_PHASES=2): recomputed with the bounded-argument sin / cos, no table
  bool overlap = false;            // an element may sit in several clusters (mlfmm.rs' octant rule): rows are summed with atomics
  bool covers_all = false;         // every dof is listed exactly once and the near field runs in two passes: its second pass WRITES all of y, the apply need not clear it
  // round 4: the near blocks' first pass runs on a second stream beside the far chain (up -> translate -> down), whose result goes to
  // d_yfar and is added by the near field's second pass -- the apply costs max(near, far) + one short pass instead of their sum
  // (MA_FMM_OVERLAP=0: one stream, the round-2 order). Created on first use.
  hipStream_t st2 = nullptr; hipEvent_t ev_fork = nullptr, ev_near = nullptr; c64* d_yfar = nullptr; int overlap_streams = -1;
  // round 5 (multi-level operator): the leaf level's translation on a third stream beside the upward chain of the levels above it
  hipStream_t st3 = nullptr; hipEvent_t ev_up = nullptr, ev_leaf = nullptr;
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

// near blocks, y_c = sum over the cluster's entries of B x (entry stored with this cluster as its rows) or B^T x (stored with this
// cluster as its columns). The two kinds are read the way they lie in memory: a block used as B^T has the cluster's rows as its
// COLUMNS, so lane = row streams it contiguously (phase A, the partner's entries one after the other, x_j broadcast); a block used as
// B has them as its rows, so a group of G lanes walks a row's entries and reduces at the end (phase B). With lanes over j in both
// cases the transposed half was read with a stride of a whole row (16 x the bytes). WPC wavefronts per cluster: 4 (one workgroup per
// cluster) for large clusters, 1 (four clusters per workgroup) for the leaves of a multi-level tree. Sums are taken in a fixed order.
template <int WPC>
__global__ __launch_bounds__(256) void slfmm_near_kernel(const int* __restrict__ eptr, const int* __restrict__ edof, const int* __restrict__ cptr,
                                                         const SlfmmEntry* __restrict__ cent, const dc* __restrict__ bval, const dc* __restrict__ x,
                                                         dc* __restrict__ y, int tmode, int overlap, int nclusters, int G) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * (4 / WPC) + wave / WPC, w = wave % WPC;
  const bool live = c < nclusters;
  const int e0 = live ? eptr[c] : 0, nc_ = live ? eptr[c + 1] - e0 : 0;
  const int q0 = live ? cptr[c] : 0, q1 = live ? cptr[c + 1] : 0;
  constexpr int U = 8;                                     // independent loads in flight per lane: the loops are latency-bound without them
  // phase A: the entries this cluster reads as B^T. G consecutive lanes take G consecutive rows (contiguous in the stored block), the
  // 64 / G lane sets split the partner's elements between them and are summed at the end.
  const int JG = 64 / G, jg = lane / G, lg = lane % G;
  for (int ic = w; ic * G < nc_; ic += WPC) {
    const int i = ic * G + lg;
    const bool vi = i < nc_;
    const int ii = vi ? i : 0;
    double sr = 0.0, si = 0.0;
    for (int q = q0; q < q1; ++q) {
      const SlfmmEntry en = cent[q];
      const bool self = en.other == c;
      const bool tr = (en.tflag != 0) != (self && tmode != 0);
      if (!tr) continue;
      const int o0 = eptr[en.other], no = eptr[en.other + 1] - o0;
      const dc* B = bval + en.boff + ii;
      const int* od = edof + o0;
      for (int j0 = jg; j0 < no; j0 += JG * U) {
        dc bb[U], xx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u * JG;
          const int jj = j < no ? j : j0;
          bb[u] = B[(long long)jj * nc_];
          xx[u] = x[od[jj]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (j0 + u * JG < no) { sr += bb[u].re * xx[u].re - bb[u].im * xx[u].im; si += bb[u].re * xx[u].im + bb[u].im * xx[u].re; }
        }
      }
    }
    for (int off = G; off < 64; off <<= 1) { sr += __shfl_xor(sr, off, 64); si += __shfl_xor(si, off, 64); }
    if (vi && jg == 0) {
      dc* o = y + edof[e0 + i];
      if (overlap) { atomicAdd(&o->re, sr); atomicAdd(&o->im, si); }       // the dof is a row of several clusters
      else *o = dc_make(sr, si);
    }
  }
  __syncthreads();
  // phase B: the entries read as B. G lanes walk a row; a lane set carries U rows at once (one gather of x serves all of them).
  for (int t = w; t * JG * U < nc_; t += WPC) {
    const int ibase = t * JG * U + jg;                     // rows ibase + r JG
    double ar[U], ai[U];
#pragma unroll
    for (int r = 0; r < U; ++r) { ar[r] = 0.0; ai[r] = 0.0; }
    for (int q = q0; q < q1; ++q) {
      const SlfmmEntry en = cent[q];
      const bool self = en.other == c;
      const bool tr = (en.tflag != 0) != (self && tmode != 0);
      if (tr) continue;
      const int o0 = eptr[en.other], no = eptr[en.other + 1] - o0;
      const dc* B = bval + en.boff;
      for (int j = lg; j < no; j += G) {
        const dc xv = x[edof[o0 + j]];
        dc bb[U];
#pragma unroll
        for (int r = 0; r < U; ++r) {
          const int i = ibase + r * JG;
          bb[r] = B[(long long)(i < nc_ ? i : 0) * no + j];
        }
#pragma unroll
        for (int r = 0; r < U; ++r) {
          if (ibase + r * JG < nc_) { ar[r] += bb[r].re * xv.re - bb[r].im * xv.im; ai[r] += bb[r].re * xv.im + bb[r].im * xv.re; }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < U; ++r) {
      double sr = ar[r], si = ai[r];
      for (int off = G >> 1; off > 0; off >>= 1) { sr += __shfl_xor(sr, off, 64); si += __shfl_xor(si, off, 64); }
      const int i = ibase + r * JG;
      if (i < nc_ && lg == 0) {
        dc* o = y + edof[e0 + i];
        if (overlap) { atomicAdd(&o->re, sr); atomicAdd(&o->im, si); }
        else { o->re += sr; o->im += si; }
      }
    }
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
  // One workgroup per cluster. The far list is walked by 256 / P thread sets at once (P sphere points each), four pairs in flight per
  // thread: with one thread per point and one pair at a time the loop is a chain of dependent gathers and the kernel latency-bound.
  // The sets' partial sums are added in set order, so the result does not depend on timing.
  __shared__ dc part[256];
  constexpr int U = 4;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int q0 = fptr[c], q1 = fptr[c + 1];
  if (P <= 128) {
    const int QG = 256 / P, qg = tid / P, p = tid % P;
    double sr = 0.0, si = 0.0;
    if (qg < QG)
      for (int q = q0 + qg; q < q1; q += QG * U) {
        dc d[U], m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int qq = q + u * QG;
          const int qc = qq < q1 ? qq : q;
          d[u] = fval[qc]; m[u] = up[(long long)foth[qc] * P + p];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (q + u * QG < q1) { sr += d[u].re * m[u].re - d[u].im * m[u].im; si += d[u].re * m[u].im + d[u].im * m[u].re; }
      }
    part[tid] = dc_make(sr, si);
    __syncthreads();
    if (tid < P) {
      double tr_ = 0.0, ti_ = 0.0;
      for (int g = 0; g < QG; ++g) { tr_ += part[g * P + tid].re; ti_ += part[g * P + tid].im; }
      tr[(long long)c * P + tid] = dc_make(tr_, ti_);
    }
    return;
  }
  for (int p = tid; p < P; p += 256) {
    double sr = 0.0, si = 0.0;
    for (int q = q0; q < q1; q += U) {
      dc d[U], m[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int qc = q + u < q1 ? q + u : q;
        d[u] = fval[qc]; m[u] = up[(long long)foth[qc] * P + p];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (q + u < q1) { sr += d[u].re * m[u].re - d[u].im * m[u].im; si += d[u].re * m[u].im + d[u].im * m[u].re; }
    }
    tr[(long long)c * P + p] = dc_make(sr, si);
  }
}
// y[dof_j] += sum_p w_p exp(i sgn k s_p . (x_j - C_c)) tr[c][p]     (sgn +1: S l; -1: T^T m)
__global__ __launch_bounds__(256) void slfmm_down_kernel(BemGeom g, const int* __restrict__ eptr, const int* __restrict__ eidx, const int* __restrict__ edof,
                                                         const double* __restrict__ cc, const double* __restrict__ sc, const double* __restrict__ sw,
                                                         int P, double k, double sgn, const dc* __restrict__ tr, dc* __restrict__ y, int overlap) {
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
    if (overlap) { atomicAdd(&o->re, sr); atomicAdd(&o->im, si); }
    else { o->re += sr; o->im += si; }
  }
}
// the self blocks' diagonal: singular integral (+ SLFMM's free term). self_vals = coefficient - gamma/2 (TBEM's free term): + gamma
// with slfmm.rs' free term, + gamma / 2 without (mlfmm.rs)
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

// The far lists of this method hold every cluster that is not near, so D is a nearly full nc x nc matrix and tr = D up a dense
// complex product [nc x nc] [nc x P]. The list kernel above re-reads a row of `up` from L2 for every pair (6 GB per leaf level of the
// 50k box: L2-bound, 0.5 ms). Here it runs on v_mfma_f64_16x16x4_f64, three real products per complex one (four until round 5). D is stored by SOURCE
// (DT[s][c], the receiving cluster c contiguous), so the A operand of a step (16 receiving clusters x 4 sources) is four 256-byte
// runs and the B operands (4 sources x 16 points) rows of `up`. A wavefront owns 16 receiving clusters and up to 128 points; the KS
// wavefronts of a workgroup split the sources and add their parts in wavefront order through LDS, so the result does not depend on
// timing. Absent pairs are zeros of D. (Vector-FMA forms were tried first: `up` through scalar loads stalls on the scalar cache,
// through LDS broadcasts is LDS-bound, through v_readlane reaches 0.31 ms.)
typedef double fmm_v4d __attribute__((ext_vector_type(4)));
template <int KS, int NT>                                   // NT point tiles of 16 per workgroup
__global__ __launch_bounds__(64 * KS) void fmm_translate_dense_kernel(const dc* __restrict__ DT, int nc, int P, const dc* __restrict__ up, dc* __restrict__ tr) {
  __shared__ dc part[16 * 16 * NT];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int c0 = blockIdx.x * 16;
  const int pb = blockIdx.y * (16 * NT);
  const int npts = P - pb < 16 * NT ? P - pb : 16 * NT;
  const int nt = (npts + 15) / 16;
  const int steps = (nc + 3) / 4, per = (steps + KS - 1) / KS;
  const int kb = w * per, ke = kb + per < steps ? kb + per : steps;
  const dc* acol = DT + (c0 + lr < nc ? c0 + lr : nc - 1);
  fmm_v4d cr[NT], ci[NT], cs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { cr[t] = (fmm_v4d){0, 0, 0, 0}; ci[t] = cr[t]; cs[t] = cr[t]; }
  dc a, bv[NT], an, bn[NT];
  auto fetch = [&](int ks, dc& fa, dc* fb) {
    const int sidx = ks * 4 + lk;
    const int sc = sidx < nc ? sidx : nc - 1;
    fa = acol[(long long)sc * nc];
    if (sidx >= nc) fa = dc_make(0.0, 0.0);                 // sources past the end multiply by zero
    const dc* urow = up + (long long)sc * P + pb;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < nt) { const int col = 16 * t + lr; fb[t] = urow[col < npts ? col : npts - 1]; }
  };
  if (kb < ke) fetch(kb, a, bv);
  for (int ks = kb; ks < ke; ++ks) {
    if (ks + 1 < ke) fetch(ks + 1, an, bn);                 // the next step's loads are in flight during this step's products
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < nt) {
        // three real products per complex one (round 5; the LU's trailing update does the same): t1 = Re Re, t2 = Im Im, t3 = (Re + Im)(Re + Im);
        // re = t1 - t2, im = t3 - t1 - t2 at the end -- normwise, not componentwise, accurate
        cr[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, bv[t].re, cr[t], 0, 0, 0);
        ci[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, bv[t].im, ci[t], 0, 0, 0);
        cs[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re + a.im, bv[t].re + bv[t].im, cs[t], 0, 0, 0);
      }
    a = an;
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = bn[t];
  }
  // C: col = lane & 15, row = (lane >> 4) + 4 reg
  for (int q = 0; q < KS; ++q) {                            // parts are added in wavefront order
    if (w == q) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            dc* o = part + (lk + 4 * g) * (16 * NT) + 16 * t + lr;
            const double pre = cr[t][g] - ci[t][g], pim = cs[t][g] - cr[t][g] - ci[t][g];
            if (q == 0) *o = dc_make(pre, pim);
            else { o->re += pre; o->im += pim; }
          }
        }
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < 16 * npts; idx += 64 * KS) {
    const int r = idx / npts, j = idx % npts;
    if (c0 + r < nc) tr[(long long)(c0 + r) * P + pb + j] = part[r * (16 * NT) + j];
  }
}

// tr = D up over one level: the dense form when it was built, the pair lists otherwise
static int fmm_launch_translate(const int* fptr, const int* foth, const c64* fval, const c64* dense, int nc, int P, const c64* up, c64* tr, hipStream_t st) {
  if (dense) {
    constexpr int KS = 8;
    const int rows = (nc + 15) / 16;
    const dc* Dd = reinterpret_cast<const dc*>(dense); const dc* u = reinterpret_cast<const dc*>(up); dc* t = reinterpret_cast<dc*>(tr);
    // few row tiles: the points are spread over more workgroups (D is then read once per 32 points, from L2 / Infinity Cache)
    if (rows * ((P + 127) / 128) < 1024) hipLaunchKernelGGL((fmm_translate_dense_kernel<KS, 2>), dim3((unsigned)rows, (unsigned)((P + 31) / 32)), dim3(64 * KS), 0, st, Dd, nc, P, u, t);
    else hipLaunchKernelGGL((fmm_translate_dense_kernel<KS, 8>), dim3((unsigned)rows, (unsigned)((P + 127) / 128)), dim3(64 * KS), 0, st, Dd, nc, P, u, t);
  } else {
    hipLaunchKernelGGL(slfmm_translate_kernel, dim3((unsigned)nc), dim3(256), 0, st, fptr, foth, reinterpret_cast<const dc*>(fval), P, reinterpret_cast<const dc*>(up),
                       reinterpret_cast<dc*>(tr));
  }
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// The translations of SEVERAL levels of a tree in one launch (round 4): they are independent of one another once every level's
// multipoles stand, and each alone fills a fraction of the chip (a top level has a handful of clusters); the same workgroup body as
// fmm_translate_dense_kernel<KS, 2>, the level picked from the block index.
struct FmmLevels { int nl; int first[9]; int rows[8]; int nc[8]; int P[8]; const dc* DT[8]; const dc* up[8]; dc* tr[8]; };
template <int KS, int NT>
__global__ __launch_bounds__(64 * KS) void fmm_translate_levels_kernel(FmmLevels V) {
  __shared__ dc part[16 * 16 * NT];
  int l = 0;
  while (l + 1 < V.nl && (int)blockIdx.x >= V.first[l + 1]) ++l;
  const int rem = (int)blockIdx.x - V.first[l];
  const int nc = V.nc[l], P = V.P[l];
  const dc* __restrict__ DT = V.DT[l]; const dc* __restrict__ up = V.up[l]; dc* __restrict__ tr = V.tr[l];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int c0 = (rem % V.rows[l]) * 16;
  const int pb = (rem / V.rows[l]) * (16 * NT);
  const int npts = P - pb < 16 * NT ? P - pb : 16 * NT;
  const int nt = (npts + 15) / 16;
  const int steps = (nc + 3) / 4, per = (steps + KS - 1) / KS;
  const int kb = w * per, ke = kb + per < steps ? kb + per : steps;
  const dc* acol = DT + (c0 + lr < nc ? c0 + lr : nc - 1);
  fmm_v4d cr[NT], ci[NT], cs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { cr[t] = (fmm_v4d){0, 0, 0, 0}; ci[t] = cr[t]; cs[t] = cr[t]; }
  dc a, bv[NT], an, bn[NT];
  auto fetch = [&](int ks, dc& fa, dc* fb) {
    const int sidx = ks * 4 + lk;
    const int sc = sidx < nc ? sidx : nc - 1;
    fa = acol[(long long)sc * nc];
    if (sidx >= nc) fa = dc_make(0.0, 0.0);
    const dc* urow = up + (long long)sc * P + pb;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < nt) { const int col = 16 * t + lr; fb[t] = urow[col < npts ? col : npts - 1]; }
  };
  if (kb < ke) fetch(kb, a, bv);
  for (int ks = kb; ks < ke; ++ks) {
    if (ks + 1 < ke) fetch(ks + 1, an, bn);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < nt) {
        // three real products per complex one (round 5; the LU's trailing update does the same): t1 = Re Re, t2 = Im Im, t3 = (Re + Im)(Re + Im);
        // re = t1 - t2, im = t3 - t1 - t2 at the end -- normwise, not componentwise, accurate
        cr[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, bv[t].re, cr[t], 0, 0, 0);
        ci[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, bv[t].im, ci[t], 0, 0, 0);
        cs[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re + a.im, bv[t].re + bv[t].im, cs[t], 0, 0, 0);
      }
    a = an;
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = bn[t];
  }
  for (int q = 0; q < KS; ++q) {                            // parts are added in wavefront order, as in the one-level kernel: the same bits
    if (w == q) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            dc* o = part + (lk + 4 * g) * (16 * NT) + 16 * t + lr;
            const double pre = cr[t][g] - ci[t][g], pim = cs[t][g] - cr[t][g] - ci[t][g];
            if (q == 0) *o = dc_make(pre, pim);
            else { o->re += pre; o->im += pim; }
          }
        }
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < 16 * npts; idx += 64 * KS) {
    const int r = idx / npts, j = idx % npts;
    if (c0 + r < nc) tr[(long long)(c0 + r) * P + pb + j] = part[r * (16 * NT) + j];
  }
}
// would fmm_launch_translate take the <KS, 2> dense kernel for this level? (those are the levels the batched launch can carry)
static bool fmm_translate_batchable(const c64* dense, int nc, int P) { return dense && ((nc + 15) / 16) * ((P + 127) / 128) < 1024; }
// points per workgroup of the batched launch: 16 NT with NT = 2 / 5 / 8 by the sphere rule of the level that carries the most work (the
// last one listed: the leaves): P = 72 (6 x 12 points) fits NT = 5, so D is read once per level instead of three times
static int fmm_levels_nt(int P) {
  (void)P;
  return 2;                                                  // measured: 32 points per workgroup (221 us on the 50k tree) beat 80 (249 us): more workgroups, the D rows come from L2
}
static int fmm_launch_translate_levels(const FmmLevels& V, int nt, hipStream_t st) {
  if (V.nl <= 0) return MA_OK;
  (void)nt;                                                  // 32 points per workgroup: the only form kept (80 and 128 were slower: fmm_levels_nt)
  hipLaunchKernelGGL((fmm_translate_levels_kernel<8, 2>), dim3((unsigned)V.first[V.nl]), dim3(64 * 8), 0, st, V);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// dense nc x nc form (stored by source) of pair lists grouped by receiving cluster; nullptr (lists stay in use) when fewer than a quarter of the pairs exist or when
// MA_FMM_DENSE_TRANSLATE=0 asks for the lists
static int fmm_dense_from_lists(const std::vector<int>& ptr, const std::vector<int>& oth, const std::vector<c64>& val, int nc, c64** d_out) {
  *d_out = nullptr;
  const char* e = getenv("MA_FMM_DENSE_TRANSLATE");
  if (e && atoi(e) == 0) return MA_OK;
  if (nc < 8 || (double)oth.size() < 0.25 * (double)nc * (double)nc || (double)nc * (double)nc * 16.0 > 16e9) return MA_OK;
  std::vector<c64> dense((size_t)nc * (size_t)nc, c64{0.0, 0.0});
  for (int c = 0; c < nc; ++c)
    for (int q = ptr[(size_t)c]; q < ptr[(size_t)c + 1]; ++q) {
      c64& o = dense[(size_t)oth[(size_t)q] * (size_t)nc + (size_t)c];          // by source: the receiving cluster is contiguous
      o.re += val[(size_t)q].re; o.im += val[(size_t)q].im;
    }
  return upload(d_out, dense);
}

// sum over an aligned set of G lanes (8 <= G <= 64, a power of two), every lane receives it: DPP inside a row of 16 lanes
// (no LDS crossbar), ds_bpermute only for the two widest steps
template <int CTRL>
__device__ __forceinline__ double fmm_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fmm_set_sum(double v, int G) {
  v += fmm_dpp<0xB1>(v);                                     // quad_perm [1,0,3,2]
  v += fmm_dpp<0x4E>(v);                                     // quad_perm [2,3,0,1]
  v += fmm_dpp<0x141>(v);                                    // row_half_mirror: the other quad of eight
  if (G >= 16) v += fmm_dpp<0x140>(v);                       // row_mirror: the other eight of sixteen
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
// near blocks in two passes. Pass 1, one wavefront (small blocks) or one workgroup (large ones) per block: the block is read ONCE
// and both products it takes part in are formed, B x[cols] for the cluster that holds its rows and, off the diagonal, B^T x[rows] for
// the other; they go to the block's two slots of `part`. Pass 2, per cluster: the slots of its entries are added in entry order.
// The per-cluster kernel above reads every off-diagonal block twice, once from each side; this reads it once and the sums are
// still taken in a fixed order. G lanes walk a row (G >= the block's width for a leaf), the 64 / G lane sets take eight rows each
// per pass so that eight loads are in flight.
template <int WPB>
__global__ __launch_bounds__(256) void slfmm_near_blocks_kernel(const int* __restrict__ eptr, const int* __restrict__ edof, const int* __restrict__ bsrc,
                                                                const int* __restrict__ bfld, const long long* __restrict__ boff, const long long* __restrict__ broff,
                                                                const long long* __restrict__ bcoff, int nblocks, const dc* __restrict__ bval,
                                                                const dc* __restrict__ x, dc* __restrict__ part, int tmode) {
  __shared__ dc cpart[WPB == 1 ? 1 : 4 * 64];
  // (round 4, measured and removed: the block's row sums parked in LDS and stored coalesced once per block instead of one 16-byte store
  // per row -- the near pass of the 50k tree went from 400 to 496 us beside the far chain, 0.839 -> 0.851 ms per apply)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int w = WPB == 1 ? 0 : wave;
  // grid-stride over the blocks (round 4): the launch may be capped at a few workgroups per CU, so that the kernels of the far
  // chain on the other stream find free slots beside it (uncapped, its tens of thousands of workgroups kept them waiting)
  for (int b = WPB == 1 ? blockIdx.x * 4 + wave : blockIdx.x; b < nblocks; b += WPB == 1 ? gridDim.x * 4 : gridDim.x) {   // WPB == 4: uniform over the workgroup
  const int a = bsrc[b], f = bfld[b];
  const int a0 = eptr[a], ns = eptr[a + 1] - a0, f0 = eptr[f], nf = eptr[f + 1] - f0;
  const dc* B = bval + boff[b];
  const bool both = a != f, self_t = !both && tmode != 0;   // the transpose of a diagonal block: its column sums take the row slot
  dc* prow = part + broff[b];
  dc* pcol = both ? part + bcoff[b] : prow;
  int G = 8; while (G < 64 && G < nf) G <<= 1;
  const int JG = 64 / G, jg = lane / G, lg = lane % G;
  constexpr int U = WPB == 1 ? 2 : 8;                        // small blocks: occupancy pays more than loads in flight (measured 1, 2, 4, 8)
  for (int j0 = 0; j0 < nf; j0 += G) {
    const int j = j0 + lg;
    const bool vj = j < nf;
    const int jc = vj ? j : j0;
    dc xf = x[edof[f0 + jc]];
    if (!vj) xf = dc_make(0.0, 0.0);
    double cr = 0.0, ci = 0.0;
    for (int i0 = w * JG + jg; i0 < ns; i0 += WPB * JG * U) {
      dc bb[U], xa[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * WPB * JG;
        const int ii = i < ns ? i : i0;
        bb[u] = B[(long long)ii * nf + jc];
        xa[u] = x[edof[a0 + ii]];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * WPB * JG;
        const bool ok = i < ns;                              // the same for the whole lane set
        double pr = (ok && vj) ? bb[u].re * xf.re - bb[u].im * xf.im : 0.0;
        double pi = (ok && vj) ? bb[u].re * xf.im + bb[u].im * xf.re : 0.0;
        pr = fmm_set_sum(pr, G); pi = fmm_set_sum(pi, G);
        if (ok && lg == 0 && !self_t) {
          if (j0 == 0) prow[i] = dc_make(pr, pi);
          else { prow[i].re += pr; prow[i].im += pi; }       // wider than 64: the same lane comes back to its row
        }
        if (ok && vj) { cr += bb[u].re * xa[u].re - bb[u].im * xa[u].im; ci += bb[u].re * xa[u].im + bb[u].im * xa[u].re; }
      }
    }
    if (both || self_t) {                                    // uniform over the block
      for (int off = G; off < 64; off <<= 1) { cr += __shfl_xor(cr, off, 64); ci += __shfl_xor(ci, off, 64); }
      if (WPB == 1) {
        if (jg == 0 && vj) pcol[j] = dc_make(cr, ci);
      } else {
        if (jg == 0) cpart[w * 64 + lg] = dc_make(cr, ci);
        __syncthreads();
        if (w == 0 && jg == 0 && vj) {
          double tr_ = 0.0, ti_ = 0.0;
          for (int q = 0; q < 4; ++q) { tr_ += cpart[q * 64 + lg].re; ti_ += cpart[q * 64 + lg].im; }
          pcol[j] = dc_make(tr_, ti_);
        }
        __syncthreads();
      }
    }
  }
  }
}
// One butterfly step of a transposing reduction: two vectors A, B in, one out whose lanes with `bit` clear hold A[l] + A[partner] and
// whose lanes with `bit` set hold B[l] + B[partner] (partner by the DPP control CTRL). Three such steps take eight vectors -- the real
// and imaginary partial sums of four rows -- to ONE register (7 folds of 7 instructions) where summing each alone takes 8 x 3 steps of 3.
template <int CTRL>
__device__ __forceinline__ double fmm_fold(double a, double b, bool bit) {
  const double keep = bit ? b : a, send = bit ? a : b;
  return keep + fmm_dpp<CTRL>(send);
}
// Leaf-sized blocks (at most 64 rows and 64 columns: the multi-level operator's near field), round 5. The kernel above spends a block's
// time on DEPENDENT loads: bsrc / bfld -> eptr -> the block, and inside the row loop edof -> x for every row -- four memory round trips
// before the first product and two more per pass, which eight wavefronts per SIMD only partly hide (382 us for the 50k tree's 164 000
// blocks). Here a block is ONE record (a scalar load, the next block's fetched while this one is worked on), x comes in cluster order
// (one coalesced load for the rows, staged in LDS, one for the columns), and eight rows per lane set are loaded at once, unconditionally
// (clamped into the block; what lies beyond meets a zero of x) and pinned ahead of the arithmetic: record -> loads -> products.
__global__ __launch_bounds__(256) void fmm_cluster_order_kernel(const int* __restrict__ edof, long long listed, const dc* __restrict__ x, dc* __restrict__ xp) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e < listed) xp[e] = x[edof[e]];
}
__global__ __launch_bounds__(256) void slfmm_near_leaf_blocks_kernel(const NearBlockDesc* __restrict__ desc, int nblocks, const dc* __restrict__ bval,
                                                                     const dc* __restrict__ xp, dc* __restrict__ part, int tmode) {
  __shared__ dc xs_all[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  dc* xs = xs_all[wave];
  constexpr int U = 4;
  const int nw = gridDim.x * 4;
  int b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
  if (b >= nblocks) return;
  NearBlockDesc d = desc[b];
  for (;;) {
    const int bn = b + nw;
    const NearBlockDesc dn = desc[bn < nblocks ? bn : b];   // the next block's record travels while this block is worked on
    const int ns = d.ns, nf = d.nf;
    const dc* B = bval + d.boff;
    const bool both = d.both != 0, self_t = !both && tmode != 0;   // the transpose of a diagonal block: its column sums take the row slot
    dc* prow = part + d.broff;
    dc* pcol = both ? part + d.bcoff : prow;
    int G = 8; while (G < 64 && G < nf) G <<= 1;
    const int JG = 64 / G, jg = lane / G, lg = lane % G;
    const dc xr = xp[d.a0 + (lane < ns ? lane : ns - 1)];
    const dc xf0 = xp[d.f0 + (lg < nf ? lg : nf - 1)];
    const int jc = lg < nf ? lg : nf - 1;
    double cr = 0.0, ci = 0.0;
    // lane bits 0 and 1 taken relative to bit 2, so that the cheap row_half_mirror pairs lanes that hold the same vector
    const int b2 = (lane >> 2) & 1;
    const bool bit0 = ((lane ^ b2) & 1) != 0, bit1 = (((lane >> 1) ^ b2) & 1) != 0, bit2 = b2 != 0;
    for (int i0 = 0; i0 < ns; i0 += JG * U) {
      dc bb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = i0 + jg * U + u; bb[u] = B[(i < ns ? i : ns - 1) * nf + jc]; }   // a lane set takes four consecutive rows
      __builtin_amdgcn_sched_barrier(0);
      if (i0 == 0) xs[lane] = lane < ns ? xr : dc_make(0.0, 0.0);   // one wavefront: its LDS operations stay in order
      const dc xf = lg < nf ? xf0 : dc_make(0.0, 0.0);
      double pr[U], pi[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + jg * U + u;
        const bool ok = i < ns;                              // the same for the whole lane set
        const dc xa = xs[ok ? i : 0];
        pr[u] = bb[u].re * xf.re - bb[u].im * xf.im; pi[u] = bb[u].re * xf.im + bb[u].im * xf.re;    // xf = 0 past the block's columns
        if (ok) { cr += bb[u].re * xa.re - bb[u].im * xa.im; ci += bb[u].re * xa.im + bb[u].im * xa.re; }
      }
      // the four rows' sums: eight vectors folded to one register on lane bits 0, 1, 2, then summed over the set's remaining lane bits;
      // lanes 0..7 of the set then hold (re, im) of its four rows
      const double r0 = fmm_fold<0xB1>(pr[0], pi[0], bit0), r1 = fmm_fold<0xB1>(pr[1], pi[1], bit0);
      const double r2 = fmm_fold<0xB1>(pr[2], pi[2], bit0), r3 = fmm_fold<0xB1>(pr[3], pi[3], bit0);
      const double q0 = fmm_fold<0x4E>(r0, r1, bit1), q1 = fmm_fold<0x4E>(r2, r3, bit1);
      double z = fmm_fold<0x141>(q0, q1, bit2);
      if (G >= 16) z += fmm_dpp<0x128>(z);                   // row_ror:8 = the lane 8 further inside the row of 16
      if (G >= 32) z += __shfl_xor(z, 16, 64);
      if (G >= 64) z += __shfl_xor(z, 32, 64);
      const int irow = i0 + jg * U + (bit1 ? 1 : 0) + (bit2 ? 2 : 0);
      if (lg < 8 && irow < ns && !self_t) reinterpret_cast<double*>(prow + irow)[bit0 ? 1 : 0] = z;
    }
    if (both || self_t) {                                    // uniform over the block
      for (int off = G; off < 64; off <<= 1) { cr += __shfl_xor(cr, off, 64); ci += __shfl_xor(ci, off, 64); }
      if (jg == 0 && lg < nf) pcol[lg] = dc_make(cr, ci);
    }
    if (bn >= nblocks) break;
    b = bn; d = dn;
  }
}
// Large blocks (wider than one lane set): the same two products with the rows outside and the column chunks inside. A lane keeps
// x[cols] of its NCH <= 8 chunks and their column sums in registers, a row is reduced over the lanes ONCE (not once per chunk) and
// written once. One workgroup per block, the four wavefronts take rows w, w + 4, ...; their column sums meet in LDS in wavefront order.
constexpr int FMM_NCH = 8;
constexpr int FMM_WIDE_ROWS = 512;                            // rows of a block whose x entries are staged in LDS (taller blocks read x per row)
template <int NCH>   // column chunks of 64 a lane may hold: 2 / 4 (next rows prefetched) / 8
__device__ __forceinline__ void near_wide_block(int b, dc* __restrict__ cpart, dc* __restrict__ xrow, const int* __restrict__ eptr, const int* __restrict__ edof,
                                                const int* __restrict__ bsrc, const int* __restrict__ bfld, const long long* __restrict__ boff,
                                                const long long* __restrict__ broff, const long long* __restrict__ bcoff,
                                                const dc* __restrict__ bval, const dc* __restrict__ x, dc* __restrict__ part, int tmode) {
  constexpr bool PIPE = NCH <= 2;                            // (the prefetching form of four chunks takes 240 registers: one wavefront per SIMD)
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
  const int a = bsrc[b], f = bfld[b];
  const int a0 = eptr[a], ns = eptr[a + 1] - a0, f0 = eptr[f], nf = eptr[f + 1] - f0;
  const dc* B = bval + boff[b];
  const bool both = a != f, self_t = !both && tmode != 0;
  dc* prow = part + broff[b];
  dc* pcol = both ? part + bcoff[b] : prow;
  const int nch = (nf + 63) >> 6;                            // <= NCH (the launcher's choice)
  for (int i = threadIdx.x; i < ns; i += 256) xrow[i] = x[edof[a0 + i]];            // ns <= FMM_WIDE_ROWS: the launcher's condition
  dc xf[NCH]; double cr[NCH], ci[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int j = ch * 64 + lane;
    xf[ch] = (ch < nch && j < nf) ? x[edof[f0 + j]] : dc_make(0.0, 0.0);
    cr[ch] = 0.0; ci[ch] = 0.0;
  }
  __syncthreads();
  constexpr int U = 4;
  // the next rows' entries are fetched while this iteration's products run (two sets of U x nch loads in flight per lane)
  dc bb[PIPE ? NCH : 1][U], bn[PIPE ? NCH : 1][U];
  // every load UNCONDITIONAL (rows clamped into the block, columns of absent chunks to column 0): under a branch the compiler cannot count
  // the loads in flight and waits for all of them
  auto fetch = [&](int i0, dc (*dst)[U]) {
#pragma unroll
    for (int ch = 0; ch < (PIPE ? NCH : 1); ++ch) {
      const int j = ch * 64 + lane;
      const int jc = j < nf ? j : 0;
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = i0 + 4 * u; dst[ch][u] = B[(long long)(i < ns ? i : ns - 1) * nf + jc]; }
    }
  };
  if (PIPE) fetch(w, bb);
  for (int i0 = w; i0 < ns; i0 += 4 * U) {
    const bool more = PIPE && i0 + 4 * U < ns;
    if constexpr (PIPE) { fetch(i0 + 4 * U, bn); __builtin_amdgcn_sched_barrier(0); }   // the next rows' loads go out first and stay there (the scheduler would sink them to their use)
    dc xa[U]; double pr[U], pi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = i0 + 4 * u; xa[u] = i < ns ? xrow[i] : dc_make(0.0, 0.0); pr[u] = 0.0; pi[u] = 0.0; }   // rows past the block meet a zero (no mask on the products)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      if (ch < nch) {                                        // uniform
        const int j = ch * 64 + lane;
        const bool vj = j < nf;
        if constexpr (!PIPE) {                               // many chunks: this chunk's U loads only (the whole row set would take every register)
#pragma unroll
          for (int u = 0; u < U; ++u) { const int i = i0 + 4 * u; bb[0][u] = B[(long long)(i < ns ? i : ns - 1) * nf + (vj ? j : 0)]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {                         // branch-free: columns past the block meet xf = 0, rows past it xa = 0; what they leave in
          const dc v = bb[PIPE ? ch : 0][u];                  // sums that are never stored does not matter
          pr[u] += v.re * xf[ch].re - v.im * xf[ch].im; pi[u] += v.re * xf[ch].im + v.im * xf[ch].re;
          cr[ch] += v.re * xa[u].re - v.im * xa[u].im; ci[ch] += v.re * xa[u].im + v.im * xa[u].re;
        }
      }
    }
    {
      // the four rows' sums by the transposing butterfly (fmm_fold): eight vectors to one register on lane bits 0, 1, 2, then over bits 3, 4, 5;
      // lanes 0..7 hold (re, im) of rows i0, i0 + 4, i0 + 8, i0 + 12
      static_assert(U == 4, "the butterfly folds four rows");
      const int b2 = (lane >> 2) & 1;
      const bool bit0 = ((lane ^ b2) & 1) != 0, bit1 = (((lane >> 1) ^ b2) & 1) != 0, bit2 = b2 != 0;
      const double r0 = fmm_fold<0xB1>(pr[0], pi[0], bit0), r1 = fmm_fold<0xB1>(pr[1], pi[1], bit0);
      const double r2 = fmm_fold<0xB1>(pr[2], pi[2], bit0), r3 = fmm_fold<0xB1>(pr[3], pi[3], bit0);
      const double q0 = fmm_fold<0x4E>(r0, r1, bit1), q1 = fmm_fold<0x4E>(r2, r3, bit1);
      double z = fmm_fold<0x141>(q0, q1, bit2);
      z += fmm_dpp<0x128>(z);                                // row_ror:8
      z += __shfl_xor(z, 16, 64);
      z += __shfl_xor(z, 32, 64);
      const int i = i0 + 4 * ((bit1 ? 1 : 0) + (bit2 ? 2 : 0));
      if (lane < 8 && i < ns && !self_t) reinterpret_cast<double*>(prow + i)[bit0 ? 1 : 0] = z;
    }
    if constexpr (PIPE) {
      (void)more;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int u = 0; u < U; ++u) bb[ch][u] = bn[ch][u];
    }
  }
  if (both || self_t) {                                      // uniform over the block
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) if (ch < nch) cpart[(w * NCH + ch) * 64 + lane] = dc_make(cr[ch], ci[ch]);
    __syncthreads();
    for (int j = threadIdx.x; j < nf; j += 256) {
      const int ch = j >> 6, l = j & 63;
      double tr_ = 0.0, ti_ = 0.0;
      for (int q = 0; q < 4; ++q) { tr_ += cpart[(q * NCH + ch) * 64 + l].re; ti_ += cpart[(q * NCH + ch) * 64 + l].im; }
      pcol[j] = dc_make(tr_, ti_);
    }
  }
  __syncthreads();                                           // cpart and xrow are reused by the workgroup's next block
  }
}
// one workgroup per block; the block's width picks the body (round 5: one launch used to run the widest block's form -- 8 chunks, no
// prefetch -- for every block, although nine blocks in ten of a grid clustering are two chunks wide)
__global__ __launch_bounds__(256) void slfmm_near_wide_blocks_kernel(const int* __restrict__ eptr, const int* __restrict__ edof, const int* __restrict__ bsrc,
                                                                     const int* __restrict__ bfld, const long long* __restrict__ boff,
                                                                     const long long* __restrict__ broff, const long long* __restrict__ bcoff, int nblocks,
                                                                     const dc* __restrict__ bval, const dc* __restrict__ x, dc* __restrict__ part, int tmode) {
  __shared__ dc cpart[4 * 64 * FMM_NCH];
  __shared__ dc xrow[FMM_WIDE_ROWS];                          // round 4: x of the block's rows, gathered once (the row loop then holds block loads only)
  for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {    // grid-stride (see slfmm_near_blocks_kernel)
    const int f = bfld[b];
    const int nch = (eptr[f + 1] - eptr[f] + 63) >> 6;       // uniform over the workgroup
    if (nch <= 2) near_wide_block<2>(b, cpart, xrow, eptr, edof, bsrc, bfld, boff, broff, bcoff, bval, x, part, tmode);
    else near_wide_block<8>(b, cpart, xrow, eptr, edof, bsrc, bfld, boff, broff, bcoff, bval, x, part, tmode);
  }
}
// (Round 4, measured and removed: the wide blocks CHUNK-major -- one chunk's x and column sum per lane, the rows' sums waiting in LDS
// between chunks, 86 registers instead of 196 and twice the wavefronts per SIMD -- takes 233 us where the kernel above takes 160 on the
// 50k box: a row reduced over the lanes once per chunk costs more than the occupancy buys. profiles/r04_fmm_apply.md)
template <int WPC>
__global__ __launch_bounds__(256) void slfmm_near_gather_kernel(const int* __restrict__ eptr, const int* __restrict__ edof, const int* __restrict__ cptr,
                                                                const SlfmmEntry* __restrict__ cent, const dc* __restrict__ part, dc* __restrict__ y,
                                                                int overlap, int nclusters, const dc* __restrict__ yfar /* NULL, or the far field's share per dof (no overlap) */) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * (4 / WPC) + wave / WPC, w = wave % WPC;
  if (c >= nclusters) return;
  const int e0 = eptr[c], nc_ = eptr[c + 1] - e0, q0 = cptr[c], q1 = cptr[c + 1];
  constexpr int U = 8;
  for (int i = w * 64 + lane; i < nc_; i += 64 * WPC) {
    double sr = 0.0, si = 0.0;
    for (int q = q0; q < q1; q += U) {
      dc v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = part[cent[q + u < q1 ? q + u : q].poff + i];
#pragma unroll
      for (int u = 0; u < U; ++u) if (q + u < q1) { sr += v[u].re; si += v[u].im; }
    }
    dc* o = y + edof[e0 + i];
    if (overlap) { atomicAdd(&o->re, sr); atomicAdd(&o->im, si); }         // the dof is a row of several clusters
    else if (yfar) { const dc f = yfar[edof[e0 + i]]; *o = dc_make(sr + f.re, si + f.im); }   // near, then + far: the sum the one-stream order forms
    else *o = dc_make(sr, si);
  }
}

// The T and S matrices of the reference (slfmm.rs stores them per cluster) as one table E[listed element][sphere point] =
// w_p e^{+i k s_p.(x_j - C_c)}: T uses its conjugate phase, S the phase itself (and the other way round for the transpose). Stored,
// the upward and downward passes stream 16 B per (element, point) instead of evaluating a double-precision sincos for it.
__global__ __launch_bounds__(256) void slfmm_phase_table_kernel(BemGeom g, const int* __restrict__ eptr, const int* __restrict__ eidx, const double* __restrict__ cc,
                                                                const double* __restrict__ sc, const double* __restrict__ sw, int P, double k, dc* __restrict__ E) {
  const int c = blockIdx.x;
  const int e0 = eptr[c], n = eptr[c + 1] - e0;
  const double Cx = cc[3 * c], Cy = cc[3 * c + 1], Cz = cc[3 * c + 2];
  for (long long idx = threadIdx.x; idx < (long long)n * P; idx += 256) {
    const int j = (int)(idx / P), p = (int)(idx % P);
    const int e = eidx[e0 + j];
    const double sd = sc[3 * p] * (g.c[0][e] - Cx) + sc[3 * p + 1] * (g.c[1][e] - Cy) + sc[3 * p + 2] * (g.c[2][e] - Cz);
    double sn, cs; sincos(k * sd, &sn, &cs);
    E[(long long)e0 * P + idx] = dc_make(cs * sw[p], sn * sw[p]);
  }
}
// up[c][p] = sum_j (E.re, sgn E.im) x[dof_j]: 256 / P thread sets split the elements, their parts are added in set order
__global__ __launch_bounds__(256) void slfmm_up_tab_kernel(const int* __restrict__ eptr, const int* __restrict__ edof, const dc* __restrict__ E, int P, double sgn,
                                                           const dc* __restrict__ x, dc* __restrict__ up) {
  __shared__ dc part[256];
  constexpr int U = 8;
  const int c = blockIdx.x, tid = threadIdx.x;
  const int e0 = eptr[c], n = eptr[c + 1] - e0;
  const int QG = P <= 256 ? 256 / P : 1;
  for (int pb = 0; pb < P; pb += 256) {
    const int qg = P <= 256 ? tid / P : 0, p = pb + (P <= 256 ? tid % P : tid);
    double sr = 0.0, si = 0.0;
    if (qg < QG && p < P)
      for (int j = qg; j < n; j += QG * U) {
        dc ev[U], xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int jj = j + u * QG < n ? j + u * QG : j;
          ev[u] = E[(long long)(e0 + jj) * P + p]; xv[u] = x[edof[e0 + jj]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (j + u * QG < n) {
            const double er = ev[u].re, ei = sgn * ev[u].im;
            sr += er * xv[u].re - ei * xv[u].im; si += er * xv[u].im + ei * xv[u].re;
          }
      }
    if (P > 256) { if (p < P) up[(long long)c * P + p] = dc_make(sr, si); continue; }
    part[tid] = dc_make(sr, si);
    __syncthreads();
    if (tid < P) {
      double tr_ = 0.0, ti_ = 0.0;
      for (int q = 0; q < QG; ++q) { tr_ += part[q * P + tid].re; ti_ += part[q * P + tid].im; }
      up[(long long)c * P + tid] = dc_make(tr_, ti_);
    }
  }
}
// y[dof_j] += sum_p (E.re, sgn E.im) tr[c][p]: a wavefront per element (four at a time), lanes over the sphere points
__global__ __launch_bounds__(256) void slfmm_down_tab_kernel(const int* __restrict__ eptr, const int* __restrict__ edof, const dc* __restrict__ E, int P, double sgn,
                                                             const dc* __restrict__ tr, dc* __restrict__ y, int overlap /* 1: atomic adds; 2: plain stores (y is the far field's own buffer) */) {
  constexpr int U = 8;
  const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int e0 = eptr[c], n = eptr[c + 1] - e0;
  const dc* l = tr + (long long)c * P;
  for (int j0 = wave * U; j0 < n; j0 += 4 * U) {
    double sr[U], si[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { sr[u] = 0.0; si[u] = 0.0; }
    for (int p = lane; p < P; p += 64) {
      const dc lv = l[p];
      dc ev[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ev[u] = E[(long long)(e0 + (j0 + u < n ? j0 + u : j0)) * P + p];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const double er = ev[u].re, ei = sgn * ev[u].im;
        sr[u] += er * lv.re - ei * lv.im; si[u] += er * lv.im + ei * lv.re;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double a = fmm_set_sum(sr[u], 64), b = fmm_set_sum(si[u], 64);
      if (lane == 0 && j0 + u < n) {
        dc* o = y + edof[e0 + j0 + u];
        if (overlap == 1) { atomicAdd(&o->re, a); atomicAdd(&o->im, b); }
        else if (overlap == 2) *o = dc_make(a, b);
        else { o->re += a; o->im += b; }
      }
    }
  }
}
// Round 4: the phases RECOMPUTED, fast. The stored table costs 16 B per (element, sphere point) and pass -- 103 MB per pass on the 50k
// box, beside a near field that is memory-bound already -- while the phase itself is 3 FMAs, one multiplication and a sin / cos of a
// SMALL argument (|k s.(x - C)| <= k x cluster radius): sincos_bounded (ma_device_math.hpp: two-FMA Cody-Waite + the fdlibm kernels, no
// library fall-back) takes ~35 FP64 instructions where libm's sincos -- which the first version of these passes called, and lost to the
// table -- takes hundreds. The cluster's element offsets (and x, or the local expansion) are staged in LDS once per workgroup.
// MA_FMM_STORE_PHASES: 2 (default) these kernels, no table; 1 the table; 0 the libm form.
constexpr int FMM_FAST_ELEMS = 512;                          // elements of a cluster staged per pass (larger clusters walk in chunks)
// up[c][p] = w_p sum_j exp(i sgn k s_p.(x_j - C)) x[dof_j]: thread sets split the elements, their parts are added in set order
__global__ __launch_bounds__(256) void slfmm_up_fast_kernel(BemGeom g, const int* __restrict__ eptr, const int* __restrict__ eidx, const int* __restrict__ edof,
                                                            const double* __restrict__ cc, const double* __restrict__ sc, const double* __restrict__ sw,
                                                            int P, double k, double sgn, const dc* __restrict__ x, dc* __restrict__ up) {
  __shared__ double ex[FMM_FAST_ELEMS], ey[FMM_FAST_ELEMS], ez[FMM_FAST_ELEMS];
  __shared__ dc xv[FMM_FAST_ELEMS];
  __shared__ dc part[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const int e0 = eptr[c], n = eptr[c + 1] - e0;
  const double Cx = cc[3 * c], Cy = cc[3 * c + 1], Cz = cc[3 * c + 2];
  const int QG = P <= 256 ? 256 / P : 1;                     // thread sets over the elements (P sphere points each)
  for (int pb = 0; pb < P; pb += 256) {
    const int qg = P <= 256 ? tid / P : 0, p = pb + (P <= 256 ? tid % P : tid);
    const bool act = qg < QG && p < P;
    const double sx = act ? k * sc[3 * p] : 0.0, sy = act ? k * sc[3 * p + 1] : 0.0, sz = act ? k * sc[3 * p + 2] : 0.0;
    double sr = 0.0, si = 0.0;
    for (int j0 = 0; j0 < n; j0 += FMM_FAST_ELEMS) {
      const int m = min(FMM_FAST_ELEMS, n - j0);
      __syncthreads();
      for (int j = tid; j < m; j += 256) {
        const int e = eidx[e0 + j0 + j];
        ex[j] = g.c[0][e] - Cx; ey[j] = g.c[1][e] - Cy; ez[j] = g.c[2][e] - Cz; xv[j] = x[edof[e0 + j0 + j]];
      }
      __syncthreads();
      if (act)
        for (int j = qg; j < m; j += QG) {
          double sn, cs;
          sincos_bounded(__builtin_fma(sx, ex[j], __builtin_fma(sy, ey[j], sz * ez[j])), sn, cs);
          const double ei = sgn * sn;
          const dc v = xv[j];
          sr += cs * v.re - ei * v.im; si += cs * v.im + ei * v.re;
        }
    }
    if (P > 256) { if (p < P) { const double w = sw[p]; up[(long long)c * P + p] = dc_make(w * sr, w * si); } continue; }
    __syncthreads();
    part[tid] = dc_make(sr, si);
    __syncthreads();
    if (tid < P) {
      double tr_ = 0.0, ti_ = 0.0;
      for (int q = 0; q < QG; ++q) { tr_ += part[q * P + tid].re; ti_ += part[q * P + tid].im; }
      const double w = sw[tid];
      up[(long long)c * P + tid] = dc_make(w * tr_, w * ti_);
    }
  }
}
// y[dof_j] (+)= sum_p w_p exp(i sgn k s_p.(x_j - C)) tr[c][p]: a wavefront per element (four at a time), lanes over the sphere points, the
// cluster's local expansion and sphere rule in LDS; mode 0 add, 1 atomic add (overlapping leaves), 2 store (y = the far field's own vector)
constexpr int FMM_FAST_PTS = 1024;
__global__ __launch_bounds__(256) void slfmm_down_fast_kernel(BemGeom g, const int* __restrict__ eptr, const int* __restrict__ eidx, const int* __restrict__ edof,
                                                              const double* __restrict__ cc, const double* __restrict__ sc, const double* __restrict__ sw,
                                                              int P, double k, double sgn, const dc* __restrict__ tr, dc* __restrict__ y, int mode) {
  __shared__ double px[FMM_FAST_PTS], py[FMM_FAST_PTS], pz[FMM_FAST_PTS];
  __shared__ dc lw[FMM_FAST_PTS];                              // w_p tr[c][p]
  const int c = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int e0 = eptr[c], n = eptr[c + 1] - e0;
  const double Cx = cc[3 * c], Cy = cc[3 * c + 1], Cz = cc[3 * c + 2];
  for (int p = tid; p < P; p += 256) {                         // P <= FMM_FAST_PTS (the launcher's choice)
    px[p] = k * sc[3 * p]; py[p] = k * sc[3 * p + 1]; pz[p] = k * sc[3 * p + 2];
    const dc l = tr[(long long)c * P + p]; const double w = sw[p];
    lw[p] = dc_make(w * l.re, w * l.im);
  }
  __syncthreads();
  constexpr int U = 4;
  for (int j0 = wave * U; j0 < n; j0 += 4 * U) {
    double dx[U], dy[U], dz[U], sr[U], si[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = eidx[e0 + (j0 + u < n ? j0 + u : j0)];
      dx[u] = g.c[0][e] - Cx; dy[u] = g.c[1][e] - Cy; dz[u] = g.c[2][e] - Cz; sr[u] = 0.0; si[u] = 0.0;
    }
    for (int p = lane; p < P; p += 64) {
      const double ax = px[p], ay = py[p], az = pz[p];
      const dc l = lw[p];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        double sn, cs;
        sincos_bounded(__builtin_fma(ax, dx[u], __builtin_fma(ay, dy[u], az * dz[u])), sn, cs);
        const double ei = sgn * sn;
        sr[u] += cs * l.re - ei * l.im; si[u] += cs * l.im + ei * l.re;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double a = fmm_set_sum(sr[u], 64), b = fmm_set_sum(si[u], 64);
      if (lane == 0 && j0 + u < n) {
        dc* o = y + edof[e0 + j0 + u];
        if (mode == 1) { atomicAdd(&o->re, a); atomicAdd(&o->im, b); }
        else if (mode == 2) *o = dc_make(a, b);
        else { o->re += a; o->im += b; }
      }
    }
  }
}
static int slfmm_launch_up(const ma_slfmm* S, double sgn, const dc* x, hipStream_t st) {
  if (S->fast_phases) hipLaunchKernelGGL(slfmm_up_fast_kernel, dim3(S->nc), dim3(256), 0, st, S->plan->geom, S->d_eptr, S->d_eidx, S->d_edof, S->d_cc, S->d_sc, S->d_sw, S->P, S->k, sgn, x,
                                         reinterpret_cast<dc*>(S->d_up));
  else if (S->d_phase) hipLaunchKernelGGL(slfmm_up_tab_kernel, dim3(S->nc), dim3(256), 0, st, S->d_eptr, S->d_edof, reinterpret_cast<const dc*>(S->d_phase), S->P, sgn, x,
                                     reinterpret_cast<dc*>(S->d_up));
  else hipLaunchKernelGGL(slfmm_up_kernel, dim3(S->nc), dim3(256), 0, st, S->plan->geom, S->d_eptr, S->d_eidx, S->d_edof, S->d_cc, S->d_sc, S->d_sw, S->P, S->k, sgn, x,
                          reinterpret_cast<dc*>(S->d_up));
  MA_HIP(hipGetLastError());
  return MA_OK;
}
static int slfmm_launch_down(const ma_slfmm* S, double sgn, dc* y, hipStream_t st, bool store = false) {
  if (S->fast_phases) hipLaunchKernelGGL(slfmm_down_fast_kernel, dim3(S->nc), dim3(256), 0, st, S->plan->geom, S->d_eptr, S->d_eidx, S->d_edof, S->d_cc, S->d_sc, S->d_sw, S->P, S->k, sgn,
                                         reinterpret_cast<const dc*>(S->d_tr), y, store ? 2 : (S->overlap ? 1 : 0));
  else if (S->d_phase) hipLaunchKernelGGL(slfmm_down_tab_kernel, dim3(S->nc), dim3(256), 0, st, S->d_eptr, S->d_edof, reinterpret_cast<const dc*>(S->d_phase), S->P, sgn,
                                     reinterpret_cast<const dc*>(S->d_tr), y, store ? 2 : (S->overlap ? 1 : 0));
  else hipLaunchKernelGGL(slfmm_down_kernel, dim3(S->nc), dim3(256), 0, st, S->plan->geom, S->d_eptr, S->d_eidx, S->d_edof, S->d_cc, S->d_sc, S->d_sw, S->P, S->k, sgn,
                          reinterpret_cast<const dc*>(S->d_tr), y, S->overlap ? 1 : 0);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// the element pair behind every near entry, block by block (row-major inside a block); the self term's slot gets a neighbouring
// panel as a stand-in partner, its value is overwritten by slfmm_fix_diag_kernel
__global__ __launch_bounds__(256) void slfmm_pairs_kernel(const int* __restrict__ eptr, const int* __restrict__ eidx, const int* __restrict__ bsrc,
                                                          const int* __restrict__ bfld, const long long* __restrict__ boff, int np, int2* __restrict__ pairs) {
  const int b = blockIdx.x;
  const int a = bsrc[b], f = bfld[b];
  const int a0 = eptr[a], ns = eptr[a + 1] - a0, f0 = eptr[f], nf = eptr[f + 1] - f0;
  int2* out = pairs + boff[b];
  for (long long idx = threadIdx.x; idx < (long long)ns * nf; idx += 256) {
    const int i = (int)(idx / nf), j = (int)(idx % nf);
    const int se = eidx[a0 + i]; int fe = eidx[f0 + j];
    if (a == f && se == fe) fe = (se + 1) % np == se ? se : (se + 1) % np;
    out[idx] = make_int2(se, fe);
  }
}

// pass 0: both passes of the near field on `st`; 1: the blocks' products only (-> the partial sums); 2: the per-cluster pass only
// (partial sums -> y, plus yfar when given)
static int slfmm_launch_near(const ma_slfmm* S, const dc* x, dc* y, int tmode, hipStream_t st, int pass = 0, const dc* yfar = nullptr) {
  const int avg = S->nc > 0 ? (int)((S->h_eptr.empty() ? 0 : S->h_eptr.back()) / S->nc) : 0;
  int G = 8; while (G < 64 && G < avg) G <<= 1;
  if (S->d_part) {
    const dc* bv = reinterpret_cast<const dc*>(S->d_bval); dc* part = reinterpret_cast<dc*>(S->d_part);
    // (round 4 measured a cap on the grid beside the far chain: every cap from 1 to 8 workgroups per CU loses -- profiles/r04_fmm_apply.md)
    auto grid = [&](long long want) { return dim3((unsigned)want); };
    if (pass == 2) { /* second pass only */ }
    else if (S->d_bdesc) {
      dc* xp = reinterpret_cast<dc*>(S->d_xp);
      hipLaunchKernelGGL(fmm_cluster_order_kernel, dim3((unsigned)((S->listed + 255) / 256)), dim3(256), 0, st, S->d_edof, S->listed, x, xp);
      hipLaunchKernelGGL(slfmm_near_leaf_blocks_kernel, grid((S->nblocks + 3) / 4), dim3(256), 0, st, S->d_bdesc, S->nblocks, bv, xp, part, tmode);
    }
    else if (S->max_block <= 64 * 64) hipLaunchKernelGGL(slfmm_near_blocks_kernel<1>, grid((S->nblocks + 3) / 4), dim3(256), 0, st, S->d_eptr, S->d_edof, S->d_bsrc, S->d_bfld,
                                                   S->d_boff, S->d_broff, S->d_bcoff, S->nblocks, bv, x, part, tmode);
    else if (S->max_width <= 64 * FMM_NCH && S->max_rows <= FMM_WIDE_ROWS) {
      hipLaunchKernelGGL(slfmm_near_wide_blocks_kernel, grid(S->nblocks), dim3(256), 0, st, S->d_eptr, S->d_edof, S->d_bsrc, S->d_bfld, S->d_boff, S->d_broff,
                         S->d_bcoff, S->nblocks, bv, x, part, tmode);
    }
    else hipLaunchKernelGGL(slfmm_near_blocks_kernel<4>, grid(S->nblocks), dim3(256), 0, st, S->d_eptr, S->d_edof, S->d_bsrc, S->d_bfld, S->d_boff, S->d_broff,
                            S->d_bcoff, S->nblocks, bv, x, part, tmode);
    MA_HIP(hipGetLastError());
    if (pass == 1) return MA_OK;
    if (avg <= 48) hipLaunchKernelGGL(slfmm_near_gather_kernel<1>, dim3((unsigned)((S->nc + 3) / 4)), dim3(256), 0, st, S->d_eptr, S->d_edof, S->d_cptr, S->d_cent, part, y,
                                      S->overlap ? 1 : 0, S->nc, yfar);
    else hipLaunchKernelGGL(slfmm_near_gather_kernel<4>, dim3((unsigned)S->nc), dim3(256), 0, st, S->d_eptr, S->d_edof, S->d_cptr, S->d_cent, part, y, S->overlap ? 1 : 0, S->nc, yfar);
    MA_HIP(hipGetLastError());
    return MA_OK;
  }
  MA_REQUIRE(pass == 0, MA_ERR_INVALID, "the near field without stored block sums has one pass");
  if (avg <= 48) hipLaunchKernelGGL(slfmm_near_kernel<1>, dim3((unsigned)((S->nc + 3) / 4)), dim3(256), 0, st, S->d_eptr, S->d_edof, S->d_cptr, S->d_cent,
                                    reinterpret_cast<const dc*>(S->d_bval), x, y, tmode, S->overlap ? 1 : 0, S->nc, G);
  else hipLaunchKernelGGL(slfmm_near_kernel<4>, dim3((unsigned)S->nc), dim3(256), 0, st, S->d_eptr, S->d_edof, S->d_cptr, S->d_cent,
                          reinterpret_cast<const dc*>(S->d_bval), x, y, tmode, S->overlap ? 1 : 0, S->nc, G);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// the diagonal of the self blocks (SparseNearfieldIlu::from_slfmm, math-bem/src/core/solver/fmm_interface.rs:249-297): diag[dof_i] = B_cc[i][i]
__global__ __launch_bounds__(256) void slfmm_self_diag_kernel(const int* __restrict__ eptr, const int* __restrict__ edof, const int* __restrict__ cptr,
                                                              const SlfmmEntry* __restrict__ cent, const dc* __restrict__ bval, dc* __restrict__ diag) {
  const int c = blockIdx.x;
  const int e0 = eptr[c], nc_ = eptr[c + 1] - e0;
  for (int q = cptr[c]; q < cptr[c + 1]; ++q) {
    const SlfmmEntry en = cent[q];
    if (en.other != c) continue;
    for (int i = threadIdx.x; i < nc_; i += 256) diag[edof[e0 + i]] = bval[en.boff + (long long)i * nc_ + i];
  }
}
int slfmm_self_diagonal(ma_slfmm* S, c64* d_diag, hipStream_t st) {
  MA_HIP(hipSetDevice(S->device));
  MA_HIP(hipMemsetAsync(d_diag, 0, sizeof(c64) * (size_t)S->n, st));
  if (S->nc > 0) hipLaunchKernelGGL(slfmm_self_diag_kernel, dim3(S->nc), dim3(256), 0, st, S->d_eptr, S->d_edof, S->d_cptr, S->d_cent, reinterpret_cast<const dc*>(S->d_bval),
                                    reinterpret_cast<dc*>(d_diag));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int slfmm_phase_mode(const ma_slfmm* S) { return S->fast_phases ? 2 : (S->d_phase ? 1 : 0); }
long long slfmm_num_dofs(const ma_slfmm* S) { return S->n; }
int slfmm_device(const ma_slfmm* S) { return S->device; }

void slfmm_destroy(ma_slfmm* S) {
  if (!S) return;
  (void)hipSetDevice(S->device);
  void* p[] = {S->d_eptr, S->d_eidx, S->d_edof, S->d_cc, S->d_sc, S->d_sw, S->d_bval, S->d_cptr, S->d_cent, S->d_fptr, S->d_foth, S->d_fval, S->d_tptr, S->d_toth,
               S->d_tval, S->d_up, S->d_tr, S->d_fdense, S->d_tdense, S->d_bsrc, S->d_bfld, S->d_boff, S->d_broff, S->d_bcoff, S->d_part, S->d_phase, S->d_bdesc, S->d_xp};
  for (void* q : p) if (q) (void)hipFree(q);
  if (S->d_yfar) (void)hipFree(S->d_yfar);
  if (S->st2) { (void)hipStreamSynchronize(S->st2); (void)hipStreamDestroy(S->st2); }
  if (S->ev_fork) (void)hipEventDestroy(S->ev_fork);
  if (S->ev_near) (void)hipEventDestroy(S->ev_near);
  if (S->st3) { (void)hipStreamSynchronize(S->st3); (void)hipStreamDestroy(S->st3); }
  if (S->ev_up) (void)hipEventDestroy(S->ev_up);
  if (S->ev_leaf) (void)hipEventDestroy(S->ev_leaf);
  delete S;
}

static int slfmm_create_ex(ma_bem_plan* plan, const ma_clusters_t* cl, const ma_physics_t* physics, int n_theta, int n_phi, int n_terms, bool free_term, bool allow_overlap,
                           ma_slfmm** out);
int slfmm_create(ma_bem_plan* plan, const ma_clusters_t* cl, const ma_physics_t* physics, int n_theta, int n_phi, int n_terms, ma_slfmm** out) {
  return slfmm_create_ex(plan, cl, physics, n_theta, n_phi, n_terms, true, false, out);
}
// free_term: slfmm.rs adds gamma / 2 (velocity panels) to the diagonal of the self blocks (:514-533), mlfmm.rs does not (:590-644);
// allow_overlap: mlfmm.rs' leaves may share elements
static int slfmm_create_ex(ma_bem_plan* plan, const ma_clusters_t* cl, const ma_physics_t* physics, int n_theta, int n_phi, int n_terms, bool free_term, bool allow_overlap,
                           ma_slfmm** out) {
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
      MA_REQUIRE(allow_overlap || !seen[(size_t)e], MA_ERR_UNSUPPORTED, "element %d belongs to more than one cluster", e);
      MA_REQUIRE(bct[(size_t)e] == 0, MA_ERR_UNSUPPORTED, "element %d: only velocity-type boundary conditions (compute_near_block's coefficient, slfmm.rs:583-584)", e);
      seen[(size_t)e] = 1;
    }
    for (int q = cl->near_ptr[c]; q < cl->near_ptr[c + 1]; ++q) MA_REQUIRE(cl->near_idx && cl->near_idx[q] >= 0 && cl->near_idx[q] < nc, MA_ERR_INVALID, "near cluster index out of range");
    for (int q = cl->far_ptr[c]; q < cl->far_ptr[c + 1]; ++q) MA_REQUIRE(cl->far_idx && cl->far_idx[q] >= 0 && cl->far_idx[q] < nc, MA_ERR_INVALID, "far cluster index out of range");
  }
#ifdef MA_DIAGNOSTICS
  const bool timing = getenv("MA_FMM_TIMING") != nullptr;    // diagnostic build only: phase times of the setup on stderr
#else
  const bool timing = false;
#endif
  auto tnow = []() { return std::chrono::steady_clock::now(); };
  auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto tt0 = tnow();
  ma_slfmm* S = new (std::nothrow) ma_slfmm(); MA_REQUIRE(S, MA_ERR_NOMEM, "host allocation failed");
  S->device = plan->device; S->plan = plan; S->n = plan->nd; S->nc = nc; S->P = n_theta * n_phi; S->k = physics->wave_number; S->overlap = allow_overlap;
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
  // the (source element, field element) pair of every near entry is written on the device (slfmm_pairs_kernel); the host keeps the
  // positions of the self terms: entry (i, i) of every diagonal block (a cluster lists an element once)
  std::vector<long long> dpos; std::vector<int> dpanel;
  for (size_t b = 0; b < bsrc.size(); ++b) {
    if (bsrc[b] != bfld[b]) continue;
    const int ci = bsrc[b];
    const int ns = eptr[(size_t)ci + 1] - eptr[(size_t)ci];
    for (int i = 0; i < ns; ++i) { dpos.push_back(boff[b] + (long long)i * ns + i); dpanel.push_back(eidx[(size_t)eptr[(size_t)ci] + i]); }
  }
  const auto tt1 = tnow();
  // ---- D entries (:663-721)
  // by source (the transpose apply): the far lists as they come; by field (the forward apply): the same pairs counted into rows, source
  // ascending within a row (the order the reference's loop appends them in)
  std::vector<int> fptr((size_t)nc + 1, 0), foth, tptr(cl->far_ptr, cl->far_ptr + nc + 1), toth; std::vector<c64> fval, tval;
  {
    const int order = std::max(n_terms, 2);
    const size_t npairs = (size_t)cl->far_ptr[nc];
    // one spherical Hankel evaluation per far pair: independent, on host threads, each into its own slot
    tval.resize(npairs); if (npairs) toth.assign(cl->far_idx, cl->far_idx + npairs);
    std::vector<int> bad((size_t)nc, -1);
    host_parallel_for(nc, 16, [&](long long c0, long long c1) {
      std::vector<double> hr, hi;
      for (long long i = c0; i < c1; ++i)
        for (int q = cl->far_ptr[i]; q < cl->far_ptr[i + 1]; ++q) {
          const int j = cl->far_idx[q];
          const double dx = cc[(size_t)3 * i] - cc[(size_t)3 * j], dy = cc[(size_t)3 * i + 1] - cc[(size_t)3 * j + 1], dz = cc[(size_t)3 * i + 2] - cc[(size_t)3 * j + 2];
          const double r = std::sqrt(dx * dx + dy * dy + dz * dz);
          if (!(r > 0.0)) { if (bad[(size_t)i] < 0) bad[(size_t)i] = j; continue; }
          spherical_hankel_first_kind(order, S->k * r, 1.0, hr, hi);
          tval[(size_t)q] = c64{-hi[0] * S->k, hr[0] * S->k};                      // h_0 * (i k)
        }
    });
    for (int i = 0; i < nc; ++i) if (bad[(size_t)i] >= 0) { set_error("far clusters %d and %d share their centre", i, bad[(size_t)i]); return fail(MA_ERR_INVALID); }
    for (size_t q = 0; q < npairs; ++q) fptr[(size_t)toth[q] + 1] += 1;
    for (int c = 0; c < nc; ++c) fptr[(size_t)c + 1] += fptr[(size_t)c];
    foth.resize(npairs); fval.resize(npairs);
    std::vector<int> cur(fptr.begin(), fptr.end() - 1);
    for (int i = 0; i < nc; ++i)
      for (int q = cl->far_ptr[i]; q < cl->far_ptr[i + 1]; ++q) { const int pos = cur[(size_t)toth[(size_t)q]]++; foth[(size_t)pos] = i; fval[(size_t)pos] = tval[(size_t)q]; }
  }
  // ---- per-cluster views of the blocks: as source (rows of the block) or as field of an off-diagonal block (its transpose)
  std::vector<std::vector<SlfmmEntry>> views((size_t)nc);
  std::vector<long long> broff(bsrc.size()), bcoff(bsrc.size()); long long npart = 0, max_block = 0;
  for (size_t b = 0; b < bsrc.size(); ++b) {
    const long long ns = eptr[(size_t)bsrc[b] + 1] - eptr[(size_t)bsrc[b]], nf = eptr[(size_t)bfld[b] + 1] - eptr[(size_t)bfld[b]];
    broff[b] = npart; npart += ns;
    bcoff[b] = npart; if (bsrc[b] != bfld[b]) npart += nf;
    max_block = std::max(max_block, ns * nf); S->max_width = std::max(S->max_width, nf); S->max_rows = std::max(S->max_rows, ns);
    views[(size_t)bsrc[b]].push_back({boff[b], broff[b], bfld[b], 0});
    if (bsrc[b] != bfld[b]) views[(size_t)bfld[b]].push_back({boff[b], bcoff[b], bsrc[b], 1});
  }
  std::vector<int> cptr(1, 0); std::vector<SlfmmEntry> cent;
  for (const auto& v : views) { cent.insert(cent.end(), v.begin(), v.end()); cptr.push_back((int)cent.size()); }
  const auto tt2 = tnow();
  int rc = MA_OK;
#define UP(dst, src) if (!rc) rc = upload(&S->dst, src)
  { std::vector<char> seen_dof((size_t)S->n, 0); bool once = (long long)edof.size() == S->n;
    for (size_t q = 0; once && q < edof.size(); ++q) { const long long dq = edof[q]; if (dq < 0 || dq >= S->n || seen_dof[(size_t)dq]) once = false; else seen_dof[(size_t)dq] = 1; }
    S->covers_all = once && !allow_overlap; }
  UP(d_eptr, eptr); UP(d_eidx, eidx); UP(d_edof, edof); UP(d_cc, cc); UP(d_sc, sc); UP(d_sw, sw); UP(d_cptr, cptr); UP(d_cent, cent);
  UP(d_fptr, fptr); UP(d_foth, foth); UP(d_fval, fval); UP(d_tptr, tptr); UP(d_toth, toth); UP(d_tval, tval);
#undef UP
  if (!rc) rc = fmm_dense_from_lists(fptr, foth, fval, nc, &S->d_fdense);
  { const char* e = getenv("MA_FMM_NEAR_BLOCKS");           // =0: the per-cluster kernel (every off-diagonal block read from both sides)
    S->nblocks = (int)bsrc.size(); S->max_block = max_block;
    if (!rc) rc = upload(&S->d_bsrc, bsrc);
    if (!rc) rc = upload(&S->d_bfld, bfld);
    if (!rc) rc = upload(&S->d_boff, boff);
    if (!rc && !(e && atoi(e) == 0) && !bsrc.empty()) {
      if (!rc) rc = upload(&S->d_broff, broff);
      if (!rc) rc = upload(&S->d_bcoff, bcoff);
      if (!rc && hipMalloc(&S->d_part, sizeof(c64) * (size_t)std::max(npart, 1LL)) != hipSuccess) { set_error("near-field partial sums"); rc = MA_ERR_NOMEM; }
      if (!rc && S->max_rows <= 64 && S->max_width <= 64 && S->max_rows > 0) {       // leaf-sized blocks: one record per block, x in cluster order
        std::vector<NearBlockDesc> bd(bsrc.size());
        for (size_t b = 0; b < bsrc.size(); ++b) {
          const int a = bsrc[b], f = bfld[b];
          bd[b] = {boff[b], broff[b], bcoff[b], eptr[(size_t)a], eptr[(size_t)f], eptr[(size_t)a + 1] - eptr[(size_t)a], eptr[(size_t)f + 1] - eptr[(size_t)f], a != f ? 1 : 0, 0};
        }
        bool all = true;
        for (const NearBlockDesc& q : bd) if (q.ns <= 0 || q.nf <= 0) all = false;      // an empty cluster: the general kernel
        S->listed = (long long)eptr.back();
        if (all) {
          rc = upload(&S->d_bdesc, bd);
          if (!rc && hipMalloc(&S->d_xp, sizeof(c64) * (size_t)std::max<long long>(S->listed, 1)) != hipSuccess) { set_error("near-field x copy"); rc = MA_ERR_NOMEM; }
        }
      }
    } }
  if (!rc && free_term) rc = fmm_dense_from_lists(tptr, toth, tval, nc, &S->d_tdense);   // the multi-level operator's leaf (no free term) is never applied transposed
  if (rc) return fail(rc);
  const auto tt3 = tnow();
  hipError_t e = hipMalloc(&S->d_bval, sizeof(c64) * (size_t)std::max<long long>(tot, 1));
  if (e == hipSuccess) e = hipMalloc(&S->d_up, sizeof(c64) * (size_t)nc * (size_t)P);
  if (e == hipSuccess) e = hipMalloc(&S->d_tr, sizeof(c64) * (size_t)nc * (size_t)P);
  int2* d_pairs = nullptr; long long* d_dpos = nullptr; int* d_dpanel = nullptr; c64* d_self = nullptr;
  if (e == hipSuccess) e = hipMalloc(&d_pairs, sizeof(int2) * (size_t)std::max<long long>(tot, 1));
  if (e == hipSuccess && S->nblocks > 0) {
    hipLaunchKernelGGL(slfmm_pairs_kernel, dim3((unsigned)S->nblocks), dim3(256), 0, nullptr, S->d_eptr, S->d_eidx, S->d_bsrc, S->d_bfld, S->d_boff, np, d_pairs);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMalloc(&d_self, sizeof(c64) * (size_t)np);
  auto drop = [&]() { if (d_pairs) (void)hipFree(d_pairs); if (d_dpos) (void)hipFree(d_dpos); if (d_dpanel) (void)hipFree(d_dpanel); if (d_self) (void)hipFree(d_self); };
  if (e != hipSuccess) { set_error("SLFMM workspace: %s", hipGetErrorString(e)); drop(); return fail(MA_ERR_NOMEM); }
  rc = upload(&d_dpos, dpos); if (!rc) rc = upload(&d_dpanel, dpanel);
  const auto tt4 = tnow();
  // coefficient of compute_near_block (:583-584): dg_dn gamma tau + d2g beta with beta = i h / k (types.rs:64-70), sign +1
  BemPhys ph;
  const double bim = physics->tau > 0.0 ? physics->harmonic_factor / physics->wave_number : 0.0;
  if (!rc) rc = ma_bem_make_phys(plan, physics, 0.0, bim, &ph);
  ph.sign = 1.0;
  if (!rc) rc = bem_launch_near_list_values(plan->geom, ph, d_pairs, tot, S->d_bval, nullptr);
  if (!rc) rc = bem_launch_self_list_values(plan->geom, ph, d_self, nullptr);
  if (!rc && !dpos.empty()) {
    hipLaunchKernelGGL(slfmm_fix_diag_kernel, dim3((unsigned)((dpos.size() + 255) / 256)), dim3(256), 0, nullptr, (int)dpos.size(), d_dpos, d_dpanel,
                       reinterpret_cast<const dc*>(d_self), free_term ? physics->gamma : 0.5 * physics->gamma, reinterpret_cast<dc*>(S->d_bval));
    if (hipGetLastError() != hipSuccess) { set_error("SLFMM diagonal kernel failed"); rc = MA_ERR_HIP; }
  }
  { const char* ev = getenv("MA_FMM_STORE_PHASES");
    const size_t listed = (size_t)eptr.back();
    int mode = ev ? atoi(ev) : 2;
    // the fast form's arguments are k x an element-to-centre distance (far below the 2^30 its reduction allows); P must fit its LDS
    // arrays -- a sphere rule of more than FMM_FAST_PTS points gets the stored table instead (it used to fall to the libm
    // form, the slowest of the three, and lost the two-stream apply with it)
    if (mode == 2 && P <= FMM_FAST_PTS) S->fast_phases = true;
    else if (mode == 2) mode = 1;
    if (!rc && mode == 1 && listed > 0 && hipMalloc(&S->d_phase, sizeof(c64) * listed * (size_t)P) == hipSuccess) {
      hipLaunchKernelGGL(slfmm_phase_table_kernel, dim3(nc), dim3(256), 0, nullptr, plan->geom, S->d_eptr, S->d_eidx, S->d_cc, S->d_sc, S->d_sw, P, S->k,
                         reinterpret_cast<dc*>(S->d_phase));
      if (hipGetLastError() != hipSuccess) { set_error("SLFMM phase table kernel failed"); rc = MA_ERR_HIP; }
    } else if (!rc) { (void)hipGetLastError(); S->d_phase = nullptr; }       // no room: the passes evaluate the phases
  }
  if (!rc && hipDeviceSynchronize() != hipSuccess) { set_error("SLFMM near-field kernels failed"); rc = MA_ERR_HIP; }
  drop();
  if (timing) fprintf(stderr, "[slfmm build] nc %d: checks + block lists %.0f, translation factors + far lists %.0f, uploads + dense D %.0f, buffers + element pairs %.0f, near-field kernels %.0f ms\n", nc, tms(tt0, tt1), tms(tt1, tt2), tms(tt2, tt3), tms(tt3, tt4), tms(tt4, tnow()));
  if (rc) return fail(rc);
  *out = S;
  return MA_OK;
}

// the second stream, its events and the far field's own result vector: made at the first apply that wants them
static bool slfmm_overlap_ready(ma_slfmm* S) {
  if (S->overlap_streams < 0) {
    const char* e = getenv("MA_FMM_OVERLAP");
    S->overlap_streams = 0;
    if (!(e && atoi(e) == 0) && S->d_part && (S->d_phase || S->fast_phases)) {
      // the near blocks' stream at the LOWEST priority: its tens of thousands of short workgroups otherwise keep the far chain's few
      // kernels waiting for slots (the first upward-pass kernel took 350 us instead of 15 beside them)
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      lo = 0;                                               // normal priority (the lowest was measured worse: r04_fmm_apply.md)
      bool ok = hipStreamCreateWithPriority(&S->st2, hipStreamNonBlocking, lo) == hipSuccess && hipEventCreateWithFlags(&S->ev_fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&S->ev_near, hipEventDisableTiming) == hipSuccess && hipMalloc(&S->d_yfar, sizeof(c64) * (size_t)S->n) == hipSuccess &&
                hipStreamCreateWithFlags(&S->st3, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&S->ev_up, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&S->ev_leaf, hipEventDisableTiming) == hipSuccess;
      if (ok) S->overlap_streams = 1; else (void)hipGetLastError();
    }
  }
  return S->overlap_streams == 1;
}
// fork: the near field on the second stream, ordered after what `st` holds so far (x is ready, y is cleared where it has to be).
// whole = false (the single-level operator): the blocks' products only -- its downward pass is as long as the near blocks and runs beside
// them into d_yfar; the short per-cluster pass then adds the two on `st`. whole = true (the multi-level operator, round 5): the
// per-cluster pass too, into y -- its far chain is the longer side, and that pass (55 us on the 50k box) used to sit behind it.
static int slfmm_fork_near(ma_slfmm* S, const dc* x, dc* y, int tmode, hipStream_t st, bool whole) {
  MA_HIP(hipEventRecord(S->ev_fork, st));
  MA_HIP(hipStreamWaitEvent(S->st2, S->ev_fork, 0));
  int rc = slfmm_launch_near(S, x, y, tmode, S->st2, whole ? 0 : 1);
  if (rc) return rc;
  MA_HIP(hipEventRecord(S->ev_near, S->st2));
  return MA_OK;
}
// join. whole = false: `st` (which has the far field in d_yfar, or will add it to y itself when elements sit in several clusters) takes
// the near field's second pass, which adds the two. whole = true: the near field is in y, the downward pass ADDS the far field's share.
// Either way near, then + far: the sum the one-stream order forms, bit for bit.
static int slfmm_join_near(ma_slfmm* S, dc* y, double s_dn, hipStream_t st, bool whole) {
  MA_HIP(hipStreamWaitEvent(st, S->ev_near, 0));
  if (whole) return slfmm_launch_down(S, s_dn, y, st);
  if (S->overlap) {                                            // atomic adds into the cleared y: near sums, then the far field
    int rc = slfmm_launch_near(S, nullptr, y, 0, st, 2);
    if (!rc) rc = slfmm_launch_down(S, s_dn, y, st);
    return rc;
  }
  return slfmm_launch_near(S, nullptr, y, 0, st, 2, reinterpret_cast<const dc*>(S->d_yfar));
}

int slfmm_apply(ma_slfmm* S, const c64* d_x, c64* d_y, int transpose, hipStream_t st) {
  MA_HIP(hipSetDevice(S->device));
  if (!(S->covers_all && S->d_part)) MA_HIP(hipMemsetAsync(d_y, 0, sizeof(c64) * (size_t)S->n, st));   // dofs outside every cluster receive nothing (none: the second pass writes all of y)
  const dc* x = reinterpret_cast<const dc*>(d_x); dc* y = reinterpret_cast<dc*>(d_y);
  if (slfmm_overlap_ready(S)) {
    const double s_up = transpose ? 1.0 : -1.0, s_dn = transpose ? -1.0 : 1.0;
    int rc = slfmm_fork_near(S, x, y, transpose, st, false);
    if (!rc) rc = slfmm_launch_up(S, s_up, x, st);
    if (!rc) rc = fmm_launch_translate(transpose ? S->d_tptr : S->d_fptr, transpose ? S->d_toth : S->d_foth, transpose ? S->d_tval : S->d_fval,
                                       transpose ? S->d_tdense : S->d_fdense, S->nc, S->P, S->d_up, S->d_tr, st);
    if (!rc && !S->overlap) rc = slfmm_launch_down(S, s_dn, reinterpret_cast<dc*>(S->d_yfar), st, true);
    if (!rc) rc = slfmm_join_near(S, y, s_dn, st, false);
    return rc;
  }
  { int rc = slfmm_launch_near(S, x, y, transpose, st); if (rc) return rc; }
  // far field: forward T (e^-), D grouped by field, S (e^+); transpose S^T (e^+), D grouped by source, T^T (e^-)
  const double s_up = transpose ? 1.0 : -1.0, s_dn = transpose ? -1.0 : 1.0;
  { int rc = slfmm_launch_up(S, s_up, x, st); if (rc) return rc; }
  { int rc = fmm_launch_translate(transpose ? S->d_tptr : S->d_fptr, transpose ? S->d_toth : S->d_foth, transpose ? S->d_tval : S->d_fval,
                                  transpose ? S->d_tdense : S->d_fdense, S->nc, S->P, S->d_up, S->d_tr, st); if (rc) return rc; }
  { int rc = slfmm_launch_down(S, s_dn, y, st); if (rc) return rc; }
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

// =====================================================================================================================
// Multi-level fast multipole operator (math-bem/src/core/assembly/mlfmm.rs)
//   build_cluster_tree (:979-1038) with estimate_num_levels (:954-974), subdivide_level (:1056-1180) and
//   compute_near_far_lists (:1183-1223): HOST code, the same arithmetic in the same order as the reference (the tests compare the
//   tree with the restatement list by list and centre by centre, bit for bit).
//   build_mlfmm_system (:483-558) + MlfmmSystem::matvec (:128-460): the leaf level is the single-level machinery above (near
//   blocks WITHOUT free term, leaf T / D / S); the levels above it add
//     upward    M_C[p] = w_p sum_sons e^{-i k s_p . (c_son - c_C)} mean(M_son)          (build_t_matrices_level's non-leaf rows
//                                                                                         are constant over the son's points)
//     translate L_j[p] += h_0(k |c_i - c_j|) i k  M_i[p]   over the far pairs of every level
//     downward  L_son[q] += (1 / P_son) sum_p e^{+i k s_p . (c_son - c_C)} w_p L_C[p]     for every point q of the son
//   None of T, D, S is stored. Levels above the first one that has a far pair carry no information (their locals stay zero) and
//   are skipped. Supported: every level from that one down has a tabulated theta_points (otherwise gauss_legendre hands the
//   reference a longer rule than theta_points * phi_points and its length guards drop stages: MA_ERR_UNSUPPORTED here);
//   velocity-type boundary conditions, no evaluation elements. apply_transpose is unimplemented in the reference
//   (fmm_interface.rs:131-134) and MA_ERR_UNSUPPORTED here.
// =====================================================================================================================
struct HostCluster {
  double c[3] = {0, 0, 0}; double radius = 0.0;
  std::vector<int> elems, near, far, sons; int father = -1;
};
struct HostLevel {
  std::vector<HostCluster> cl;
  int terms = 4, theta = 4, phi = 8;
  double max_r = 0.0, avg_r = 0.0, min_r = 1.7976931348623157e308;
};
struct ma_cluster_tree { std::vector<HostLevel> levels; double k = 0.0; int n_elem = 0; };

namespace {
int tree_expansion_terms(double kr) {
  // ((kr + 6.0 * kr.ln().max(1.0)) as usize).clamp(4, 30): f64::max drops a NaN, `as usize` truncates and saturates
  const double ln = std::log(kr);
  const double m = (ln != ln || ln < 1.0) ? 1.0 : ln;
  const double v = kr + 6.0 * m;
  unsigned long long u;
  if (v != v || v <= 0.0) u = 0; else if (v >= 1.8446744073709552e19) u = ~0ull; else u = (unsigned long long)v;
  return (int)std::min<unsigned long long>(std::max<unsigned long long>(u, 4), 30);
}
int tree_estimate_num_levels(long long n, long long per_leaf, int min_levels, int max_levels) {
  if (n == 0) return min_levels;
  int levels = 1;
  while (n > per_leaf && levels < max_levels) { n /= 8; levels += 1; }
  return std::min(std::max(levels, min_levels), max_levels);
}
void tree_near_far(std::vector<HostCluster>& cl, double k) {
  const int n = (int)cl.size();
  for (int i = 0; i < n; ++i) {
    cl[(size_t)i].near.clear(); cl[(size_t)i].far.clear();
    for (int j = 0; j < n; ++j) {
      if (i == j) continue;
      const double d0 = cl[(size_t)i].c[0] - cl[(size_t)j].c[0], d1 = cl[(size_t)i].c[1] - cl[(size_t)j].c[1], d2 = cl[(size_t)i].c[2] - cl[(size_t)j].c[2];
      const double dist = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2);
      const double separation = dist / std::max(cl[(size_t)i].radius + cl[(size_t)j].radius, 1e-15);
      const double kr = k * dist;
      if (separation > 2.0 && kr > 2.0) cl[(size_t)i].far.push_back(j); else cl[(size_t)i].near.push_back(j);
    }
  }
}
void tree_subdivide(std::vector<HostLevel>& levels, const double* center, int parent_level, long long target, double k) {
  const std::vector<HostCluster> parents = levels[(size_t)parent_level].cl;      // the reference clones the parents first
  HostLevel child;
  double max_r = 0.0, min_r = 1.7976931348623157e308, sum_r = 0.0;
  static const double off[8][3] = {{-1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {-1, 1, 1}, {1, -1, -1}, {1, -1, 1}, {1, 1, -1}, {1, 1, 1}};
  for (size_t pi = 0; pi < parents.size(); ++pi) {
    const HostCluster& par = parents[pi];
    if ((long long)par.elems.size() <= target) {                                 // becomes a leaf: copied to the child level
      HostCluster leaf = par;
      leaf.father = (int)pi;
      max_r = std::max(max_r, leaf.radius); min_r = std::min(min_r, leaf.radius); sum_r += leaf.radius;
      levels[(size_t)parent_level].cl[pi].sons.push_back((int)child.cl.size());
      child.cl.push_back(leaf);
      continue;
    }
    const double half = par.radius / 2.0;
    for (int o = 0; o < 8; ++o) {
      HostCluster ch;
      for (int d = 0; d < 3; ++d) ch.c[d] = par.c[d] + off[o][d] * half * 0.5;
      for (int idx : par.elems) {
        const double dx = center[3 * (size_t)idx] - par.c[0], dy = center[3 * (size_t)idx + 1] - par.c[1], dz = center[3 * (size_t)idx + 2] - par.c[2];
        if (dx * off[o][0] >= 0.0 && dy * off[o][1] >= 0.0 && dz * off[o][2] >= 0.0) ch.elems.push_back(idx);
      }
      if (ch.elems.empty()) continue;
      ch.radius = half; ch.father = (int)pi;
      max_r = std::max(max_r, ch.radius); min_r = std::min(min_r, ch.radius); sum_r += ch.radius;
      levels[(size_t)parent_level].cl[pi].sons.push_back((int)child.cl.size());
      child.cl.push_back(ch);
    }
  }
  const size_t nc = child.cl.size();
  child.max_r = max_r; child.min_r = min_r; child.avg_r = nc > 0 ? sum_r / (double)nc : 0.0;
  child.terms = tree_expansion_terms(k * child.avg_r); child.theta = child.terms; child.phi = 2 * child.terms;
  levels.push_back(child);
  const int cur = (int)levels.size() - 1;
  bool more = false;
  for (const HostCluster& c : levels[(size_t)cur].cl) more = more || (long long)c.elems.size() > target;
  if (more && cur < 7) tree_subdivide(levels, center, cur, target, k);
}
}  // namespace

int cluster_tree_build(int n_elem, const double* center, long long target, double k, ma_cluster_tree** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(n_elem >= 1 && center && target >= 1, MA_ERR_INVALID, "bad argument");
  ma_cluster_tree* T = new (std::nothrow) ma_cluster_tree(); MA_REQUIRE(T, MA_ERR_NOMEM, "host allocation failed");
  T->k = k; T->n_elem = n_elem;
  const int num_levels = tree_estimate_num_levels(n_elem, target, 1, 8);
  double lo[3] = {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308}, hi[3] = {-1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308};
  for (int e = 0; e < n_elem; ++e) for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], center[3 * (size_t)e + d]); hi[d] = std::max(hi[d], center[3 * (size_t)e + d]); }
  HostLevel l0; HostCluster root;
  for (int d = 0; d < 3; ++d) root.c[d] = (lo[d] + hi[d]) / 2.0;
  const double e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
  root.radius = std::sqrt(e0 * e0 + e1 * e1 + e2 * e2) / 2.0;
  root.elems.resize((size_t)n_elem); for (int e = 0; e < n_elem; ++e) root.elems[(size_t)e] = e;
  l0.terms = tree_expansion_terms(k * root.radius); l0.theta = l0.terms; l0.phi = 2 * l0.terms;
  l0.max_r = l0.avg_r = l0.min_r = root.radius;
  l0.cl.push_back(root);
  T->levels.push_back(l0);
  if (num_levels > 1) tree_subdivide(T->levels, center, 0, target, k);
  for (HostLevel& lv : T->levels) tree_near_far(lv.cl, k);
  *out = T;
  return MA_OK;
}
void cluster_tree_destroy(ma_cluster_tree* T) { delete T; }
int cluster_tree_num_levels(const ma_cluster_tree* T) { return (int)T->levels.size(); }
int cluster_tree_level_info(const ma_cluster_tree* T, int level, int* nclusters, int* terms, int* theta, int* phi, long long* n_elem_listed, long long* n_near, long long* n_far,
                            long long* n_sons) {
  MA_REQUIRE(T && level >= 0 && level < (int)T->levels.size(), MA_ERR_INVALID, "level %d outside the tree", level);
  const HostLevel& lv = T->levels[(size_t)level];
  long long ne = 0, nn = 0, nf = 0, ns = 0;
  for (const HostCluster& c : lv.cl) { ne += (long long)c.elems.size(); nn += (long long)c.near.size(); nf += (long long)c.far.size(); ns += (long long)c.sons.size(); }
  if (nclusters) *nclusters = (int)lv.cl.size();
  if (terms) *terms = lv.terms;
  if (theta) *theta = lv.theta;
  if (phi) *phi = lv.phi;
  if (n_elem_listed) *n_elem_listed = ne;
  if (n_near) *n_near = nn;
  if (n_far) *n_far = nf;
  if (n_sons) *n_sons = ns;
  return MA_OK;
}
int cluster_tree_level_get(const ma_cluster_tree* T, int level, double* center, double* radius, int* elem_ptr, int* elem_idx, int* near_ptr, int* near_idx, int* far_ptr, int* far_idx,
                           int* son_ptr, int* son_idx, int* father) {
  MA_REQUIRE(T && level >= 0 && level < (int)T->levels.size(), MA_ERR_INVALID, "level %d outside the tree", level);
  const HostLevel& lv = T->levels[(size_t)level];
  int pe = 0, pn = 0, pf = 0, ps = 0;
  for (size_t c = 0; c < lv.cl.size(); ++c) {
    const HostCluster& q = lv.cl[c];
    if (center) for (int d = 0; d < 3; ++d) center[3 * c + (size_t)d] = q.c[d];
    if (radius) radius[c] = q.radius;
    if (father) father[c] = q.father;
    if (elem_ptr) elem_ptr[c] = pe;
    if (near_ptr) near_ptr[c] = pn;
    if (far_ptr) far_ptr[c] = pf;
    if (son_ptr) son_ptr[c] = ps;
    for (int v : q.elems) { if (elem_idx) elem_idx[pe] = v; ++pe; }
    for (int v : q.near) { if (near_idx) near_idx[pn] = v; ++pn; }
    for (int v : q.far) { if (far_idx) far_idx[pf] = v; ++pf; }
    for (int v : q.sons) { if (son_idx) son_idx[ps] = v; ++ps; }
  }
  const size_t nc = lv.cl.size();
  if (elem_ptr) elem_ptr[nc] = pe;
  if (near_ptr) near_ptr[nc] = pn;
  if (far_ptr) far_ptr[nc] = pf;
  if (son_ptr) son_ptr[nc] = ps;
  return MA_OK;
}

struct MlLevel {                                      // one level at or above the leaves that takes part in the far field
  int nc = 0, P = 0;
  double* d_cc = nullptr; double* d_sc = nullptr; double* d_sw = nullptr;
  int* d_fptr = nullptr; int* d_foth = nullptr; c64* d_fval = nullptr;    // far pairs grouped by field cluster
  c64* d_fdense = nullptr;                                                 // the same as a dense nc x nc matrix, when the lists are nearly full
  int* d_sptr = nullptr; int* d_sidx = nullptr;                            // sons (indices into the level below)
  c64* d_M = nullptr; c64* d_L = nullptr;                                  // multipoles / locals, nc x P (the leaf level uses the single-level buffers)
};
struct ma_mlfmm {
  int device = 0; ma_slfmm* leaf = nullptr; long long n = 0; double k = 0.0;
  bool far_field = false;
  std::vector<MlLevel> up;                            // levels l0 .. leaf-1, top first
};

namespace {
// M_C[p] = w_p sum_sons e^{-i k s_p . (c_son - c_C)} * mean over the son's points of M_son
__global__ __launch_bounds__(256) void mlfmm_m2m_kernel(const int* __restrict__ sptr, const int* __restrict__ sidx, const double* __restrict__ cc, const double* __restrict__ cc_child,
                                                        const double* __restrict__ sc, const double* __restrict__ sw, int P, int Pc, double k,
                                                        const dc* __restrict__ Mc, dc* __restrict__ M) {
  __shared__ double red[2][256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const double Cx = cc[3 * c], Cy = cc[3 * c + 1], Cz = cc[3 * c + 2];
  double ar[4] = {0, 0, 0, 0}, ai[4] = {0, 0, 0, 0};                     // this thread's points p = tid + 256 u (P <= 800)
  for (int q = sptr[c]; q < sptr[c + 1]; ++q) {
    const int s = sidx[q];
    double sr = 0.0, si = 0.0;
    for (int t = tid; t < Pc; t += 256) { const dc v = Mc[(long long)s * Pc + t]; sr += v.re; si += v.im; }
    red[0][tid] = sr; red[1][tid] = si;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) { if (tid < h) { red[0][tid] += red[0][tid + h]; red[1][tid] += red[1][tid + h]; } __syncthreads(); }
    const double mr = red[0][0] / (double)Pc, mi = red[1][0] / (double)Pc;
    __syncthreads();
    const double dx = cc_child[3 * s] - Cx, dy = cc_child[3 * s + 1] - Cy, dz = cc_child[3 * s + 2] - Cz;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = tid + 256 * u;
      if (p < P) {
        const double sd = sc[3 * p] * dx + sc[3 * p + 1] * dy + sc[3 * p + 2] * dz;
        double sn, cs; sincos_bounded(k * sd, sn, cs);   // |k s.(c_son - c_C)| <= k x the father's radius
        ar[u] += cs * mr + sn * mi; ai[u] += cs * mi - sn * mr;          // e^{-i t} (mr + i mi)
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int p = tid + 256 * u; if (p < P) M[(long long)c * P + p] = dc_make(sw[p] * ar[u], sw[p] * ai[u]); }
}
// L_son[q] += (1 / Pc) sum_p e^{+i k s_p . (c_son - c_C)} w_p L_C[p]
__global__ __launch_bounds__(256) void mlfmm_l2l_kernel(const int* __restrict__ sptr, const int* __restrict__ sidx, const double* __restrict__ cc, const double* __restrict__ cc_child,
                                                        const double* __restrict__ sc, const double* __restrict__ sw, int P, int Pc, double k,
                                                        const dc* __restrict__ L, dc* __restrict__ Lc) {
  __shared__ double red[2][256];
  const int c = blockIdx.x, tid = threadIdx.x;
  const double Cx = cc[3 * c], Cy = cc[3 * c + 1], Cz = cc[3 * c + 2];
  for (int q = sptr[c]; q < sptr[c + 1]; ++q) {
    const int s = sidx[q];
    const double dx = cc_child[3 * s] - Cx, dy = cc_child[3 * s + 1] - Cy, dz = cc_child[3 * s + 2] - Cz;
    double sr = 0.0, si = 0.0;
    for (int p = tid; p < P; p += 256) {
      const double sd = sc[3 * p] * dx + sc[3 * p + 1] * dy + sc[3 * p + 2] * dz;
      double sn, cs; sincos_bounded(k * sd, sn, cs);   // |k s.(c_son - c_C)| <= k x the father's radius
      const dc l = L[(long long)c * P + p]; const double w = sw[p];
      sr += w * (cs * l.re - sn * l.im); si += w * (cs * l.im + sn * l.re);
    }
    red[0][tid] = sr; red[1][tid] = si;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) { if (tid < h) { red[0][tid] += red[0][tid + h]; red[1][tid] += red[1][tid + h]; } __syncthreads(); }
    const double ar = red[0][0] / (double)Pc, ai = red[1][0] / (double)Pc;
    __syncthreads();
    for (int t = tid; t < Pc; t += 256) { dc* o = Lc + (long long)s * Pc + t; o->re += ar; o->im += ai; }   // a son has one father: no race
  }
}
void ml_level_free(MlLevel& L) {
  void* p[] = {L.d_cc, L.d_sc, L.d_sw, L.d_fptr, L.d_foth, L.d_fval, L.d_fdense, L.d_sptr, L.d_sidx, L.d_M, L.d_L};
  for (void* q : p) if (q) (void)hipFree(q);
}
bool theta_tabulated(int t) { return t >= 1 && t <= 20 && mat_gl_index[t][1] == t; }
void sphere_rule(int n_theta, int n_phi, std::vector<double>& sc, std::vector<double>& sw) {
  const int off = mat_gl_index[n_theta][0];
  const double pi = 3.14159265358979323846, dphi = 2.0 * pi / (double)n_phi;
  sc.resize((size_t)n_theta * n_phi * 3); sw.resize((size_t)n_theta * n_phi);
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
}  // namespace

void mlfmm_destroy(ma_mlfmm* S) {
  if (!S) return;
  (void)hipSetDevice(S->device);
  for (MlLevel& L : S->up) ml_level_free(L);
  if (S->leaf) slfmm_destroy(S->leaf);
  delete S;
}

int mlfmm_create(ma_bem_plan* plan, const ma_cluster_tree* T, const ma_physics_t* physics, ma_mlfmm** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(plan && T && physics, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(T->n_elem == plan->np, MA_ERR_DIM, "the tree was built for %d elements, the plan holds %d", T->n_elem, plan->np);
  MA_REQUIRE(!T->levels.empty(), MA_ERR_INVALID, "empty tree");
  const int nl = (int)T->levels.size(), leaf = nl - 1;
  const double k = physics->wave_number;
  // far field: only with more than one level (mlfmm.rs:186); levels above the first far pair carry nothing
  int l0 = -1;
  if (nl > 1) for (int l = 0; l < nl && l0 < 0; ++l) for (const HostCluster& c : T->levels[(size_t)l].cl) if (!c.far.empty()) { l0 = l; break; }
  if (l0 >= 0)
    for (int l = l0; l < nl; ++l)
      MA_REQUIRE(theta_tabulated(T->levels[(size_t)l].theta), MA_ERR_UNSUPPORTED,
                 "level %d has theta_points = %d, not a tabulated Gauss-Legendre order: the reference's sphere rule would be longer than theta_points * phi_points and its "
                 "matvec would skip stages (gauss.rs:27-60, mlfmm.rs:229-447)", l, T->levels[(size_t)l].theta);
  ma_mlfmm* S = new (std::nothrow) ma_mlfmm(); MA_REQUIRE(S, MA_ERR_NOMEM, "host allocation failed");
  S->device = plan->device; S->n = plan->nd; S->k = k; S->far_field = l0 >= 0;
  auto fail = [&](int code) { mlfmm_destroy(S); return code; };
  // ---- leaf level through the single-level machinery: near lists, leaf far pairs, leaf sphere rule
  {
    const HostLevel& lv = T->levels[(size_t)leaf];
    const int nc = (int)lv.cl.size();
    std::vector<double> cc((size_t)nc * 3);
    std::vector<int> ep(1, 0), ei, np_(1, 0), ni, fp(1, 0), fi;
    std::vector<char> seen((size_t)plan->np, 0); bool overlap = false;
    for (int c = 0; c < nc; ++c) {
      const HostCluster& q = lv.cl[(size_t)c];
      for (int d = 0; d < 3; ++d) cc[(size_t)3 * c + d] = q.c[d];
      for (int e : q.elems) { ei.push_back(e); if (seen[(size_t)e]) overlap = true; seen[(size_t)e] = 1; }
      ni.insert(ni.end(), q.near.begin(), q.near.end());
      if (S->far_field) fi.insert(fi.end(), q.far.begin(), q.far.end());
      ep.push_back((int)ei.size()); np_.push_back((int)ni.size()); fp.push_back((int)fi.size());
    }
    const ma_clusters_t cl{nc, cc.data(), ep.data(), ei.data(), np_.data(), ni.data(), fp.data(), fi.data()};
    const int theta = S->far_field ? lv.theta : 4, phi = S->far_field ? lv.phi : 8;
    int rc = slfmm_create_ex(plan, &cl, physics, theta, phi, lv.terms, false, overlap, &S->leaf);
    if (rc) return fail(rc);
  }
  MA_HIP(hipSetDevice(plan->device));
  // ---- the levels above the leaves that take part
  if (S->far_field)
    for (int l = l0; l < leaf; ++l) {
      const HostLevel& lv = T->levels[(size_t)l];
      MlLevel L; L.nc = (int)lv.cl.size(); L.P = lv.theta * lv.phi;
      if (L.P > 1024) { set_error("level %d: %d sphere points", l, L.P); return fail(MA_ERR_UNSUPPORTED); }
      std::vector<double> cc((size_t)L.nc * 3), sc, sw;
      sphere_rule(lv.theta, lv.phi, sc, sw);
      std::vector<int> sptr(1, 0), sidx;
      const int order = std::max(lv.terms, 2);
      // one spherical Hankel evaluation per far pair, on host threads, each into its own slot (skipped pairs keep the flag)
      std::vector<std::vector<c64>> dv((size_t)L.nc); std::vector<std::vector<char>> dskip((size_t)L.nc);
      host_parallel_for(L.nc, 16, [&](long long c0, long long c1) {
        std::vector<double> hr, hi;
        for (long long i = c0; i < c1; ++i) {
          const HostCluster& q = lv.cl[(size_t)i];
          dv[(size_t)i].resize(q.far.size()); dskip[(size_t)i].assign(q.far.size(), 0);
          for (size_t t = 0; t < q.far.size(); ++t) {
            const HostCluster& o = lv.cl[(size_t)q.far[t]];
            const double dx = q.c[0] - o.c[0], dy = q.c[1] - o.c[1], dz = q.c[2] - o.c[2];
            const double r = std::sqrt(dx * dx + dy * dy + dz * dz);
            if (r < 1e-15) { dskip[(size_t)i][t] = 1; continue; }          // mlfmm.rs:826-828
            spherical_hankel_first_kind(order, k * r, 1.0, hr, hi);
            dv[(size_t)i][t] = c64{-hi[0] * k, hr[0] * k};
          }
        }
      });
      std::vector<int> fptr((size_t)L.nc + 1, 0), foth; std::vector<c64> fval;       // by field, source ascending within a row
      for (int i = 0; i < L.nc; ++i) {
        const HostCluster& q = lv.cl[(size_t)i];
        for (int d = 0; d < 3; ++d) cc[(size_t)3 * i + d] = q.c[d];
        for (size_t t = 0; t < q.far.size(); ++t) if (!dskip[(size_t)i][t]) fptr[(size_t)q.far[t] + 1] += 1;
        sidx.insert(sidx.end(), q.sons.begin(), q.sons.end()); sptr.push_back((int)sidx.size());
      }
      for (int c = 0; c < L.nc; ++c) fptr[(size_t)c + 1] += fptr[(size_t)c];
      foth.resize((size_t)fptr[(size_t)L.nc]); fval.resize(foth.size());
      { std::vector<int> cur(fptr.begin(), fptr.end() - 1);
        for (int i = 0; i < L.nc; ++i) {
          const HostCluster& q = lv.cl[(size_t)i];
          for (size_t t = 0; t < q.far.size(); ++t)
            if (!dskip[(size_t)i][t]) { const int pos = cur[(size_t)q.far[t]]++; foth[(size_t)pos] = i; fval[(size_t)pos] = dv[(size_t)i][t]; }   // locals[field j] += d multipoles[source i]
        } }
      int rc = upload(&L.d_cc, cc);
      if (!rc) rc = upload(&L.d_sc, sc);
      if (!rc) rc = upload(&L.d_sw, sw);
      if (!rc) rc = upload(&L.d_fptr, fptr);
      if (!rc) rc = upload(&L.d_foth, foth);
      if (!rc) rc = upload(&L.d_fval, fval);
      if (!rc) rc = fmm_dense_from_lists(fptr, foth, fval, L.nc, &L.d_fdense);
      if (!rc) rc = upload(&L.d_sptr, sptr);
      if (!rc) rc = upload(&L.d_sidx, sidx);
      if (!rc && hipMalloc(&L.d_M, sizeof(c64) * (size_t)L.nc * (size_t)L.P) != hipSuccess) { set_error("MLFMM level buffers"); rc = MA_ERR_NOMEM; }
      if (!rc && hipMalloc(&L.d_L, sizeof(c64) * (size_t)L.nc * (size_t)L.P) != hipSuccess) { set_error("MLFMM level buffers"); rc = MA_ERR_NOMEM; }
      S->up.push_back(L);
      if (rc) return fail(rc);
    }
  *out = S;
  return MA_OK;
}

int mlfmm_apply(ma_mlfmm* S, const c64* d_x, c64* d_y, hipStream_t st) {
  MA_HIP(hipSetDevice(S->device));
  ma_slfmm* F = S->leaf;
  if (!(F->covers_all && F->d_part)) MA_HIP(hipMemsetAsync(d_y, 0, sizeof(c64) * (size_t)S->n, st));
  const dc* x = reinterpret_cast<const dc*>(d_x); dc* y = reinterpret_cast<dc*>(d_y);
  // rounds 4-5: with a far field, the whole near field runs on the leaf operator's second stream beside the far chain
  const bool two = S->far_field && slfmm_overlap_ready(F);
  const bool whole = two && !F->overlap;                    // the per-cluster pass too (overlapping leaves accumulate with atomics: the round-4 order)
  if (two) { int rc = slfmm_fork_near(F, x, y, 0, st, whole); if (rc) return rc; }
  else { int rc = slfmm_launch_near(F, x, y, 0, st); if (rc) return rc; }
  if (!S->far_field) return MA_OK;
  // upward pass: leaf multipoles, then level by level to the top level that has far pairs
  { int rc = slfmm_launch_up(F, -1.0, x, st); if (rc) return rc; }
  const int nu = (int)S->up.size();
  const int nt = fmm_levels_nt(F->P);
  // translation at every level: the levels whose dense D takes the small-tile kernel travel in one launch per stream
  auto add_or_launch = [&](FmmLevels& V, const int* fptr, const int* foth, const c64* fval, const c64* dense, int nc, int P, const c64* up, c64* tr, hipStream_t s_) -> int {
    if (V.nl < 8 && fmm_translate_batchable(dense, nc, P)) {
      const int q = V.nl++;
      V.rows[q] = (nc + 15) / 16; V.nc[q] = nc; V.P[q] = P;
      V.DT[q] = reinterpret_cast<const dc*>(dense); V.up[q] = reinterpret_cast<const dc*>(up); V.tr[q] = reinterpret_cast<dc*>(tr);
      V.first[q + 1] = V.first[q] + V.rows[q] * ((P + 16 * nt - 1) / (16 * nt));
      return MA_OK;
    }
    return fmm_launch_translate(fptr, foth, fval, dense, nc, P, up, tr, s_);
  };
  // round 5: the leaf level's translation (the largest; it needs the leaf multipoles only) on a third stream beside the upward chain of
  // the levels above, their translations and the downward chain as far as the leaves' fathers; the last l2l step, which ADDS the
  // fathers' locals to what the leaf translation WROTE, waits for it
  const bool three = two && nu > 0 && F->st3;
  if (three) {
    MA_HIP(hipEventRecord(F->ev_up, st));
    MA_HIP(hipStreamWaitEvent(F->st3, F->ev_up, 0));
    FmmLevels VL; VL.nl = 0; VL.first[0] = 0;
    { int rc = add_or_launch(VL, F->d_fptr, F->d_foth, F->d_fval, F->d_fdense, F->nc, F->P, F->d_up, F->d_tr, F->st3); if (rc) return rc; }
    { int rc = fmm_launch_translate_levels(VL, nt, F->st3); if (rc) return rc; }
    MA_HIP(hipEventRecord(F->ev_leaf, F->st3));
  }
  for (int l = nu - 1; l >= 0; --l) {
    MlLevel& L = S->up[(size_t)l];
    const bool below_is_leaf = l == nu - 1;
    const double* ccc = below_is_leaf ? F->d_cc : S->up[(size_t)l + 1].d_cc;
    const int Pc = below_is_leaf ? F->P : S->up[(size_t)l + 1].P;
    const dc* Mc = reinterpret_cast<const dc*>(below_is_leaf ? F->d_up : S->up[(size_t)l + 1].d_M);
    hipLaunchKernelGGL(mlfmm_m2m_kernel, dim3(L.nc), dim3(256), 0, st, L.d_sptr, L.d_sidx, L.d_cc, ccc, L.d_sc, L.d_sw, L.P, Pc, S->k, Mc, reinterpret_cast<dc*>(L.d_M));
    MA_HIP(hipGetLastError());
  }
  FmmLevels V; V.nl = 0; V.first[0] = 0;
  for (int l = 0; l < nu; ++l) {
    MlLevel& L = S->up[(size_t)l];
    { int rc = add_or_launch(V, L.d_fptr, L.d_foth, L.d_fval, L.d_fdense, L.nc, L.P, L.d_M, L.d_L, st); if (rc) return rc; }
  }
  if (!three) { int rc = add_or_launch(V, F->d_fptr, F->d_foth, F->d_fval, F->d_fdense, F->nc, F->P, F->d_up, F->d_tr, st); if (rc) return rc; }
  { int rc = fmm_launch_translate_levels(V, nt, st); if (rc) return rc; }
  // downward pass
  for (int l = 0; l < nu; ++l) {
    MlLevel& L = S->up[(size_t)l];
    const bool below_is_leaf = l == nu - 1;
    const double* ccc = below_is_leaf ? F->d_cc : S->up[(size_t)l + 1].d_cc;
    const int Pc = below_is_leaf ? F->P : S->up[(size_t)l + 1].P;
    dc* Lc = reinterpret_cast<dc*>(below_is_leaf ? F->d_tr : S->up[(size_t)l + 1].d_L);
    if (below_is_leaf && three) MA_HIP(hipStreamWaitEvent(st, F->ev_leaf, 0));
    hipLaunchKernelGGL(mlfmm_l2l_kernel, dim3(L.nc), dim3(256), 0, st, L.d_sptr, L.d_sidx, L.d_cc, ccc, L.d_sc, L.d_sw, L.P, Pc, S->k, reinterpret_cast<const dc*>(L.d_L), Lc);
    MA_HIP(hipGetLastError());
  }
  if (two) {
    // whole: the near field is in y, the downward pass adds the far field's share. Otherwise (overlapping leaves) both accumulate into
    // the cleared y with atomics, near sums first, as before
    return slfmm_join_near(F, y, 1.0, st, whole);
  }
  { int rc = slfmm_launch_down(F, 1.0, y, st); if (rc) return rc; }
  return MA_OK;
}
