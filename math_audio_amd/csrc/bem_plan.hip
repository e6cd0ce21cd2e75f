// bem_plan.hip — host side of the TBEM assembly: mesh validation, HBM layout, the per-mesh
// near-pair plan, and the C-ABI entry points of include/mathaudio_hip.h for this row.
#include "bem_kernels.hpp"
#include <algorithm>
#include "ma_tables.h"
#include <vector>
#include <cmath>
#include <cstdlib>
#include <new>

namespace ma {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

int use_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no HIP device visible (%s); libmathaudio_hip has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    return MA_ERR_NO_DEVICE;
  }
  MA_REQUIRE(device >= 0 && device < n, MA_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
  MA_HIP(hipSetDevice(device));
  hipDeviceProp_t p;
  MA_HIP(hipGetDeviceProperties(&p, device));
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; this library ships gfx950 (MI355X) code objects only", device, p.gcnArchName);
    return MA_ERR_NO_DEVICE;
  }
  return MA_OK;
}

}  // namespace ma

using namespace ma;

namespace {

// Host geometry exactly as the reference's per-point code evaluates it for a flat Tri3
// (regular.rs:236-257): dx_ds = p1 - p0, dx_dt = p2 - p0, normal = dx_ds x dx_dt,
// jacobian = sqrt(n.n), el_norm = n / jacobian (zero if jacobian <= 1e-15). No contraction.
#pragma clang fp contract(off)
void panel_frame(const double* p0, const double* p1, const double* p2, double* e1, double* e2, double* ny, double* jac) {
  for (int d = 0; d < 3; ++d) { e1[d] = p1[d] - p0[d]; e2[d] = p2[d] - p0[d]; }
  double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
  double j = std::sqrt(((n[0] * n[0]) + n[1] * n[1]) + n[2] * n[2]);
  *jac = j;
  if (j > 1e-15) { ny[0] = n[0] / j; ny[1] = n[1] / j; ny[2] = n[2] / j; }
  else { ny[0] = ny[1] = ny[2] = 0.0; }
}

double avg_center_radius(const ma_mesh_t* m) {
  int nc = m->n_elem < 100 ? m->n_elem : 100;
  double s = 0.0;
  for (int e = 0; e < nc; ++e) {
    const double* c = m->center + 3 * e;
    s += std::sqrt(((c[0] * c[0]) + c[1] * c[1]) + c[2] * c[2]);
  }
  if (nc > 0) s /= (double)nc;
  return s;
}
#pragma clang fp contract(fast)

int validate_mesh(const ma_mesh_t* m) {
  MA_REQUIRE(m, MA_ERR_INVALID, "mesh is NULL");
  MA_REQUIRE(m->n_elem > 0 && m->n_nodes > 0, MA_ERR_INVALID, "empty mesh (n_elem=%d, n_nodes=%d)", m->n_elem, m->n_nodes);
  MA_REQUIRE(m->nodes && m->conn && m->center && m->normal && m->area && m->dof && m->bc_type, MA_ERR_INVALID,
             "mesh has a NULL required array");
  for (int e = 0; e < m->n_elem; ++e) {
    const int32_t* c = m->conn + 4 * e;
    const int nn = c[3] < 0 ? 3 : 4;                    // Tri3 rows carry -1 in the fourth slot
    for (int a = 0; a < nn; ++a)
      MA_REQUIRE(c[a] >= 0 && c[a] < m->n_nodes, MA_ERR_INVALID, "element %d references node %d (n_nodes=%d)", e, c[a], m->n_nodes);
    if (m->bc_values) {
      int len = m->bc_len ? m->bc_len[e] : 1;
      MA_REQUIRE(len >= 1 && len <= 4, MA_ERR_INVALID, "element %d: bc_len %d outside 1..4", e, len);
    }
  }
  return MA_OK;
}

}  // namespace

extern "C" {

const char* ma_last_error_string(void) { return ma::g_err.c_str(); }
const char* ma_version(void) { return "mathaudio_hip 0.1 (gfx950)"; }

int ma_device_count(int* count) {
  MA_REQUIRE(count, MA_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *count = (e == hipSuccess) ? n : 0;
  return MA_OK;
}

int ma_bem_plan_create(const ma_mesh_t* m, int device, ma_bem_plan_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int rc = validate_mesh(m);
  if (rc) return rc;
  rc = use_device(device);
  if (rc) return rc;

  // panels = non-evaluation elements, in element order (tbem.rs:126-155 skips evaluation elements)
  std::vector<int> elems;
  for (int e = 0; e < m->n_elem; ++e)
    if (!(m->is_eval && m->is_eval[e])) elems.push_back(e);
  const int np = (int)elems.size();
  MA_REQUIRE(np > 0, MA_ERR_INVALID, "mesh has no boundary (non-evaluation) elements");
  {
    std::vector<unsigned char> seen((size_t)np, 0);
    for (int p = 0; p < np; ++p) {
      int d = m->dof[elems[p]];
      MA_REQUIRE(d >= 0 && d < np && !seen[d], MA_ERR_UNSUPPORTED,
                 "dof_addresses must enumerate 0..%d once each (element %d has dof %d)", np - 1, elems[p], d);
      seen[d] = 1;
    }
  }

  // host SoA: 25 double arrays of np + dof(int) + bc(uchar)
  const int ND = 28;                                     // 25 Tri3 arrays + the fourth vertex of Quad4 panels
  const size_t stride = ((size_t)np + 63) & ~(size_t)63;   // keep every array 512-B aligned
  std::vector<double> h((size_t)ND * stride, 0.0);
  std::vector<int> hdof(stride, 0);
  std::vector<unsigned char> hbc(stride, 0), hpt(stride, 3);
  std::vector<int> hquads;
  auto arr = [&](int a) { return h.data() + (size_t)a * stride; };
  for (int p = 0; p < np; ++p) {
    const int e = elems[p];
    const int32_t* c = m->conn + 4 * e;
    const double* q0 = m->nodes + 3 * c[0];
    const double* q1 = m->nodes + 3 * c[1];
    const double* q2 = m->nodes + 3 * c[2];
    double e1[3], e2[3], ny[3], jac;
    panel_frame(q0, q1, q2, e1, e2, ny, &jac);
    for (int d = 0; d < 3; ++d) {
      arr(0 + d)[p] = q0[d]; arr(3 + d)[p] = q1[d]; arr(6 + d)[p] = q2[d];
      arr(9 + d)[p] = e1[d]; arr(12 + d)[p] = e2[d]; arr(15 + d)[p] = ny[d];
      arr(19 + d)[p] = m->center[3 * e + d]; arr(22 + d)[p] = m->normal[3 * e + d];
    }
    arr(18)[p] = jac;
    if (c[3] >= 0) {
      const double* q3 = m->nodes + 3 * c[3];
      for (int d = 0; d < 3; ++d) arr(25 + d)[p] = q3[d];
      hpt[p] = 4; hquads.push_back(p);
    }
    hdof[p] = m->dof[e];
    hbc[p] = m->bc_type[e] > 1 ? 2 : m->bc_type[e];
  }
  std::vector<double> harea(stride, 0.0);
  for (int p = 0; p < np; ++p) harea[p] = m->area[elems[p]];

  ma_bem_plan* P = new (std::nothrow) ma_bem_plan();
  MA_REQUIRE(P, MA_ERR_NOMEM, "host allocation failed");
  P->device = device; P->np = np; P->nd = np;
  P->avg_radius = avg_center_radius(m);
  {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = 0; i < m->n_nodes; ++i)
      for (int d = 0; d < 3; ++d) { lo[d] = std::min(lo[d], m->nodes[3 * i + d]); hi[d] = std::max(hi[d], m->nodes[3 * i + d]); }
    P->diameter = m->n_nodes > 0 ? std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2])) : 0.0;
  }

  const size_t bytes_d = (size_t)(ND + 1) * stride * sizeof(double);
  const size_t nq_pad = (hquads.size() + 63) & ~(size_t)63;
  const size_t bytes = bytes_d + stride * sizeof(int) + stride + stride + (nq_pad + 64) * sizeof(int);
  hipError_t he = hipMalloc(&P->pool, bytes);
  if (he != hipSuccess) { set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(he)); delete P; return MA_ERR_NOMEM; }
  char* base = (char*)P->pool;
  auto fail = [&](int code) { (void)hipFree(P->pool); if (P->d_pairs) (void)hipFree(P->d_pairs); delete P; return code; };
#define MA_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); return fail(MA_ERR_HIP); } } while (0)
  MA_TRY(hipMemcpy(base, h.data(), (size_t)ND * stride * sizeof(double), hipMemcpyHostToDevice));
  MA_TRY(hipMemcpy(base + (size_t)ND * stride * sizeof(double), harea.data(), stride * sizeof(double), hipMemcpyHostToDevice));
  MA_TRY(hipMemcpy(base + bytes_d, hdof.data(), stride * sizeof(int), hipMemcpyHostToDevice));
  MA_TRY(hipMemcpy(base + bytes_d + stride * sizeof(int), hbc.data(), stride, hipMemcpyHostToDevice));
  MA_TRY(hipMemcpy(base + bytes_d + stride * sizeof(int) + stride, hpt.data(), stride, hipMemcpyHostToDevice));
  if (!hquads.empty()) MA_TRY(hipMemcpy(base + bytes_d + stride * sizeof(int) + 2 * stride, hquads.data(), hquads.size() * sizeof(int), hipMemcpyHostToDevice));
  BemGeom& g = P->geom;
  g.np = np; g.nd = np;
  const double* dd = (const double*)base;
  for (int d = 0; d < 3; ++d) {
    g.p0[d] = dd + (size_t)(0 + d) * stride; g.p1[d] = dd + (size_t)(3 + d) * stride; g.p2[d] = dd + (size_t)(6 + d) * stride;
    g.e1[d] = dd + (size_t)(9 + d) * stride; g.e2[d] = dd + (size_t)(12 + d) * stride; g.ny[d] = dd + (size_t)(15 + d) * stride;
    g.c[d] = dd + (size_t)(19 + d) * stride; g.nx[d] = dd + (size_t)(22 + d) * stride;
  }
  g.jac = dd + (size_t)18 * stride;
  g.area = dd + (size_t)ND * stride;
  g.dof = (const int*)(base + bytes_d);
  g.bc_type = (const unsigned char*)(base + bytes_d + stride * sizeof(int));
  for (int d = 0; d < 3; ++d) g.p3[d] = dd + (size_t)(25 + d) * stride;
  g.ptype = (const unsigned char*)(base + bytes_d + stride * sizeof(int) + stride);
  g.quad_ids = (const int*)(base + bytes_d + stride * sizeof(int) + 2 * stride);
  g.nquad = (int)hquads.size();
  g.all_velocity = 1;
  for (int p = 0; p < np; ++p) if (hbc[p] != 0) g.all_velocity = 0;

  // constant tables (13-point rule scaled by 0.5 as triangle_quadrature does, gauss.rs:70)
  double t13[13][3];
  for (int q = 0; q < 13; ++q) { t13[q][0] = mat_tri13[q][0]; t13[q][1] = mat_tri13[q][1]; t13[q][2] = mat_tri13[q][2] * 0.5; }
  rc = bem_upload_tables(t13, mat_gl_x, mat_gl_w, mat_gl_index);
  if (rc) return fail(rc);

  // near-pair list: count, scan on host, fill (one-off per mesh; creation is synchronous)
  int* d_counts = nullptr; long long* d_offsets = nullptr;
  MA_TRY(hipMalloc(&d_counts, sizeof(int) * (size_t)np));
  MA_TRY(hipMalloc(&d_offsets, sizeof(long long) * (size_t)(np + 1)));
  rc = bem_launch_near_list(g, 0, d_counts, nullptr, nullptr, nullptr);
  if (rc) { (void)hipFree(d_counts); (void)hipFree(d_offsets); return fail(rc); }
  std::vector<int> hc((size_t)np);
  MA_TRY(hipMemcpy(hc.data(), d_counts, sizeof(int) * (size_t)np, hipMemcpyDeviceToHost));
  std::vector<long long> ho((size_t)np + 1);
  long long tot = 0;
  for (int i = 0; i < np; ++i) { ho[i] = tot; tot += hc[i]; }
  ho[np] = tot;
  P->npairs = tot;
  MA_TRY(hipMemcpy(d_offsets, ho.data(), sizeof(long long) * (size_t)(np + 1), hipMemcpyHostToDevice));
  if (tot > 0) {
    MA_TRY(hipMalloc(&P->d_pairs, sizeof(int2) * (size_t)tot));
    rc = bem_launch_near_list(g, 1, d_counts, d_offsets, P->d_pairs, nullptr);
    if (rc) { (void)hipFree(d_counts); (void)hipFree(d_offsets); return fail(rc); }
  }
  MA_TRY(hipDeviceSynchronize());
  (void)hipFree(d_counts);
  P->d_pair_off = d_offsets;

  // boundary values (get_bc_type_and_value, tbem.rs:234-244: transfer admittances carry a single zero)
  if (m->bc_values) {
    std::vector<c64> hv((size_t)4 * np); std::vector<int> hl((size_t)np, 1); std::vector<unsigned char> hz((size_t)np, 0);
    bool any = false;
    for (int p = 0; p < np; ++p) {
      const int e = elems[p];
      const int len = m->bc_len ? m->bc_len[e] : 1;
      hl[p] = len;
      for (int a = 0; a < 4; ++a) { hv[4 * (size_t)p + a].re = 0.0; hv[4 * (size_t)p + a].im = 0.0; }
      if (hbc[p] <= 1)
        for (int a = 0; a < len; ++a) {
          const ma_c64 v = m->bc_values[4 * e + a];
          hv[4 * (size_t)p + a].re = v.re; hv[4 * (size_t)p + a].im = v.im;
          if (std::hypot(v.re, v.im) > 1e-15) hz[p] = 1;                   // has_nonzero_bc, tbem.rs:247-249
        }
      any = any || hz[p];
    }
    if (any) {
      const size_t b_val = sizeof(c64) * 4 * (size_t)np, b_len = (sizeof(int) * (size_t)np + 15) & ~(size_t)15;
      auto fail2 = [&](int code) { if (P->bcpool) (void)hipFree(P->bcpool); if (P->d_rhs_scratch) (void)hipFree(P->d_rhs_scratch); (void)hipFree(P->d_pair_off); return fail(code); };
      hipError_t e2 = hipMalloc(&P->bcpool, b_val + b_len + (size_t)np);
      if (e2 == hipSuccess) e2 = hipMalloc(&P->d_rhs_scratch, sizeof(c64) * (7 * (size_t)np + (size_t)tot + 1));
      char* bb = (char*)P->bcpool;
      if (e2 == hipSuccess) e2 = hipMemcpy(bb, hv.data(), b_val, hipMemcpyHostToDevice);
      if (e2 == hipSuccess) e2 = hipMemcpy(bb + b_val, hl.data(), sizeof(int) * (size_t)np, hipMemcpyHostToDevice);
      if (e2 == hipSuccess) e2 = hipMemcpy(bb + b_val + b_len, hz.data(), (size_t)np, hipMemcpyHostToDevice);
      if (e2 != hipSuccess) { set_error("boundary-value upload failed: %s", hipGetErrorString(e2)); return fail2(MA_ERR_HIP); }
      P->bc.val = reinterpret_cast<const dc*>(bb); P->bc.len = (const int*)(bb + b_val); P->bc.nz = (const unsigned char*)(bb + b_val + b_len);
      P->has_bc = true;
    }
  }
  for (int i = 0; i < 4; ++i) MA_TRY(hipEventCreate(&P->ev[i]));
#undef MA_TRY
  *out = P;
  return MA_OK;
}

int ma_bem_plan_destroy(ma_bem_plan_t* P) {
  if (!P) return MA_OK;
  (void)hipSetDevice(P->device);
  for (int i = 0; i < 4; ++i) if (P->ev[i]) (void)hipEventDestroy(P->ev[i]);
  if (P->d_pairs) (void)hipFree(P->d_pairs);
  if (P->d_pair_off) (void)hipFree(P->d_pair_off);
  if (P->pool) (void)hipFree(P->pool);
  if (P->bcpool) (void)hipFree(P->bcpool);
  if (P->d_rhs_scratch) (void)hipFree(P->d_rhs_scratch);
  delete P;
  return MA_OK;
}

int ma_bem_plan_device(const ma_bem_plan_t* P, int* device) {
  MA_REQUIRE(P && device, MA_ERR_INVALID, "NULL argument");
  *device = P->device;
  return MA_OK;
}

int ma_bem_plan_num_dofs(const ma_bem_plan_t* P, int32_t* n) {
  MA_REQUIRE(P && n, MA_ERR_INVALID, "NULL argument");
  *n = P->nd;
  return MA_OK;
}

int ma_bem_plan_num_near_pairs(const ma_bem_plan_t* P, int64_t* n) {
  MA_REQUIRE(P && n, MA_ERR_INVALID, "NULL argument");
  *n = P->npairs;
  return MA_OK;
}

extern "C++" {
int ma_bem_make_phys(const ma_bem_plan* P, const ma_physics_t* ph, double bre, double bim, ma::BemPhys* o) {
  MA_REQUIRE(ph, MA_ERR_INVALID, "physics is NULL");
  MA_REQUIRE(std::isfinite(ph->wave_number) && ph->wave_number > 0.0, MA_ERR_INVALID, "wave_number must be finite and > 0");
  // the kernels' sin/cos takes arguments below 2^30 (ma_device_math.hpp: sincos_bounded); k r never gets near that for acoustics
  MA_REQUIRE(std::fabs(ph->wave_number * ph->harmonic_factor) * P->diameter < 5.0e8, MA_ERR_UNSUPPORTED, "k * mesh diameter = %g is beyond the kernels' sin/cos range",
             std::fabs(ph->wave_number * ph->harmonic_factor) * P->diameter);
  o->k = ph->wave_number; o->harmonic = ph->harmonic_factor; o->tau = ph->tau; o->gamma = ph->gamma;
  o->beta_re = bre; o->beta_im = bim;
  const double ka = ph->wave_number * P->avg_radius;       // tbem.rs:118-123
  o->sign = ka < 0.5 ? 1.0 : -1.0;
  return MA_OK;
}
}  // extern "C++"

int ma_bem_plan_assemble_dev(ma_bem_plan_t* P, const ma_physics_t* ph, double bre, double bim, void* dA, void* drhs, void* stream) {
  MA_REQUIRE(P && dA && drhs, MA_ERR_INVALID, "NULL argument");
  BemPhys bp;
  int rc = ma_bem_make_phys(P, ph, bre, bim, &bp);
  if (rc) return rc;
  MA_HIP(hipSetDevice(P->device));
  hipStream_t st = (hipStream_t)stream;
  c64* A = (c64*)dA;
  // TbemSystem.rhs: zero for zero BC values, else the free-term shares and rhs_contributions of the panels that carry values
  if (P->has_bc) { if ((rc = bem_launch_rhs_bc(P->geom, bp, P->bc, P->d_pairs, P->d_pair_off, P->npairs, P->d_rhs_scratch, (c64*)drhs, st))) return rc; }
  else if ((rc = bem_launch_zero((c64*)drhs, P->nd, st))) return rc;
  if (P->timing) MA_HIP(hipEventRecord(P->ev[0], st));
  if ((rc = bem_launch_far(P->geom, bp, A, st))) return rc;
  if (P->timing) MA_HIP(hipEventRecord(P->ev[1], st));
  if ((rc = bem_launch_near(P->geom, bp, P->d_pairs, P->npairs, A, st))) return rc;
  if (P->timing) MA_HIP(hipEventRecord(P->ev[2], st));
  if ((rc = bem_launch_self(P->geom, bp, A, st))) return rc;
  if (P->timing) { MA_HIP(hipEventRecord(P->ev[3], st)); P->ev_valid = true; }
  return MA_OK;
}

// The same mesh at nf wavenumbers (the next systems of a frequency sweep), each into its own matrix and right-hand side: the far
// pairs of up to three systems go through ONE pass that computes every quadrature point's geometry (position, distance, the two
// normal projections) once and only the wavenumber-dependent half (sin / cos, the kernel values) per system; near pairs and self
// terms per system as in ma_bem_plan_assemble_dev. build_tbem_system_with_beta (tbem.rs:96-222) called nf times.
// part `part` of `nparts` of the assembly of nf systems: every part a slice of the far pairs' row strips; part 0 also prepares the
// right-hand sides, the last part also runs the near and self pairs (which overwrite far entries: they come after every far part
// in stream order). nparts = 1 is the whole assembly. A driver that has the matrices' memory early (a sweep assembling ahead)
// feeds the parts to its stream one at a time where that stream would otherwise wait (sweep_plan.hip, bench.py).
static int assemble_multi_part(ma_bem_plan_t* P, int32_t nf, const ma_physics_t* ph, const double* bre, const double* bim, void* const* dA, void* const* drhs,
                               int32_t part, int32_t nparts, void* stream) {
  MA_REQUIRE(P && ph && bre && bim && dA && drhs && nf >= 1 && nf <= 16, MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(nparts >= 1 && part >= 0 && part < nparts, MA_ERR_INVALID, "part %d of %d", part, nparts);
  BemPhys bp[16];
  int rc;
  for (int f = 0; f < nf; ++f) {
    MA_REQUIRE(dA[f] && drhs[f], MA_ERR_INVALID, "system %d has a NULL pointer", f);
    for (int o = 0; o < f; ++o) MA_REQUIRE(dA[o] != dA[f], MA_ERR_INVALID, "systems %d and %d alias", o, f);
    if ((rc = ma_bem_make_phys(P, &ph[f], bre[f], bim[f], &bp[f]))) return rc;
  }
  MA_HIP(hipSetDevice(P->device));
  hipStream_t st = (hipStream_t)stream;
  const bool first = part == 0, last = part == nparts - 1, whole = nparts == 1;
  if (first) for (int f = 0; f < nf; ++f) {
    if (P->has_bc) { if ((rc = bem_launch_rhs_bc(P->geom, bp[f], P->bc, P->d_pairs, P->d_pair_off, P->npairs, P->d_rhs_scratch, (c64*)drhs[f], st))) return rc; }
    else if ((rc = bem_launch_zero((c64*)drhs[f], P->nd, st))) return rc;
  }
  if (P->timing && whole) MA_HIP(hipEventRecord(P->ev[0], st));
  const int strips = bem_far_row_strips(P->geom);
  const int b0 = (int)((long long)strips * part / nparts), b1 = (int)((long long)strips * (part + 1) / nparts);
  for (int f0 = 0; f0 < nf; f0 += 3) {
    c64* As[3]; const int cnt = std::min(3, nf - f0);
    for (int t = 0; t < cnt; ++t) As[t] = (c64*)dA[f0 + t];
    if ((rc = bem_launch_far_multi(P->geom, cnt, bp + f0, As, st, b0, b1 - b0))) return rc;
  }
  if (P->timing && whole) MA_HIP(hipEventRecord(P->ev[1], st));
  // the near pairs likewise three systems per pass (round 4: the leaves and the points' geometry are the mesh's)
  const bool near_multi = true;
  if (last && near_multi)
    for (int f0 = 0; f0 < nf; f0 += 3) {
      c64* As[3]; const int cnt = std::min(3, nf - f0);
      for (int t = 0; t < cnt; ++t) As[t] = (c64*)dA[f0 + t];
      if ((rc = bem_launch_near_multi(P->geom, cnt, bp + f0, As, P->d_pairs, P->npairs, st))) return rc;
    }
  if (last && !near_multi) for (int f = 0; f < nf; ++f) if ((rc = bem_launch_near(P->geom, bp[f], P->d_pairs, P->npairs, (c64*)dA[f], st))) return rc;
  if (P->timing && whole) MA_HIP(hipEventRecord(P->ev[2], st));
  if (last) for (int f = 0; f < nf; ++f) if ((rc = bem_launch_self(P->geom, bp[f], (c64*)dA[f], st))) return rc;
  if (P->timing && whole) { MA_HIP(hipEventRecord(P->ev[3], st)); P->ev_valid = true; }
  return MA_OK;
}
int ma_bem_plan_assemble_multi_dev(ma_bem_plan_t* P, int32_t nf, const ma_physics_t* ph, const double* bre, const double* bim, void* const* dA, void* const* drhs, void* stream) {
  return assemble_multi_part(P, nf, ph, bre, bim, dA, drhs, 0, 1, stream);
}
int ma_bem_plan_assemble_multi_part_dev(ma_bem_plan_t* P, int32_t nf, const ma_physics_t* ph, const double* bre, const double* bim, void* const* dA, void* const* drhs,
                                        int32_t part, int32_t nparts, void* stream) {
  return assemble_multi_part(P, nf, ph, bre, bim, dA, drhs, part, nparts, stream);
}

int ma_bem_plan_set_timing(ma_bem_plan_t* P, int enable) {
  MA_REQUIRE(P, MA_ERR_INVALID, "NULL plan");
  P->timing = enable != 0;
  P->ev_valid = false;
  return MA_OK;
}

int ma_bem_plan_last_timing(ma_bem_plan_t* P, double* out3) {
  MA_REQUIRE(P && out3, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(P->ev_valid, MA_ERR_INVALID, "no timed assemble has run on this plan");
  MA_HIP(hipSetDevice(P->device));
  MA_HIP(hipEventSynchronize(P->ev[3]));
  for (int i = 0; i < 3; ++i) {
    float ms = 0.f;
    MA_HIP(hipEventElapsedTime(&ms, P->ev[i], P->ev[i + 1]));
    out3[i] = ms;
  }
  return MA_OK;
}

int ma_bem_plan_incident_rhs_dev(ma_bem_plan_t* P, const ma_physics_t* ph, double bre, double bim, int kind,
                                 const double* vec3, double are, double aim, int accumulate, void* drhs, void* stream) {
  MA_REQUIRE(P && vec3 && drhs, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(kind == 0 || kind == 1, MA_ERR_INVALID, "kind must be 0 (plane wave) or 1 (point source)");
  BemPhys bp;
  int rc = ma_bem_make_phys(P, ph, bre, bim, &bp);
  if (rc) return rc;
  MA_HIP(hipSetDevice(P->device));
  return bem_launch_incident(P->geom, bp, kind, vec3, are, aim, accumulate, (c64*)drhs, (hipStream_t)stream);
}

// Parity-test hooks (not part of the reference seam): raw panel integrals on the device.
// pairs: npairs x (i, j) with i != j, panel indices in plan order; out: npairs x 5 complex
// {leaf count, G, H, H^T, E} (IntegrationResult, types.rs:722-734).
int ma_bem_plan_probe_pairs(ma_bem_plan_t* P, const ma_physics_t* ph, int64_t npairs, const int32_t* pairs, ma_c64* out5) {
  MA_REQUIRE(P && pairs && out5 && npairs >= 0, MA_ERR_INVALID, "bad argument");
  BemPhys bp;
  int rc = ma_bem_make_phys(P, ph, 0.0, 0.0, &bp);
  if (rc) return rc;
  for (int64_t q = 0; q < npairs; ++q) {
    int i = pairs[2 * q], j = pairs[2 * q + 1];
    MA_REQUIRE(i >= 0 && i < P->np && j >= 0 && j < P->np && i != j, MA_ERR_INVALID, "pair %lld = (%d,%d) invalid", (long long)q, i, j);
  }
  if (npairs == 0) return MA_OK;
  MA_HIP(hipSetDevice(P->device));
  int2* dp = nullptr; c64* dout = nullptr;
  MA_HIP(hipMalloc(&dp, sizeof(int2) * (size_t)npairs));
  MA_HIP(hipMalloc(&dout, sizeof(c64) * 5 * (size_t)npairs));
  MA_HIP(hipMemcpy(dp, pairs, sizeof(int2) * (size_t)npairs, hipMemcpyHostToDevice));
  rc = bem_launch_probe_pairs(P->geom, bp, dp, npairs, dout, nullptr);
  if (!rc) {
    hipError_t e = hipMemcpy(out5, dout, sizeof(c64) * 5 * (size_t)npairs, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("hipMemcpy failed: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; }
  }
  (void)hipFree(dp); (void)hipFree(dout);
  return rc;
}

// out: np x 5 complex {point count, G, H, H^T, E} of every panel's self integral.
int ma_bem_plan_probe_self(ma_bem_plan_t* P, const ma_physics_t* ph, ma_c64* out5) {
  MA_REQUIRE(P && out5, MA_ERR_INVALID, "bad argument");
  BemPhys bp;
  int rc = ma_bem_make_phys(P, ph, 0.0, 0.0, &bp);
  if (rc) return rc;
  MA_HIP(hipSetDevice(P->device));
  c64* dout = nullptr;
  MA_HIP(hipMalloc(&dout, sizeof(c64) * 5 * (size_t)P->np));
  rc = bem_launch_probe_self(P->geom, bp, dout, nullptr);
  if (!rc) {
    hipError_t e = hipMemcpy(out5, dout, sizeof(c64) * 5 * (size_t)P->np, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("hipMemcpy failed: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; }
  }
  (void)hipFree(dout);
  return rc;
}

// near-pair list (plan order panel indices), for the parity tests: out = npairs x 2 int32
int ma_bem_plan_get_near_pairs(const ma_bem_plan_t* P, int32_t* out) {
  MA_REQUIRE(P && (out || P->npairs == 0), MA_ERR_INVALID, "bad argument");
  if (P->npairs == 0) return MA_OK;
  MA_HIP(hipSetDevice(P->device));
  MA_HIP(hipMemcpy(out, P->d_pairs, sizeof(int2) * (size_t)P->npairs, hipMemcpyDeviceToHost));
  return MA_OK;
}

int ma_bem_assemble_tbem(const ma_mesh_t* mesh, const ma_physics_t* ph, double bre, double bim, ma_c64* A, ma_c64* rhs) {
  MA_REQUIRE(A && rhs, MA_ERR_INVALID, "A or rhs is NULL");
  ma_bem_plan_t* P = nullptr;
  int dev = 0;
  if (const char* s = getenv("MA_DEVICE")) dev = atoi(s);
  int rc = ma_bem_plan_create(mesh, dev, &P);
  if (rc) return rc;
  const size_t n = (size_t)P->nd;
  void *dA = nullptr, *dr = nullptr;
  hipError_t e = hipMalloc(&dA, n * n * sizeof(c64));
  if (e == hipSuccess) e = hipMalloc(&dr, n * sizeof(c64));
  if (e != hipSuccess) {
    set_error("hipMalloc for a %zu x %zu system failed: %s", n, n, hipGetErrorString(e));
    if (dA) (void)hipFree(dA);
    ma_bem_plan_destroy(P);
    return MA_ERR_NOMEM;
  }
  rc = ma_bem_plan_assemble_dev(P, ph, bre, bim, dA, dr, nullptr);
  if (!rc) {
    e = hipMemcpy(A, dA, n * n * sizeof(c64), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(rhs, dr, n * sizeof(c64), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("copy back failed: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; }
  }
  (void)hipFree(dA); (void)hipFree(dr);
  ma_bem_plan_destroy(P);
  return rc;
}

int ma_bem_incident_rhs(int n, const double* centers, const double* normals, const ma_physics_t* ph,
                        double bre, double bim, int kind, const double* vec3, double are, double aim, ma_c64* rhs) {
  MA_REQUIRE(n > 0 && centers && normals && ph && vec3 && rhs, MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(kind == 0 || kind == 1, MA_ERR_INVALID, "kind must be 0 or 1");
  int dev = 0;
  if (const char* s = getenv("MA_DEVICE")) dev = atoi(s);
  int rc = use_device(dev);
  if (rc) return rc;
  // transient SoA upload of centres and normals; identity dof
  const size_t stride = ((size_t)n + 63) & ~(size_t)63;
  std::vector<double> h(6 * stride, 0.0);
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) { h[(size_t)d * stride + i] = centers[3 * i + d]; h[(size_t)(3 + d) * stride + i] = normals[3 * i + d]; }
  std::vector<int> hd(stride);
  for (size_t i = 0; i < stride; ++i) hd[i] = (int)i;
  double* dbuf = nullptr; int* ddof = nullptr; c64* dr = nullptr;
  MA_HIP(hipMalloc(&dbuf, 6 * stride * sizeof(double)));
  MA_HIP(hipMalloc(&ddof, stride * sizeof(int)));
  MA_HIP(hipMalloc(&dr, (size_t)n * sizeof(c64)));
  MA_HIP(hipMemcpy(dbuf, h.data(), 6 * stride * sizeof(double), hipMemcpyHostToDevice));
  MA_HIP(hipMemcpy(ddof, hd.data(), stride * sizeof(int), hipMemcpyHostToDevice));
  BemGeom g{};
  g.np = n; g.nd = n;
  for (int d = 0; d < 3; ++d) { g.c[d] = dbuf + (size_t)d * stride; g.nx[d] = dbuf + (size_t)(3 + d) * stride; }
  g.dof = ddof;
  BemPhys bp{};
  bp.k = ph->wave_number; bp.harmonic = ph->harmonic_factor; bp.tau = ph->tau; bp.gamma = ph->gamma; bp.beta_re = bre; bp.beta_im = bim; bp.sign = 1.0;
  rc = bem_launch_incident(g, bp, kind, vec3, are, aim, 0, dr, nullptr);
  if (!rc) {
    hipError_t e = hipMemcpy(rhs, dr, (size_t)n * sizeof(c64), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("copy back failed: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; }
  }
  (void)hipFree(dbuf); (void)hipFree(ddof); (void)hipFree(dr);
  return rc;
}

// IncidentField::evaluate_pressure (incident.rs:93-166) and ::evaluate_normal_derivative (:177-280) at arbitrary points:
// p and/or dp/dn (either output may be NULL; normals may be NULL when dp/dn is not asked for). They are the two terms of
// compute_rhs_with_beta = -(gamma p + beta tau dp/dn), read off with (gamma, beta) = (1, 0) and (0, 1).
int ma_bem_incident_evaluate(int n, const double* points, const double* normals, const ma_physics_t* ph, int kind, const double* vec3, double are, double aim,
                             ma_c64* p_out, ma_c64* dpdn_out) {
  MA_REQUIRE(n > 0 && points && ph && vec3 && (p_out || dpdn_out), MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(!dpdn_out || normals, MA_ERR_INVALID, "normals are needed for dp/dn");
  std::vector<double> zero;
  if (!normals) { zero.assign(3 * (size_t)n, 0.0); normals = zero.data(); }
  ma_physics_t q = *ph;
  int rc = MA_OK;
  if (p_out) {
    q.gamma = 1.0; q.tau = 1.0;
    rc = ma_bem_incident_rhs(n, points, normals, &q, 0.0, 0.0, kind, vec3, are, aim, p_out);
    for (int i = 0; i < n && !rc; ++i) { p_out[i].re = -p_out[i].re; p_out[i].im = -p_out[i].im; }
  }
  if (dpdn_out && !rc) {
    q.gamma = 0.0; q.tau = 1.0;
    rc = ma_bem_incident_rhs(n, points, normals, &q, 1.0, 0.0, kind, vec3, are, aim, dpdn_out);
    for (int i = 0; i < n && !rc; ++i) { dpdn_out[i].re = -dpdn_out[i].re; dpdn_out[i].im = -dpdn_out[i].im; }
  }
  return rc;
}

}  // extern "C"
