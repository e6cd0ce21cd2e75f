// op_plan.hip — the operator boundary (LinearOperator<Complex64>, math-solvers/src/traits.rs:316-327) and
// restarted GMRES (math-solvers/src/iterative/gmres.rs:105-277) on the device.
#include "op_kernels.hpp"
#include "lu_kernels.hpp"
#include "csr_kernels.hpp"
#include "fmm_plan.hpp"
#include "amg_setup.hpp"
#include <chrono>
#include "ma_tables.h"
#include <vector>
#include <algorithm>
#include <complex>
#include <new>
#include <cmath>
#include <mutex>
#include <condition_variable>
#include <dlfcn.h>
#include <rccl/rccl.h>     // types and prototypes only: the library is bound at first use (dlopen), not at link time

using namespace ma;
typedef std::complex<double> cplx;

struct ma_csr;   // csr_plan.hip
extern "C" int ma_csr_spmv_dev(ma_csr* h, const void* d_x, void* d_y, void* stream);
extern "C" int ma_csr_num_rows(const ma_csr* h, int64_t* n, int64_t* nnz);
extern "C" int ma_csr_transpose(ma_csr* h, ma_csr** out);
extern "C" int ma_csr_destroy(ma_csr* h);
extern "C" unsigned long long ma_csr_epoch(const ma_csr* h);
extern "C" int ma_csr_refresh_transpose(const ma_csr* src, ma_csr* dst, int* rebuild);

struct ma_op;
// one row block of a matrix-free TBEM operator spread over several GPUs (kind 3): its own plan, operator, stream and
// replicas of x and y on its device
struct OpShard {
  int device = 0; int row0 = 0, row1 = 0;
  ma_bem_plan* plan = nullptr; ma_op* op = nullptr;
  hipStream_t st = nullptr; hipEvent_t done = nullptr;
  c64* d_x = nullptr; c64* d_y = nullptr;
};

struct ma_op {
  int kind = 0;                 // 0 dense, 1 csr, 2 on-the-fly TBEM, 3 on-the-fly TBEM row-sharded over several devices
  int device = 0;
  long long n = 0;
  c64* dA = nullptr; bool own_A = false;
  ma_csr* csr = nullptr;
  ma_csr* csr_t = nullptr;      // transposed CSR operator, built at the first apply_transpose (owned)
  unsigned long long csr_t_epoch = 0;   // value epoch of `csr` the transposed copy corresponds to
  c64* d_tpart = nullptr;       // dense A^T x: per-row-chunk partial sums
  c64* d_cx = nullptr;          // conj(x) / pre-conjugation result of the hermitian CSR apply
  // TBEM
  ma_bem_plan* plan = nullptr; BemPhys ph{}; int row0 = 0, row1 = 0, nchunks = 1;
  c64* d_corr = nullptr; c64* d_diag = nullptr; c64* d_partial = nullptr;
  // transposed apply: the plan's near pairs listed by column, and per-row-chunk partial sums over all np columns
  long long* d_t_off = nullptr; int* d_t_idx = nullptr; c64* d_tpartial = nullptr;
  // staging for the host-buffer entry points
  c64* d_x = nullptr; c64* d_y = nullptr;
  // kind 3: the shards (shards[0] lives on `device`, the home of every vector the callers pass), the event that says x is
  // ready on the caller's stream, and ndev x n partial results of a transposed apply gathered on the home device
  std::vector<OpShard> shards; hipEvent_t ev_home = nullptr; c64* d_tgather = nullptr;
  // kind 4: the reference's single-level fast multipole operator (SlfmmSystem, assembly/slfmm.rs)
  ma_slfmm* fmm = nullptr;
  // kind 6: the multi-level operator (MlfmmSystem, assembly/mlfmm.rs)
  ma_mlfmm* mlfmm = nullptr;
  // kind 8: one RANK's row block of an operator whose rows are spread over processes: `inner` (not owned) writes rows [row0, row1)
  // of y, then the caller's exchange (an all-gather over the ranks' communicator: RCCL with the "nccl" backend) completes y
  ma_op* inner = nullptr; ma_gather_fn gather = nullptr; void* gather_user = nullptr;
  void (*gather_free)(void*) = nullptr;   // set when the library owns gather_user (ma_op_create_gathered_rccl)
};

extern "C" int ma_op_destroy(ma_op_t* o);
namespace {
void op_free(ma_op* o) {
  for (OpShard& sh : o->shards) {
    (void)hipSetDevice(sh.device);
    if (sh.op) (void)ma_op_destroy(sh.op);
    if (sh.plan) (void)ma_bem_plan_destroy(sh.plan);
    if (sh.d_x) (void)hipFree(sh.d_x);
    if (sh.d_y) (void)hipFree(sh.d_y);
    if (sh.done) (void)hipEventDestroy(sh.done);
    if (sh.st) (void)hipStreamDestroy(sh.st);
  }
  o->shards.clear();
  (void)hipSetDevice(o->device);
  if (o->fmm) { slfmm_destroy(o->fmm); o->fmm = nullptr; }
  if (o->mlfmm) { mlfmm_destroy(o->mlfmm); o->mlfmm = nullptr; }
  if (o->ev_home) (void)hipEventDestroy(o->ev_home);
  if (o->d_tgather) (void)hipFree(o->d_tgather);
  if (o->own_A && o->dA) (void)hipFree(o->dA);
  void* p[] = {o->d_corr, o->d_diag, o->d_partial, o->d_x, o->d_y, o->d_tpart, o->d_cx, o->d_t_off, o->d_t_idx, o->d_tpartial};
  for (void* q : p) if (q) (void)hipFree(q);
  if (o->csr_t) (void)ma_csr_destroy(o->csr_t);
  if (o->gather_free && o->gather_user) { o->gather_free(o->gather_user); o->gather_user = nullptr; }
}
int op_stage(ma_op* o) {
  MA_HIP(hipMalloc(&o->d_x, sizeof(c64) * (size_t)o->n));
  MA_HIP(hipMalloc(&o->d_y, sizeof(c64) * (size_t)o->n));
  return MA_OK;
}
}  // namespace

extern "C" {

// DenseOperator::new(matrix) (math-bem/src/core/solver/fmm_interface.rs:25-53): A is n x n row-major on the host
int ma_op_create_dense(int64_t n, const ma_c64* A, int device, ma_op_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(n > 0 && A, MA_ERR_INVALID, "bad argument");
  int rc = use_device(device); if (rc) return rc;
  ma_op* o = new (std::nothrow) ma_op(); MA_REQUIRE(o, MA_ERR_NOMEM, "host allocation failed");
  o->kind = 0; o->device = device; o->n = n; o->own_A = true;
  hipError_t e = hipMalloc(&o->dA, sizeof(c64) * (size_t)n * (size_t)n);
  if (e == hipSuccess) e = hipMemcpy(o->dA, A, sizeof(c64) * (size_t)n * (size_t)n, hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("dense operator upload failed: %s", hipGetErrorString(e)); op_free(o); delete o; return MA_ERR_NOMEM; }
  rc = op_stage(o); if (rc) { op_free(o); delete o; return rc; }
  *out = o; return MA_OK;
}
// dense operator over a matrix that already lives in HBM (borrowed; e.g. the output of ma_bem_plan_assemble_dev)
int ma_op_create_dense_dev(int64_t n, const void* d_A, int device, ma_op_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(n > 0 && d_A, MA_ERR_INVALID, "bad argument");
  int rc = use_device(device); if (rc) return rc;
  ma_op* o = new (std::nothrow) ma_op(); MA_REQUIRE(o, MA_ERR_NOMEM, "host allocation failed");
  o->kind = 0; o->device = device; o->n = n; o->dA = (c64*)d_A; o->own_A = false;
  rc = op_stage(o); if (rc) { op_free(o); delete o; return rc; }
  *out = o; return MA_OK;
}
// impl LinearOperator for CsrMatrix (csr.rs:420-440); the CSR handle is borrowed
int ma_op_create_csr(ma_csr_t* csr, ma_op_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(csr, MA_ERR_INVALID, "csr is NULL");
  int64_t n = 0, nnz = 0;
  int rc = ma_csr_num_rows(csr, &n, &nnz); if (rc) return rc;
  ma_op* o = new (std::nothrow) ma_op(); MA_REQUIRE(o, MA_ERR_NOMEM, "host allocation failed");
  o->kind = 1; o->n = n; o->csr = csr;
  int dev = 0; MA_HIP(hipGetDevice(&dev)); o->device = dev;
  rc = op_stage(o); if (rc) { op_free(o); delete o; return rc; }
  *out = o; return MA_OK;
}
// Matrix-free TBEM operator (new, must equal A x of the dense TBEM matrix): rows [row0, row1) of
// y = A x are produced (the whole operator for row0 = 0, row1 = num_dofs; a row block per GPU when sharded).
// The plan is borrowed and must outlive the operator.
int ma_op_create_tbem(ma_bem_plan_t* P, const ma_physics_t* physics, double beta_re, double beta_im, int32_t row0, int32_t row1, ma_op_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(P, MA_ERR_INVALID, "plan is NULL");
  MA_REQUIRE(row0 >= 0 && row1 <= P->np && row0 < row1, MA_ERR_INVALID, "row range [%d,%d) outside 0..%d", row0, row1, P->np);
  ma_op* o = new (std::nothrow) ma_op(); MA_REQUIRE(o, MA_ERR_NOMEM, "host allocation failed");
  int rc = ma_bem_make_phys(P, physics, beta_re, beta_im, &o->ph);
  if (rc) { delete o; return rc; }
  o->kind = 2; o->device = P->device; o->n = P->nd; o->plan = P; o->row0 = row0; o->row1 = row1;
  MA_HIP(hipSetDevice(P->device));
  // enough (row strip, column chunk) workgroups to fill 256 CUs several times over
  const int strips = (row1 - row0 + 255) / 256;
  int nch = (2048 + strips - 1) / strips; if (nch < 1) nch = 1; if (nch > 64) nch = 64; if (nch > P->np) nch = P->np;
  o->nchunks = nch;
  hipError_t e = hipMalloc(&o->d_corr, sizeof(c64) * (size_t)(P->npairs > 0 ? P->npairs : 1));
  if (e == hipSuccess) e = hipMalloc(&o->d_diag, sizeof(c64) * (size_t)P->np);
  if (e == hipSuccess) e = hipMalloc(&o->d_partial, sizeof(c64) * (size_t)std::max(nch, op_tbem_matvec_strips(P->np) + bem_quad_strips(P->geom)) * (size_t)(row1 - row0));
  c64* tmp = nullptr;
  if (e == hipSuccess) e = hipMalloc(&tmp, sizeof(c64) * (size_t)((P->npairs > P->np ? P->npairs : P->np) + 1));
  int2* dpairs_diag = nullptr;
  if (e == hipSuccess) e = hipMalloc(&dpairs_diag, sizeof(int2) * (size_t)P->np);
  if (e != hipSuccess) { set_error("operator workspace allocation failed: %s", hipGetErrorString(e)); if (tmp) (void)hipFree(tmp); if (dpairs_diag) (void)hipFree(dpairs_diag); op_free(o); delete o; return MA_ERR_NOMEM; }
  double t13[13][3];
  for (int q = 0; q < 13; ++q) { t13[q][0] = mat_tri13[q][0]; t13[q][1] = mat_tri13[q][1]; t13[q][2] = mat_tri13[q][2] * 0.5; }
  rc = op_upload_tables(t13);
  // corrections: true coefficient (K2 / K3 kernels) minus the 13-point coefficient the streaming kernel will add
  if (!rc) rc = bem_launch_near_list_values(P->geom, o->ph, P->d_pairs, P->npairs, o->d_corr, nullptr);
  if (!rc) rc = op_launch_pairs13(P->geom, o->ph, P->d_pairs, P->npairs, tmp, nullptr);
  if (!rc) rc = op_launch_sub_inplace(P->npairs, o->d_corr, tmp, nullptr);
  if (!rc) rc = bem_launch_self_list_values(P->geom, o->ph, o->d_diag, nullptr);
  if (!rc) {
    std::vector<int2> hp((size_t)P->np);
    for (int i = 0; i < P->np; ++i) hp[i] = make_int2(i, i);
    e = hipMemcpy(dpairs_diag, hp.data(), sizeof(int2) * (size_t)P->np, hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error("upload failed: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; }
  }
  if (!rc) rc = op_launch_pairs13(P->geom, o->ph, dpairs_diag, P->np, tmp, nullptr);
  if (!rc) rc = op_launch_sub_inplace(P->np, o->d_diag, tmp, nullptr);
  if (!rc) { e = hipDeviceSynchronize(); if (e != hipSuccess) { set_error("correction kernels failed: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; } }
  (void)hipFree(tmp); (void)hipFree(dpairs_diag);
  if (!rc) rc = op_stage(o);
  if (rc) { op_free(o); delete o; return rc; }
  *out = o; return MA_OK;
}

// Matrix-free TBEM operator row-sharded over the GPUs of one node (BASELINE.json configs[4]): device g
// owns the collocation rows [g N / G, (g+1) N / G) -- a whole BEM plan per device (geometry is O(N)), the row-block operator of
// ma_op_create_tbem on it. All vectors the callers see live on devices[0] ("home"), so ma_gmres and the preconditioners
// drive this operator like any other. The exchange per apply is the all-gather of SURVEY C2, done with peer copies inside
// the library: x goes from home to every shard, every shard returns its y slice into the caller's y (16 B N in total).
int ma_op_create_tbem_multi(const ma_mesh_t* mesh, const ma_physics_t* physics, double beta_re, double beta_im, const int32_t* devices, int32_t ndev,
                            ma_op_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(mesh && physics && devices && ndev >= 1 && ndev <= 64, MA_ERR_INVALID, "bad argument");
  int count = 0;
  int rc = ma_device_count(&count); if (rc) return rc;
  MA_REQUIRE(count > 0, MA_ERR_NO_DEVICE, "no gfx950 device visible");
  for (int g = 0; g < ndev; ++g) {
    MA_REQUIRE(devices[g] >= 0 && devices[g] < count, MA_ERR_INVALID, "device %d (entry %d) outside 0..%d", devices[g], g, count - 1);
#ifdef MA_DIAGNOSTICS
    // diagnostic build only (MA_TEST_ALLOW_DUPLICATE_DEVICES=1): several shards on one GPU, so that a one-GPU box exercises the exchange
    const bool dup_ok = getenv("MA_TEST_ALLOW_DUPLICATE_DEVICES") != nullptr;
#else
    const bool dup_ok = false;
#endif
    for (int q = 0; q < g; ++q) MA_REQUIRE(devices[q] != devices[g] || dup_ok, MA_ERR_INVALID, "device %d listed twice", devices[g]);
  }
  ma_op* o = new (std::nothrow) ma_op(); MA_REQUIRE(o, MA_ERR_NOMEM, "host allocation failed");
  o->kind = 3; o->device = devices[0];
  o->shards.resize((size_t)ndev);
  auto fail = [&](int code) { op_free(o); delete o; return code; };
  for (int g = 0; g < ndev && !rc; ++g) {
    OpShard& sh = o->shards[(size_t)g];
    sh.device = devices[g];
    rc = ma_bem_plan_create(mesh, sh.device, &sh.plan);
    if (rc) break;
    const long long np = sh.plan->np;
    if (np < ndev) { set_error("%d devices for %lld panels", ndev, np); rc = MA_ERR_INVALID; break; }
    sh.row0 = (int)(np * g / ndev); sh.row1 = (int)(np * (g + 1) / ndev);
    o->n = sh.plan->nd;
    rc = ma_op_create_tbem(sh.plan, physics, beta_re, beta_im, sh.row0, sh.row1, &sh.op);
    if (rc) break;
    hipError_t e = hipSetDevice(sh.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&sh.st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sh.done, hipEventDisableTiming);
    if (e == hipSuccess && g > 0) e = hipMalloc(&sh.d_x, sizeof(c64) * (size_t)o->n);
    if (e == hipSuccess) e = hipMalloc(&sh.d_y, sizeof(c64) * (size_t)o->n);
    if (e != hipSuccess) { set_error("sharded operator: device %d: %s", sh.device, hipGetErrorString(e)); rc = MA_ERR_HIP; }
  }
  if (rc) return fail(rc);
  hipError_t e = hipSetDevice(o->device);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&o->ev_home, hipEventDisableTiming);
  if (e != hipSuccess) { set_error("sharded operator: %s", hipGetErrorString(e)); return fail(MA_ERR_HIP); }
  rc = op_stage(o);
  if (rc) return fail(rc);
  *out = o; return MA_OK;
}
// SlfmmSystem as a LinearOperator (slfmm.rs:378-395): build_slfmm_system(elements, nodes, clusters, physics, n_theta, n_phi, n_terms)
// over the plan's mesh; apply = matvec, apply_transpose = matvec_transpose. The plan is borrowed.
int ma_op_create_slfmm(ma_bem_plan_t* plan, const ma_clusters_t* clusters, const ma_physics_t* physics, int32_t n_theta, int32_t n_phi, int32_t n_terms, ma_op_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(plan, MA_ERR_INVALID, "plan is NULL");
  ma_op* o = new (std::nothrow) ma_op(); MA_REQUIRE(o, MA_ERR_NOMEM, "host allocation failed");
  o->kind = 4; o->device = plan->device; o->n = plan->nd; o->plan = plan;
  int rc = slfmm_create(plan, clusters, physics, n_theta, n_phi, n_terms, &o->fmm);
  if (!rc) rc = op_stage(o);
  if (rc) { op_free(o); delete o; return rc; }
  *out = o; return MA_OK;
}
// build_cluster_tree(elements, target_elements_per_leaf, physics) (mlfmm.rs:979-1038): the tree only looks at the elements' centres
int ma_cluster_tree_build(const ma_mesh_t* mesh, int32_t target_elements_per_leaf, double wave_number, ma_cluster_tree_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(mesh && mesh->center && mesh->n_elem >= 1, MA_ERR_INVALID, "mesh without element centres");
  return cluster_tree_build(mesh->n_elem, mesh->center, target_elements_per_leaf, wave_number, out);
}
int ma_cluster_tree_destroy(ma_cluster_tree_t* tree) { cluster_tree_destroy(tree); return MA_OK; }
int ma_cluster_tree_num_levels(const ma_cluster_tree_t* tree, int32_t* levels) {
  MA_REQUIRE(tree && levels, MA_ERR_INVALID, "NULL argument");
  *levels = cluster_tree_num_levels(tree);
  return MA_OK;
}
int ma_cluster_tree_level_info(const ma_cluster_tree_t* tree, int32_t level, int32_t* n_clusters, int32_t* expansion_terms, int32_t* theta_points, int32_t* phi_points,
                               int64_t* n_elem_listed, int64_t* n_near, int64_t* n_far, int64_t* n_sons) {
  MA_REQUIRE(tree, MA_ERR_INVALID, "NULL tree");
  long long a = 0, b = 0, c = 0, d = 0;
  int rc = cluster_tree_level_info(tree, level, n_clusters, expansion_terms, theta_points, phi_points, &a, &b, &c, &d);
  if (rc) return rc;
  if (n_elem_listed) *n_elem_listed = a;
  if (n_near) *n_near = b;
  if (n_far) *n_far = c;
  if (n_sons) *n_sons = d;
  return MA_OK;
}
int ma_cluster_tree_level_get(const ma_cluster_tree_t* tree, int32_t level, double* center, double* radius, int32_t* elem_ptr, int32_t* elem_idx, int32_t* near_ptr, int32_t* near_idx,
                              int32_t* far_ptr, int32_t* far_idx, int32_t* son_ptr, int32_t* son_idx, int32_t* father) {
  MA_REQUIRE(tree, MA_ERR_INVALID, "NULL tree");
  return cluster_tree_level_get(tree, level, center, radius, elem_ptr, elem_idx, near_ptr, near_idx, far_ptr, far_idx, son_ptr, son_idx, father);
}
// MlfmmOperator (fmm_interface.rs:98-135) over build_mlfmm_system(elements, nodes, cluster_tree, physics) (mlfmm.rs:483-558): apply = MlfmmSystem::matvec;
// apply_transpose is unimplemented!() in the reference and MA_ERR_UNSUPPORTED here. The plan is borrowed, the tree is only read during the call.
int ma_op_create_mlfmm(ma_bem_plan_t* plan, const ma_cluster_tree_t* tree, const ma_physics_t* physics, ma_op_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(plan, MA_ERR_INVALID, "plan is NULL");
  ma_op* o = new (std::nothrow) ma_op(); MA_REQUIRE(o, MA_ERR_NOMEM, "host allocation failed");
  o->kind = 6; o->device = plan->device; o->n = plan->nd; o->plan = plan;
  int rc = mlfmm_create(plan, tree, physics, &o->mlfmm);
  if (!rc) rc = op_stage(o);
  if (rc) { op_free(o); delete o; return rc; }
  *out = o; return MA_OK;
}
// which of the three forms of the upward / downward passes an SLFMM operator runs (2 phases recomputed, 1 stored table, 0 libm)
int ma_op_slfmm_phase_mode(ma_op_t* o, int32_t* mode) {
  MA_REQUIRE(o && mode && o->kind == 4 && o->fmm, MA_ERR_INVALID, "not an SLFMM operator");
  *mode = slfmm_phase_mode(o->fmm);
  return MA_OK;
}
// SlfmmSystem::extract_near_field_matrix (slfmm.rs:104-132): [N] as a dense num_dofs x num_dofs matrix (host buffer, row-major)
int ma_op_slfmm_near_matrix(ma_op_t* o, ma_c64* A_rowmajor) {
  MA_REQUIRE(o && A_rowmajor && o->kind == 4 && o->fmm, MA_ERR_INVALID, "not a single-level FMM operator");
  MA_HIP(hipSetDevice(o->device));
  c64* dA = nullptr;
  const size_t bytes = sizeof(c64) * (size_t)o->n * (size_t)o->n;
  if (hipMalloc(&dA, bytes) != hipSuccess) { set_error("near-field matrix of %lld dofs does not fit the device", o->n); return MA_ERR_NOMEM; }
  int rc = slfmm_near_matrix(o->fmm, dA, nullptr);
  if (!rc && hipMemcpy(A_rowmajor, dA, bytes, hipMemcpyDeviceToHost) != hipSuccess) { set_error("copy back failed"); rc = MA_ERR_HIP; }
  (void)hipFree(dA);
  return rc;
}
int ma_op_num_shards(const ma_op_t* o, int32_t* shards, int32_t* row_begin_or_null, int32_t* device_or_null) {
  MA_REQUIRE(o && shards, MA_ERR_INVALID, "NULL argument");
  *shards = o->kind == 3 ? (int32_t)o->shards.size() : 1;
  for (size_t g = 0; o->kind == 3 && g < o->shards.size(); ++g) {
    if (row_begin_or_null) row_begin_or_null[g] = o->shards[g].row0;
    if (device_or_null) device_or_null[g] = o->shards[g].device;
  }
  return MA_OK;
}

// LinearOperator over PROCESSES (SURVEY 8e.2, config #5 as the driver launches it: one rank per GPU): `inner` is this rank's
// operator restricted to rows [row0, row1) (ma_op_create_tbem with a row range); after it has written its rows of y, `gather` must
// complete y in place -- every other rank's rows -- ordered on `stream` (an all-gather on the ranks' communicator; the library
// does not link a collective library itself: the communicator stays with the caller, behind `user`). The handle is an ordinary
// ma_op_t: ma_gmres / ma_gmres_preconditioned / ma_gmres_pipelined drive it, every rank running the same iteration on the same
// vectors. apply_transpose / apply_hermitian are not defined for it. `inner` is not owned and must outlive the handle.
int ma_op_create_gathered(ma_op_t* inner, int64_t row0, int64_t row1, ma_gather_fn gather, void* user, ma_op_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(inner && gather, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(row0 >= 0 && row0 <= row1 && row1 <= inner->n, MA_ERR_INVALID, "row block [%lld, %lld) outside 0..%lld", (long long)row0, (long long)row1, inner->n);
  ma_op* o = new (std::nothrow) ma_op(); MA_REQUIRE(o, MA_ERR_NOMEM, "host allocation failed");
  o->kind = 8; o->device = inner->device; o->n = inner->n; o->inner = inner; o->gather = gather; o->gather_user = user;
  o->row0 = (int)row0; o->row1 = (int)row1;
  MA_HIP(hipSetDevice(o->device));
  if (hipMalloc(&o->d_x, sizeof(c64) * (size_t)o->n) != hipSuccess || hipMalloc(&o->d_y, sizeof(c64) * (size_t)o->n) != hipSuccess) {
    set_error("rank-sharded operator: staging vectors do not fit"); op_free(o); delete o; return MA_ERR_NOMEM;
  }
  *out = o;
  return MA_OK;
}
// ---- RCCL inside the library (round 4; north_star: "RCCL over xGMI only for the block reductions"): the row exchange of a rank-sharded
// operator as ncclAllGather on the operator's stream, so that a host that keeps one rank per GPU needs no callback per apply. librccl is
// bound at first use with dlopen -- in a process that already holds a copy (torch.distributed's "nccl" backend brings its own) the
// loader hands back THAT copy, so a communicator and the collective always come from one library instance; a process that never
// shards rows never loads it. Communicators are made through ma_rccl_comm_create or by the caller with the same librccl.
namespace {
struct RcclApi {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) { api.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (api.h) break; }
    if (!api.h) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.h, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.h, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(api.h, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.h, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
  });
  return &api;
}
#ifdef MA_DIAGNOSTICS
// Diagnostic build only: VIRTUAL ranks in one process on one device, so that a one-GPU box runs the nranks > 1 arithmetic of
// rccl_gather_cb (block offsets, the padded last block, the status entries, the copies back) which RCCL itself refuses there (two ranks
// on one device). A "communicator" of the loopback is {group, rank}; the collective has the semantics of ncclAllGather for ranks that are
// host threads: every rank posts its block on the group's board (its stream, synchronised), all meet at a host barrier, every rank
// copies the whole board into its receive buffer, all meet again. ma_rccl_test_loopback_* below; installed INSTEAD of ncclAllGather.
struct LoopGroup {
  int nranks = 0, device = 0; size_t bytes = 0; char* board = nullptr;
  std::mutex mu; std::condition_variable cv; int arrived = 0; long long gen = 0;
  int poison_rank = -1;                                      // this rank's status entry is forced to "a wait was abandoned" on the board
};
struct LoopComm { unsigned magic; LoopGroup* g; int rank; };
void loop_barrier(LoopGroup* g) {
  std::unique_lock<std::mutex> lk(g->mu);
  const long long my = g->gen;
  if (++g->arrived == g->nranks) { g->arrived = 0; ++g->gen; g->cv.notify_all(); }
  else g->cv.wait(lk, [&] { return g->gen != my; });
}
ncclResult_t loop_allgather(const void* send, void* recv, size_t count, ncclDataType_t, ncclComm_t comm, hipStream_t st) {
  LoopComm* c = (LoopComm*)comm;
  if (!c || c->magic != 0x4c4f4f50u) return ncclInvalidArgument;
  LoopGroup* g = c->g;
  const size_t bytes = count * sizeof(double);
  {
    std::lock_guard<std::mutex> lk(g->mu);
    if (!g->board || g->bytes != bytes) {
      if (g->board) (void)hipFree(g->board);
      if (hipMalloc(&g->board, bytes * (size_t)g->nranks) != hipSuccess) return ncclSystemError;
      g->bytes = bytes;
    }
  }
  if (hipMemcpyAsync(g->board + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclSystemError;
  if (g->poison_rank == c->rank) {                           // the block's last complex entry is the rank's status
    const double one[2] = {1.0, 0.0};
    if (hipMemcpyAsync(g->board + (size_t)c->rank * bytes + bytes - sizeof(one), one, sizeof(one), hipMemcpyHostToDevice, st) != hipSuccess) return ncclSystemError;
  }
  if (hipStreamSynchronize(st) != hipSuccess) return ncclSystemError;
  loop_barrier(g);                                           // every rank has posted
  if (hipMemcpyAsync(recv, g->board, bytes * (size_t)g->nranks, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclSystemError;
  if (hipStreamSynchronize(st) != hipSuccess) return ncclSystemError;
  loop_barrier(g);                                           // nobody posts the next round before everybody has read this one
  return ncclSuccess;
}
#endif

#define MA_RCCL(call)                                                                                        \
  do {                                                                                                       \
    ncclResult_t r_ = (call);                                                                                \
    if (r_ != ncclSuccess) { set_error("%s failed: %s", #call, R->GetErrorString(r_)); return MA_ERR_HIP; }  \
  } while (0)
struct RcclGather { ncclComm_t comm; int nranks, rank; long long per; c64* stage; int device; };
// The ranks must agree on failure: every rank's block carries one extra entry, its device-wide "a kernel abandoned a wait"
// word at the time of the exchange; after the gather every rank raises its own word if ANY rank's was set, so the Krylov driver of every
// rank returns MA_ERR_HIP at its end instead of one rank leaving while the others wait in the next collective.
__global__ void rccl_status_out_kernel(const unsigned* __restrict__ spin_err, c64* __restrict__ slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { slot->re = (spin_err && *spin_err != 0u) ? 1.0 : 0.0; slot->im = 0.0; }
}
__global__ void rccl_status_in_kernel(const c64* __restrict__ stage, long long stride, int nranks, unsigned* __restrict__ spin_err) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && spin_err) {
    bool any = false;
    for (int r = 0; r < nranks; ++r) any = any || stage[(long long)r * stride + (stride - 1)].re != 0.0;
    if (any) *spin_err = 1u;
  }
}
void rccl_gather_free(void* u) {
  RcclGather* g = (RcclGather*)u;
  if (g->stage) { (void)hipSetDevice(g->device); (void)hipFree(g->stage); }
  delete g;
}
// ma_gather_fn: this rank's rows of y -> its block of the staging vector (equal blocks of `per` rows, the last one padded),
// ncclAllGather in place, the first n entries back into y; all on `stream`
int rccl_gather_cb(void* user, void* d_y, int64_t n, int64_t row0, int64_t row1, void* stream) {
  RcclGather* g = (RcclGather*)user;
  RcclApi* R = rccl_api();
  hipStream_t st = (hipStream_t)stream;
  const size_t stride = (size_t)g->per + 1;                 // a rank's block: `per` rows + its status entry
  c64* y = (c64*)d_y; c64* mine = g->stage + (size_t)g->rank * stride;
  if (row1 - row0 < g->per && hipMemsetAsync(mine, 0, sizeof(c64) * (size_t)g->per, st) != hipSuccess) return -2;
  if (row1 > row0 && hipMemcpyAsync(mine, y + row0, sizeof(c64) * (size_t)(row1 - row0), hipMemcpyDeviceToDevice, st) != hipSuccess) return -3;
  unsigned* werr = spin_error_word();
  hipLaunchKernelGGL(rccl_status_out_kernel, dim3(1), dim3(64), 0, st, werr, mine + g->per);
  const ncclResult_t r = R->AllGather(mine, g->stage, 2 * stride, ncclDouble, g->comm, st);
  if (r != ncclSuccess) { set_error("ncclAllGather failed: %s", R->GetErrorString(r)); return -4; }
  hipLaunchKernelGGL(rccl_status_in_kernel, dim3(1), dim3(64), 0, st, g->stage, (long long)stride, g->nranks, werr);
  // the ranks' rows back into y: block r holds rows [r per, min(n, (r + 1) per))
  for (int q = 0; q < g->nranks; ++q) {
    const long long a = (long long)q * g->per, b = std::min<long long>(n, a + g->per);
    if (b > a && hipMemcpyAsync(y + a, g->stage + (size_t)q * stride, sizeof(c64) * (size_t)(b - a), hipMemcpyDeviceToDevice, st) != hipSuccess) return -5;
  }
  if (hipGetLastError() != hipSuccess) return -6;
  return 0;
}
}  // namespace

int ma_rccl_get_unique_id(void* id128) {
  MA_REQUIRE(id128, MA_ERR_INVALID, "NULL argument");
  RcclApi* R = rccl_api();
  { const char* why = R->ok ? nullptr : dlerror();           // (one call: dlerror() clears its state when read)
    MA_REQUIRE(R->ok, MA_ERR_UNSUPPORTED, "librccl could not be loaded (%s)", why ? why : "symbols missing"); }
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  MA_RCCL(R->GetUniqueId((ncclUniqueId*)id128));
  return MA_OK;
}
int ma_rccl_comm_create(int32_t nranks, int32_t rank, const void* id128, int device, void** comm) {
  MA_REQUIRE(comm && id128 && nranks >= 1 && rank >= 0 && rank < nranks, MA_ERR_INVALID, "bad argument");
  *comm = nullptr;
  RcclApi* R = rccl_api();
  MA_REQUIRE(R->ok, MA_ERR_UNSUPPORTED, "librccl could not be loaded");
  int rc = use_device(device);
  if (rc) return rc;
  ncclUniqueId id; memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  MA_RCCL(R->CommInitRank(&c, nranks, id, rank));
  *comm = (void*)c;
  return MA_OK;
}
int ma_rccl_comm_destroy(void* comm) {
  if (!comm) return MA_OK;
  RcclApi* R = rccl_api();
  MA_REQUIRE(R->ok, MA_ERR_UNSUPPORTED, "librccl could not be loaded");
  MA_RCCL(R->CommDestroy((ncclComm_t)comm));
  return MA_OK;
}
// ma_op_create_gathered with the exchange inside the library: rank `rank` of `nranks` holds rows [rank * per, min(n, (rank + 1) * per)),
// per = ceil(n / nranks) (the row_block rule of sharded.py and of ma_op_create_tbem_multi), and every apply ends with one
// ncclAllGather of per rows per rank on the apply's stream. `inner` is not owned; the communicator stays the caller's.
int ma_op_create_gathered_rccl(ma_op_t* inner, void* nccl_comm, int32_t nranks, int32_t rank, ma_op_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(inner && nccl_comm && nranks >= 1 && rank >= 0 && rank < nranks, MA_ERR_INVALID, "bad argument");
  RcclApi* R = rccl_api();
  MA_REQUIRE(R->ok, MA_ERR_UNSUPPORTED, "librccl could not be loaded");
  const long long n = inner->n, per = (n + nranks - 1) / nranks;
  const long long row0 = std::min(n, (long long)rank * per), row1 = std::min(n, row0 + per);
  MA_REQUIRE(inner->kind != 2 || (inner->row0 == row0 && inner->row1 == row1), MA_ERR_INVALID,
             "rank %d of %d owns rows [%lld, %lld); the inner operator was created for [%d, %d)", rank, nranks, row0, row1, inner->row0, inner->row1);
  RcclGather* g = new (std::nothrow) RcclGather{(ncclComm_t)nccl_comm, nranks, rank, per, nullptr, inner->device};
  MA_REQUIRE(g, MA_ERR_NOMEM, "host allocation failed");
  int prev = -1;
  const bool had = hipGetDevice(&prev) == hipSuccess;        // the caller's current device is left as it was
  (void)hipSetDevice(inner->device);
  int rc = MA_OK;
  if (hipMalloc(&g->stage, sizeof(c64) * ((size_t)per + 1) * (size_t)nranks) != hipSuccess) { delete g; g = nullptr; set_error("rank-sharded operator: the gather's staging vector does not fit"); rc = MA_ERR_NOMEM; }
  if (!rc) {
    rc = ma_op_create_gathered(inner, row0, row1, rccl_gather_cb, g, out);
    if (rc) rccl_gather_free(g); else (*out)->gather_free = rccl_gather_free;
  }
  if (had) (void)hipSetDevice(prev);
  return rc;
}

#ifdef MA_DIAGNOSTICS
// Diagnostic build only: `nranks` loopback communicators (virtual ranks on `device`, see loop_allgather) in comms[0..nranks), and the
// loopback collective in place of ncclAllGather for every communicator of the process from here on; _poison: rank `rank`'s status entry
// reads "a wait was abandoned" in every later exchange (-1: none); _destroy frees the group.
int ma_rccl_test_loopback_create(int32_t nranks, int device, void** comms) {
  MA_REQUIRE(comms && nranks >= 1 && nranks <= 64, MA_ERR_INVALID, "bad argument");
  RcclApi* R = rccl_api();
  MA_REQUIRE(R->ok, MA_ERR_UNSUPPORTED, "librccl could not be loaded");
  LoopGroup* g = new (std::nothrow) LoopGroup();
  MA_REQUIRE(g, MA_ERR_NOMEM, "host allocation failed");
  g->nranks = nranks; g->device = device;
  for (int r = 0; r < nranks; ++r) comms[r] = new LoopComm{0x4c4f4f50u, g, r};
  R->AllGather = loop_allgather;
  return MA_OK;
}
int ma_rccl_test_loopback_poison(void* comm, int32_t rank) {
  LoopComm* c = (LoopComm*)comm;
  MA_REQUIRE(c && c->magic == 0x4c4f4f50u, MA_ERR_INVALID, "not a loopback communicator");
  c->g->poison_rank = rank;
  return MA_OK;
}
int ma_rccl_test_loopback_destroy(void** comms, int32_t nranks) {
  MA_REQUIRE(comms && nranks >= 1, MA_ERR_INVALID, "bad argument");
  LoopGroup* g = ((LoopComm*)comms[0])->g;
  if (g->board) { (void)hipSetDevice(g->device); (void)hipFree(g->board); }
  for (int r = 0; r < nranks; ++r) delete (LoopComm*)comms[r];
  delete g;
  return MA_OK;
}
#endif

int ma_op_destroy(ma_op_t* o) {
  if (!o) return MA_OK;
  (void)hipSetDevice(o->device);
  op_free(o);
  delete o;
  return MA_OK;
}
int ma_op_num_rows(const ma_op_t* o, int64_t* n) {
  MA_REQUIRE(o && n, MA_ERR_INVALID, "NULL argument");
  *n = o->n; return MA_OK;
}

// mode 0: y = A x (every shard writes its own slice of y); 1 / 2: y = A^T x / A^H x (every shard's rows contribute to all of y:
// the contributions are gathered on the home device and summed there, in shard order, so that the result does not depend on
// the timing of the copies). x and y live on the home device; `st` is the caller's stream there.
static int op_apply_sharded(ma_op* o, const c64* d_x, c64* d_y, int mode, hipStream_t st) {
  const size_t n = (size_t)o->n, bytes = sizeof(c64) * n;
  const int home = o->device, G = (int)o->shards.size();
  MA_HIP(hipSetDevice(home));
  MA_HIP(hipEventRecord(o->ev_home, st));
  if (mode != 0 && !o->d_tgather) MA_HIP(hipMalloc(&o->d_tgather, bytes * (size_t)G));
  int rc = MA_OK;
  for (int g = 0; g < G && !rc; ++g) {
    OpShard& sh = o->shards[(size_t)g];
    MA_HIP(hipSetDevice(sh.device));
    MA_HIP(hipStreamWaitEvent(sh.st, o->ev_home, 0));
    const c64* xg = d_x;
    const bool remote = g > 0;                           // shard 0 is the home shard: it works on the caller's vectors
    if (remote) { MA_HIP(hipMemcpyPeerAsync(sh.d_x, sh.device, d_x, home, bytes, sh.st)); xg = sh.d_x; }
    if (mode == 0) {
      c64* yg = remote ? sh.d_y : d_y;                   // the home shard writes straight into the caller's y
      rc = ma_op_apply_dev(sh.op, xg, yg, sh.st);
      if (!rc && remote)
        MA_HIP(hipMemcpyPeerAsync(d_y + sh.row0, home, sh.d_y + sh.row0, sh.device, sizeof(c64) * (size_t)(sh.row1 - sh.row0), sh.st));
    } else {
      rc = mode == 1 ? ma_op_apply_transpose_dev(sh.op, xg, sh.d_y, sh.st) : ma_op_apply_hermitian_dev(sh.op, xg, sh.d_y, sh.st);
      if (!rc) MA_HIP(hipMemcpyPeerAsync(o->d_tgather + (size_t)g * n, home, sh.d_y, sh.device, bytes, sh.st));
    }
    if (!rc) MA_HIP(hipEventRecord(sh.done, sh.st));
  }
  MA_HIP(hipSetDevice(home));
  if (rc) return rc;
  for (int g = 0; g < G; ++g) MA_HIP(hipStreamWaitEvent(st, o->shards[(size_t)g].done, 0));
  if (mode != 0) {
    MA_HIP(hipMemcpyAsync(d_y, o->d_tgather, bytes, hipMemcpyDeviceToDevice, st));
    for (int g = 1; g < G && !rc; ++g) rc = op_launch_axpby((long long)n, 1.0, 0.0, d_y, 1.0, 0.0, o->d_tgather + (size_t)g * n, d_y, st);
  }
  return rc;
}

// LinearOperator::apply: y = A x, device pointers
int ma_op_apply_dev(ma_op_t* o, const void* d_x, void* d_y, void* stream) {
  MA_REQUIRE(o && d_x && d_y, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(o->device));
  hipStream_t st = (hipStream_t)stream;
  if (o->kind == 8) {                                    // this rank's rows, then the exchange that brings the other ranks' rows
    int rc = ma_op_apply_dev(o->inner, d_x, d_y, stream);
    if (rc) return rc;
    rc = o->gather(o->gather_user, d_y, o->n, o->row0, o->row1, stream);
    MA_REQUIRE(rc == 0, MA_ERR_HIP, "the row exchange of a rank-sharded operator failed (callback returned %d)", rc);
    return MA_OK;
  }
  if (o->kind == 3) return op_apply_sharded(o, (const c64*)d_x, (c64*)d_y, 0, st);
  if (o->kind == 4) return slfmm_apply(o->fmm, (const c64*)d_x, (c64*)d_y, 0, st);
  if (o->kind == 6) return mlfmm_apply(o->mlfmm, (const c64*)d_x, (c64*)d_y, st);
  if (o->kind == 0) return op_launch_zgemv(o->n, o->dA, (const c64*)d_x, (c64*)d_y, st);
  if (o->kind == 1) return ma_csr_spmv_dev(o->csr, d_x, d_y, stream);
  return op_launch_tbem_matvec(o->plan->geom, o->ph, o->row0, o->row1, o->nchunks, (const c64*)d_x, o->d_partial, o->plan->d_pair_off,
                               o->plan->d_pairs, o->d_corr, o->d_diag, (c64*)d_y, st);
}
// host buffers (apply(&self, x: &Array1<T>) -> Array1<T>, traits.rs:324)
int ma_op_apply(ma_op_t* o, const ma_c64* x, ma_c64* y) {
  MA_REQUIRE(o && x && y, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(o->device));
  MA_HIP(hipMemcpy(o->d_x, x, sizeof(c64) * (size_t)o->n, hipMemcpyHostToDevice));
  if (o->kind == 2 && (o->row0 != 0 || o->row1 != o->plan->np)) MA_HIP(hipMemset(o->d_y, 0, sizeof(c64) * (size_t)o->n));
  int rc = ma_op_apply_dev(o, o->d_x, o->d_y, nullptr);
  if (rc) return rc;
  MA_HIP(hipMemcpy(y, o->d_y, sizeof(c64) * (size_t)o->n, hipMemcpyDeviceToHost));
  return MA_OK;
}

// LinearOperator::apply_transpose (y = A^T x) and apply_hermitian (y = A^H x = conj(A^T conj(x)), traits.rs:326-358),
// device pointers (x and y distinct). Dense: DenseOperator's matrix.t().dot(x) (fmm_interface.rs:40-48); CSR: an SpMV on
// the transposed operator, built once. Matrix-free TBEM: the streamed kernel with the loop nest turned around.
// column-sorted view of the plan's near-pair list (pairs are stored by row): one counting sort on the host, once per operator
static int tbem_build_column_index(ma_op* o) {
  if (o->d_t_off) return MA_OK;
  const ma_bem_plan* P = o->plan;
  const long long np = P->np, npairs = P->npairs;
  std::vector<int2> hp((size_t)std::max<long long>(npairs, 1));
  if (npairs > 0) MA_HIP(hipMemcpy(hp.data(), P->d_pairs, sizeof(int2) * (size_t)npairs, hipMemcpyDeviceToHost));
  std::vector<long long> off((size_t)np + 1, 0);
  for (long long q = 0; q < npairs; ++q) off[(size_t)hp[(size_t)q].y + 1]++;
  for (long long j = 0; j < np; ++j) off[(size_t)j + 1] += off[(size_t)j];
  std::vector<long long> cur(off.begin(), off.end() - 1);
  std::vector<int> idx((size_t)std::max<long long>(npairs, 1));
  for (long long q = 0; q < npairs; ++q) idx[(size_t)cur[(size_t)hp[(size_t)q].y]++] = (int)q;     // ascending q = ascending row inside a column
  MA_HIP(hipMalloc(&o->d_t_off, sizeof(long long) * ((size_t)np + 1)));
  MA_HIP(hipMalloc(&o->d_t_idx, sizeof(int) * idx.size()));
  MA_HIP(hipMalloc(&o->d_tpartial, sizeof(c64) * (size_t)o->nchunks * (size_t)np));
  MA_HIP(hipMemcpy(o->d_t_off, off.data(), sizeof(long long) * ((size_t)np + 1), hipMemcpyHostToDevice));
  MA_HIP(hipMemcpy(o->d_t_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice));
  return MA_OK;
}

static int op_apply_t(ma_op_t* o, const void* d_x, void* d_y, bool herm, hipStream_t st) {
  MA_REQUIRE(o && d_x && d_y, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(o->device));
  if (o->kind == 3) return op_apply_sharded(o, (const c64*)d_x, (c64*)d_y, herm ? 2 : 1, st);
  MA_REQUIRE(o->kind != 8, MA_ERR_UNSUPPORTED, "the transposed apply of a rank-sharded operator needs a reduction over the ranks: not defined for it");
  MA_REQUIRE(o->kind != 6, MA_ERR_UNSUPPORTED, "MLFMM transpose not yet implemented (the reference's MlfmmOperator::apply_transpose is unimplemented!(), fmm_interface.rs:131-134)");
  if (o->kind == 4) {                                    // matvec_transpose (slfmm.rs:262-376); hermitian = conj(A^T conj(x)) (traits.rs:340-358)
    const c64* xin = (const c64*)d_x;
    int rc = MA_OK;
    if (herm) {
      if (!o->d_cx) MA_HIP(hipMalloc(&o->d_cx, sizeof(c64) * (size_t)o->n));
      rc = op_launch_conj(o->n, xin, o->d_cx, st); if (rc) return rc;
      xin = o->d_cx;
    }
    rc = slfmm_apply(o->fmm, xin, (c64*)d_y, 1, st);
    if (!rc && herm) rc = op_launch_conj(o->n, (const c64*)d_y, (c64*)d_y, st);
    return rc;
  }
  if (o->kind == 2) {
    // streamed like apply, with the loop nest turned around (lane = field panel). A row-sharded operator returns its
    // rows' contribution to every entry of y: the shards' results add up (the caller's all-reduce).
    MA_REQUIRE(o->plan->npairs < 2147483647LL, MA_ERR_UNSUPPORTED, "near-pair list too long for the column index");
    int rc = tbem_build_column_index(o); if (rc) return rc;
    const c64* xin = (const c64*)d_x;
    if (herm) {
      if (!o->d_cx) MA_HIP(hipMalloc(&o->d_cx, sizeof(c64) * (size_t)o->n));
      rc = op_launch_conj(o->n, xin, o->d_cx, st); if (rc) return rc;
      xin = o->d_cx;
    }
    rc = op_launch_tbem_matvec_t(o->plan->geom, o->ph, o->row0, o->row1, o->nchunks, xin, o->d_tpartial, o->d_t_off, o->d_t_idx,
                                 o->plan->d_pairs, o->d_corr, o->d_diag, (c64*)d_y, st);
    if (!rc && herm) rc = op_launch_conj(o->n, (const c64*)d_y, (c64*)d_y, st);
    return rc;
  }
  if (o->kind == 0) {
    if (!o->d_tpart) MA_HIP(hipMalloc(&o->d_tpart, sizeof(c64) * (size_t)op_zgemv_t_chunks() * (size_t)o->n));
    return op_launch_zgemv_t(o->n, o->dA, (const c64*)d_x, o->d_tpart, (c64*)d_y, herm, st);
  }
  if (o->csr_t && o->csr_t_epoch != ma_csr_epoch(o->csr)) {          // the source was re-assembled for another frequency
    int rebuild = 1;
    int rc = ma_csr_refresh_transpose(o->csr, o->csr_t, &rebuild); if (rc) return rc;
    if (rebuild) { (void)ma_csr_destroy(o->csr_t); o->csr_t = nullptr; }
    o->csr_t_epoch = ma_csr_epoch(o->csr);
  }
  if (!o->csr_t) { int rc = ma_csr_transpose(o->csr, &o->csr_t); if (rc) return rc; o->csr_t_epoch = ma_csr_epoch(o->csr); }
  if (!herm) return ma_csr_spmv_dev(o->csr_t, d_x, d_y, st);
  if (!o->d_cx) MA_HIP(hipMalloc(&o->d_cx, sizeof(c64) * (size_t)o->n));
  int rc = op_launch_conj(o->n, (const c64*)d_x, o->d_cx, st);
  if (!rc) rc = ma_csr_spmv_dev(o->csr_t, o->d_cx, d_y, st);
  if (!rc) rc = op_launch_conj(o->n, (const c64*)d_y, (c64*)d_y, st);
  return rc;
}
int ma_op_apply_transpose_dev(ma_op_t* o, const void* d_x, void* d_y, void* stream) { return op_apply_t(o, d_x, d_y, false, (hipStream_t)stream); }
int ma_op_apply_hermitian_dev(ma_op_t* o, const void* d_x, void* d_y, void* stream) { return op_apply_t(o, d_x, d_y, true, (hipStream_t)stream); }
static int op_apply_t_host(ma_op_t* o, const ma_c64* x, ma_c64* y, bool herm) {
  MA_REQUIRE(o && x && y, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(o->device));
  MA_HIP(hipMemcpy(o->d_x, x, sizeof(c64) * (size_t)o->n, hipMemcpyHostToDevice));
  int rc = op_apply_t(o, o->d_x, o->d_y, herm, nullptr);
  if (rc) return rc;
  MA_HIP(hipMemcpy(y, o->d_y, sizeof(c64) * (size_t)o->n, hipMemcpyDeviceToHost));
  return MA_OK;
}
int ma_op_apply_transpose(ma_op_t* o, const ma_c64* x, ma_c64* y) { return op_apply_t_host(o, x, y, false); }
int ma_op_apply_hermitian(ma_op_t* o, const ma_c64* x, ma_c64* y) { return op_apply_t_host(o, x, y, true); }

// ------------------------------------------------------------------ Preconditioner boundary (traits.rs:370-375)
// apply(r) -> z. The device preconditioners are the one-level smoothers of the AMG V-cycle applied from z = 0
// (what AmgPreconditioner::apply does on its coarsest level, amg.rs:981-1005): `sweeps` Jacobi (kind 1) or
// l1-Jacobi (kind 2) sweeps on A z = r with the CSR handle's current values. kind 0 is the identity.
// one level of an AMG hierarchy on the device (AmgLevel, amg.rs:229-249): the level's operator, its transfer operators (null on
// the coarsest level) and the level's work vectors
struct AmgLevelDev {
  ma_csr* A = nullptr; ma_csr* P = nullptr; ma_csr* R = nullptr; long long n = 0;
  c64* x = nullptr; c64* b = nullptr; c64* r = nullptr; c64* tmp = nullptr;     // x / b: this level's unknown and right-hand side (levels > 0)
};
struct ma_precond {
  int kind = 0; ma_csr* csr = nullptr; double omega = 2.0 / 3.0; int sweeps = 2; long long n = 0; int device = 0;
  c64* d_tmp = nullptr;
  c64* d_invdiag = nullptr;      // kind 4: 1 / a_ii of an operator (DiagonalPreconditioner::from_diagonal)
  // kind 5: AmgPreconditioner::apply (amg.rs:1068-1103)
  std::vector<AmgLevelDev> lv; int amg_smoother = 0, amg_pre = 1, amg_post = 1, amg_cycle = 0;
  std::vector<ma_csr*> amg_owned; double amg_gc = 1.0, amg_oc = 1.0, amg_setup_ms = 0.0;   // from_csr: the levels it built, amg.rs:375-392
  // kind 6: IluPreconditioner (ilu.rs): L (strictly lower, unit diagonal implied) and U (diagonal + upper) as operators of their own
  ma_csr* ilu_l = nullptr; ma_csr* ilu_u = nullptr;
  // kind 7: AdditiveSchwarzPreconditioner (schwarz.rs): restriction E onto the stacked subdomains, ILU(0) of their block-diagonal matrix,
  // weighted prolongation back; sch_stats = subdomains, min / max size, total size
  ma_precond* sch_inner = nullptr; ma_csr* sch_E = nullptr; ma_csr* sch_Et = nullptr; c64* sch_a = nullptr; c64* sch_b = nullptr; long long sch_stats[4] = {0, 0, 0, 0};
};

extern "C" int ma_csr_jacobi_dev(ma_csr* h, void* d_x, const void* d_b, double omega, int sweeps, void* d_tmp, void* stream);
extern "C" int ma_csr_l1jacobi_dev(ma_csr* h, void* d_x, const void* d_b, int sweeps, void* d_tmp, void* stream);
extern "C" int ma_csr_sym_gauss_seidel_dev(ma_csr* h, void* d_x, const void* d_b, int sweeps, void* stream);
extern "C" int ma_csr_residual_dev(ma_csr* h, const void* d_x, const void* d_b, void* d_r, void* stream);
extern "C" int ma_csr_num_cols(const ma_csr* h, int64_t* ncols);
extern "C" int ma_csr_device(const ma_csr* h, int* device);
extern "C" int ma_csr_get(ma_csr* h, int64_t* row_ptrs, int64_t* col_indices, ma_c64* values);
extern "C" int ma_csr_gauss_seidel_sweep_dev(ma_csr* h, void* d_x, const void* d_b, int mode, int backward, void* stream);

// AmgPreconditioner::v_cycle (amg.rs:981-1065) on device vectors of level `level`: smoothers are the AMG sweeps of the CSR
// handles (smooth_jacobi :855-884, smooth_l1_jacobi :887-929, smooth_sym_gauss_seidel :932-978), the coarsest level runs 20
// Jacobi / 20 l1-Jacobi / 10 symmetric Gauss-Seidel sweeps (:986-1003), restriction and prolongation are SpMVs with the level's R and P.
extern "C" int ma_csr_jacobi_from_zero_dev(ma_csr* h, void* d_x, const void* d_b, double omega, int sweeps, void* d_tmp, int l1, void* stream);
extern "C" int ma_csr_spmv_add_dev(ma_csr* h, const void* d_e, void* d_x_inout, void* stream);
// x_is_zero: the iterate is zero by construction and its array has not been cleared (Jacobi-type smoothers write the first sweep outright)
static int amg_smooth(ma_precond* M, AmgLevelDev& L, c64* x, const c64* b, int sweeps, hipStream_t st, bool x_is_zero = false) {
  if (sweeps <= 0) return MA_OK;
  if (x_is_zero && M->amg_smoother != 2) return ma_csr_jacobi_from_zero_dev(L.A, x, b, M->omega, sweeps, L.tmp, M->amg_smoother == 1 ? 1 : 0, st);
  if (M->amg_smoother == 1) return ma_csr_l1jacobi_dev(L.A, x, b, sweeps, L.tmp, st);
  if (M->amg_smoother == 2) return ma_csr_sym_gauss_seidel_dev(L.A, x, b, sweeps, st);
  return ma_csr_jacobi_dev(L.A, x, b, M->omega, sweeps, L.tmp, st);
}
// the sweeps a level runs first on its (zero) iterate: when they are Jacobi-type and there is at least one, the caller may leave the
// iterate uncleared
static bool amg_first_sweep_writes(const ma_precond* M, size_t level) {
  const bool coarsest = level + 1 == M->lv.size() || !M->lv[level].P;
  return M->amg_smoother != 2 && (coarsest ? 20 : M->amg_pre) >= 1;
}
static int amg_v_cycle(ma_precond* M, size_t level, c64* x, const c64* b, hipStream_t st, bool x_is_zero = false) {
  AmgLevelDev& L = M->lv[level];
  if (level + 1 == M->lv.size() || !L.P)
    return amg_smooth(M, L, x, b, M->amg_smoother == 2 ? 10 : 20, st, x_is_zero);
  int rc = amg_smooth(M, L, x, b, M->amg_pre, st, x_is_zero);
  if (!rc) rc = ma_csr_residual_dev(L.A, x, b, L.r, st);                       // r = b - A x
  AmgLevelDev& C = M->lv[level + 1];
  if (!rc) rc = ma_csr_spmv_dev(L.R, L.r, C.b, st);                            // r_c = R r
  const bool lazy = amg_first_sweep_writes(M, level + 1);                     // the coarse iterate starts at zero: cleared, or written by its first sweep
  if (!rc && !lazy && hipMemsetAsync(C.x, 0, sizeof(c64) * (size_t)C.n, st) != hipSuccess) { set_error("AMG: clearing the coarse correction failed"); rc = MA_ERR_HIP; }
  if (!rc) rc = amg_v_cycle(M, level + 1, C.x, C.b, st, lazy);
  if (!rc) rc = ma_csr_spmv_add_dev(L.P, C.x, x, st);                          // x = x + P e_c in one pass
  if (!rc) rc = amg_smooth(M, L, x, b, M->amg_post, st);
  return rc;
}
int ma_precond_create_jacobi(ma_csr_t* csr, double omega, int32_t sweeps, ma_precond_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(csr && sweeps >= 0, MA_ERR_INVALID, "bad argument");
  int64_t n = 0, nnz = 0; int rc = ma_csr_num_rows(csr, &n, &nnz); if (rc) return rc;
  ma_precond* M = new (std::nothrow) ma_precond(); MA_REQUIRE(M, MA_ERR_NOMEM, "host allocation failed");
  M->kind = 1; M->csr = csr; M->omega = omega; M->sweeps = sweeps; M->n = n;
  MA_HIP(hipGetDevice(&M->device));
  hipError_t e = hipMalloc(&M->d_tmp, sizeof(c64) * (size_t)n);
  if (e != hipSuccess) { set_error("preconditioner workspace: %s", hipGetErrorString(e)); delete M; return MA_ERR_NOMEM; }
  *out = M; return MA_OK;
}
int ma_precond_create_l1jacobi(ma_csr_t* csr, int32_t sweeps, ma_precond_t** out) {
  int rc = ma_precond_create_jacobi(csr, 1.0, sweeps, out);
  if (!rc) (*out)->kind = 2;
  return rc;
}
int ma_precond_create_sym_gauss_seidel(ma_csr_t* csr, int32_t sweeps, ma_precond_t** out) {
  int rc = ma_precond_create_jacobi(csr, 1.0, sweeps, out);
  if (!rc) (*out)->kind = 3;
  return rc;
}
// DiagonalPreconditioner::from_diagonal(diag of the operator) (math-bem/src/core/solver/fmm_interface.rs:177-212): z_i = r_i / a_ii
// (z_i = r_i where |a_ii| <= 1e-15). Dense: the matrix diagonal; CSR: jacobi(omega = 1, one sweep); matrix-free TBEM: the
// self terms (singular integration + free term), i.e. the true diagonal of the operator.
int ma_precond_create_diagonal(ma_op_t* op, ma_precond_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(op, MA_ERR_INVALID, "operator is NULL");
  if (op->kind == 8) return ma_precond_create_diagonal(op->inner, out);             // every rank's plan holds every panel's self term
  if (op->kind == 1) return ma_precond_create_jacobi(op->csr, 1.0, 1, out);
  if (op->kind == 3) return ma_precond_create_diagonal(op->shards[0].op, out);    // the home shard's plan holds every panel
  MA_REQUIRE(op->kind != 6, MA_ERR_UNSUPPORTED, "diagonal preconditioner of the multi-level operator: the reference defines it for the single-level system only");
  MA_HIP(hipSetDevice(op->device));
  ma_precond* M = new (std::nothrow) ma_precond(); MA_REQUIRE(M, MA_ERR_NOMEM, "host allocation failed");
  M->kind = 4; M->n = op->n; M->device = op->device;
  if (hipMalloc(&M->d_invdiag, sizeof(c64) * (size_t)op->n) != hipSuccess) { delete M; set_error("diagonal preconditioner: out of device memory"); return MA_ERR_NOMEM; }
  int rc = MA_OK;
  if (op->kind == 0) rc = op_launch_diag_invert(op->n, op->dA, op->n + 1, nullptr, M->d_invdiag, nullptr);
  else if (op->kind == 4) {                     // SparseNearfieldIlu::from_slfmm (fmm_interface.rs:249-297): the diagonal of the self blocks, 1 where its norm <= 1e-15
    c64* d = nullptr;
    if (hipMalloc(&d, sizeof(c64) * (size_t)op->n) != hipSuccess) { set_error("diagonal preconditioner: out of device memory"); rc = MA_ERR_NOMEM; }
    if (!rc) rc = slfmm_self_diagonal(op->fmm, d, nullptr);
    if (!rc) rc = op_launch_diag_invert(op->n, d, 1, nullptr, M->d_invdiag, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) { set_error("diagonal preconditioner: kernels failed"); rc = MA_ERR_HIP; }
    if (d) (void)hipFree(d);
  } else {
    const ma_bem_plan* P = op->plan;
    c64* d = nullptr;
    if (hipMalloc(&d, sizeof(c64) * (size_t)P->np) != hipSuccess) { set_error("diagonal preconditioner: out of device memory"); rc = MA_ERR_NOMEM; }
    if (!rc) rc = bem_launch_self_list_values(P->geom, op->ph, d, nullptr);
    if (!rc) rc = op_launch_diag_invert(P->np, d, 1, P->geom.dof, M->d_invdiag, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) { set_error("diagonal preconditioner: kernels failed"); rc = MA_ERR_HIP; }
    if (d) (void)hipFree(d);
  }
  if (rc) { (void)hipFree(M->d_invdiag); delete M; return rc; }
  *out = M; return MA_OK;
}
// AmgPreconditioner over a hierarchy built on the host (AmgPreconditioner::from_csr keeps the setup -- strength, coarsening,
// interpolation, Galerkin products, amg.rs:276-372 -- where it is): level l brings its operator A_l and, except the
// coarsest, the prolongation P_l (n_l x n_{l+1}) and restriction R_l (n_{l+1} x n_l) as CSR handles (borrowed).
// smoother: 0 Jacobi(omega) (AmgSmoother::Jacobi / Chebyshev), 1 l1-Jacobi, 2 symmetric Gauss-Seidel; cycle: 0 V, 1 W, 2 F.
int ma_precond_create_amg(int32_t nlevels, ma_csr_t* const* A, ma_csr_t* const* P, ma_csr_t* const* R, int32_t smoother, double jacobi_weight,
                          int32_t num_pre_smooth, int32_t num_post_smooth, int32_t cycle, ma_precond_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(nlevels >= 1 && nlevels <= 64 && A, MA_ERR_INVALID, "bad level count");
  MA_REQUIRE(nlevels == 1 || (P && R), MA_ERR_INVALID, "transfer operators missing");
  MA_REQUIRE(smoother >= 0 && smoother <= 2 && cycle >= 0 && cycle <= 2 && num_pre_smooth >= 0 && num_post_smooth >= 0, MA_ERR_INVALID, "bad smoother / cycle");
  ma_precond* M = new (std::nothrow) ma_precond(); MA_REQUIRE(M, MA_ERR_NOMEM, "host allocation failed");
  M->kind = 5; M->amg_smoother = smoother; M->omega = jacobi_weight; M->amg_pre = num_pre_smooth; M->amg_post = num_post_smooth; M->amg_cycle = cycle;
  M->lv.resize((size_t)nlevels);
  int rc = MA_OK;
  MA_HIP(hipGetDevice(&M->device));
  for (int l = 0; l < nlevels && !rc; ++l) {
    AmgLevelDev& L = M->lv[(size_t)l];
    int64_t n = 0, nnz = 0, nc = 0;
    if (!A[l]) { set_error("level %d has no operator", l); rc = MA_ERR_INVALID; break; }
    rc = ma_csr_num_rows(A[l], &n, &nnz); if (rc) break;
    rc = ma_csr_num_cols(A[l], &nc); if (rc) break;
    if (nc != n) { set_error("level %d: operator is %lld x %lld", l, (long long)n, (long long)nc); rc = MA_ERR_DIM; break; }
    L.A = A[l]; L.n = n;
    if (l + 1 < nlevels) {
      if (!P[l] || !R[l]) { set_error("level %d lacks P or R", l); rc = MA_ERR_INVALID; break; }
      int64_t pr = 0, pc = 0, rr = 0, rcn = 0, cn = 0, z = 0;
      (void)ma_csr_num_rows(P[l], &pr, &z); (void)ma_csr_num_cols(P[l], &pc); (void)ma_csr_num_rows(R[l], &rr, &z); (void)ma_csr_num_cols(R[l], &rcn);
      if (A[l + 1]) (void)ma_csr_num_rows(A[l + 1], &cn, &z);
      if (pr != n || rcn != n || pc != cn || rr != cn) { set_error("level %d: P is %lld x %lld, R is %lld x %lld, levels have %lld and %lld rows", l, (long long)pr, (long long)pc, (long long)rr, (long long)rcn, (long long)n, (long long)cn); rc = MA_ERR_DIM; break; }
      L.P = P[l]; L.R = R[l];
    }
    hipError_t e = hipMalloc(&L.r, sizeof(c64) * (size_t)n);
    if (e == hipSuccess) e = hipMalloc(&L.tmp, sizeof(c64) * (size_t)n);
    if (e == hipSuccess && l > 0) e = hipMalloc(&L.x, sizeof(c64) * (size_t)n);
    if (e == hipSuccess && l > 0) e = hipMalloc(&L.b, sizeof(c64) * (size_t)n);
    if (e != hipSuccess) { set_error("AMG level %d vectors: %s", l, hipGetErrorString(e)); rc = MA_ERR_NOMEM; }
  }
  if (!rc) { M->n = M->lv[0].n; if (hipMalloc(&M->d_tmp, sizeof(c64) * (size_t)M->n * 2) != hipSuccess) { set_error("AMG workspace"); rc = MA_ERR_NOMEM; } }
  if (rc) { ma_precond_destroy(M); return rc; }
  *out = M; return MA_OK;
}
int ma_precond_destroy(ma_precond_t* M) {
  if (!M) return MA_OK;
  for (AmgLevelDev& L : M->lv) { void* p[] = {L.x, L.b, L.r, L.tmp}; for (void* q : p) if (q) (void)hipFree(q); }
  if (M->d_tmp) (void)hipFree(M->d_tmp);
  if (M->d_invdiag) (void)hipFree(M->d_invdiag);
  if (M->ilu_l) (void)ma_csr_destroy(M->ilu_l);
  if (M->ilu_u) (void)ma_csr_destroy(M->ilu_u);
  for (ma_csr* h : M->amg_owned) if (h) (void)ma_csr_destroy(h);
  if (M->sch_inner) (void)ma_precond_destroy(M->sch_inner);
  if (M->sch_E) (void)ma_csr_destroy(M->sch_E);
  if (M->sch_Et) (void)ma_csr_destroy(M->sch_Et);
  if (M->sch_a) (void)hipFree(M->sch_a);
  if (M->sch_b) (void)hipFree(M->sch_b);
  delete M; return MA_OK;
}
// AmgPreconditioner::from_csr (amg.rs:276-372): the matrix' current values come back to the host, the hierarchy is built there with
// the reference's steps (amg_setup.hip), every level goes up as an operator of its own and the device cycle of
// ma_precond_create_amg runs over them.
extern "C" int ma_csr_create(int64_t n, const int64_t* row_ptrs, const int64_t* col_indices, const ma_c64* values, int device, ma_csr** out);
extern "C" int ma_csr_create_rect(int64_t nrows, int64_t ncols, const int64_t* row_ptrs, const int64_t* col_indices, const ma_c64* values, int device, ma_csr** out);
int ma_precond_create_amg_from_csr(ma_csr_t* A, const ma_amg_config_t* cfg, ma_precond_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(A && cfg, MA_ERR_INVALID, "A or cfg is NULL");
  MA_REQUIRE(cfg->coarsening >= 0 && cfg->coarsening <= 2 && cfg->interpolation >= 0 && cfg->interpolation <= 2 && cfg->smoother >= 0 && cfg->smoother <= 3 &&
             cfg->cycle >= 0 && cfg->cycle <= 2, MA_ERR_INVALID, "AmgConfig: coarsening %d, interpolation %d, smoother %d, cycle %d", (int)cfg->coarsening,
             (int)cfg->interpolation, (int)cfg->smoother, (int)cfg->cycle);
  MA_REQUIRE(cfg->max_levels >= 1 && cfg->max_levels <= 64 && cfg->coarse_size >= 0 && cfg->num_pre_smooth >= 0 && cfg->num_post_smooth >= 0 &&
             cfg->max_interp_elements >= 0, MA_ERR_INVALID, "AmgConfig: max_levels %d (1..64), coarse_size %d, sweeps %d / %d", (int)cfg->max_levels,
             (int)cfg->coarse_size, (int)cfg->num_pre_smooth, (int)cfg->num_post_smooth);
  const auto t0 = std::chrono::steady_clock::now();
  int64_t n = 0, nnz = 0, nc = 0; int dev = 0;
  int rc = ma_csr_num_rows(A, &n, &nnz); if (rc) return rc;
  rc = ma_csr_num_cols(A, &nc); if (rc) return rc;
  rc = ma_csr_device(A, &dev); if (rc) return rc;
  MA_REQUIRE(nc == n && n > 0, MA_ERR_INVALID, "AMG needs a square, non-empty operator");
  HostCsr H; H.nr = n; H.nc = n; H.ptr.resize((size_t)n + 1); H.col.resize((size_t)nnz); H.val.resize((size_t)nnz);
  H.col.resize((size_t)std::max<int64_t>(nnz, 1)); H.val.resize((size_t)std::max<int64_t>(nnz, 1));
  rc = ma_csr_get(A, H.ptr.data(), H.col.data(), reinterpret_cast<ma_c64*>(H.val.data())); if (rc) return rc;      // c64 and ma_c64 share their layout (ma_common.hpp)
  H.col.resize((size_t)nnz); H.val.resize((size_t)nnz);
  const auto t1 = std::chrono::steady_clock::now();
  std::vector<HostCsr> As, Ps, Rs; double gc = 1.0, oc = 1.0;
  rc = amg_setup_host(H, *cfg, As, Ps, Rs, &gc, &oc); if (rc) return rc;
  const auto t2 = std::chrono::steady_clock::now();
  const size_t L = As.size();
  std::vector<ma_csr*> hA(L, nullptr), hP(L, nullptr), hR(L, nullptr), owned;
  hA[0] = A;
  auto up = [&](const HostCsr& m, bool square, ma_csr** h) -> int {
    static const int64_t zero_col = 0; static const ma_c64 zero_val = {0.0, 0.0};
    const int64_t* cp = m.col.empty() ? &zero_col : m.col.data();
    const ma_c64* vp = m.val.empty() ? &zero_val : reinterpret_cast<const ma_c64*>(m.val.data());
    int r = square ? ma_csr_create(m.nr, m.ptr.data(), cp, vp, dev, h) : ma_csr_create_rect(m.nr, m.nc, m.ptr.data(), cp, vp, dev, h);
    if (!r) owned.push_back(*h);
    return r;
  };
  for (size_t l = 0; l < L && !rc; ++l) {
    if (l > 0) rc = up(As[l], true, &hA[l]);
    if (!rc && l + 1 < L) { rc = up(Ps[l], false, &hP[l]); if (!rc) rc = up(Rs[l], false, &hR[l]); }
  }
  ma_precond* M = nullptr;
  if (!rc) rc = ma_precond_create_amg((int32_t)L, hA.data(), hP.data(), hR.data(), cfg->smoother == 3 ? 0 : cfg->smoother, cfg->jacobi_weight, cfg->num_pre_smooth,
                                      cfg->num_post_smooth, cfg->cycle, &M);
  if (rc) { for (ma_csr* h : owned) (void)ma_csr_destroy(h); return rc; }
  M->amg_owned = owned; M->amg_gc = gc; M->amg_oc = oc;
  M->amg_setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
#ifdef MA_DIAGNOSTICS
  if (getenv("MA_AMG_TIMING"))                               // diagnostic build only
#else
  if (false)
#endif
    fprintf(stderr, "[amg setup] read back %.0f, hierarchy %.0f, uploads + level vectors %.0f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                                       std::chrono::duration<double, std::milli>(t2 - t1).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());
  *out = M; return MA_OK;
}
int ma_precond_amg_info(ma_precond_t* M, int32_t* num_levels, double* grid_complexity, double* operator_complexity, double* setup_time_ms) {
  MA_REQUIRE(M && M->kind == 5, MA_ERR_INVALID, "not an AMG preconditioner");
  if (num_levels) *num_levels = (int32_t)M->lv.size();
  if (grid_complexity) *grid_complexity = M->amg_gc;
  if (operator_complexity) *operator_complexity = M->amg_oc;
  if (setup_time_ms) *setup_time_ms = M->amg_setup_ms;
  return MA_OK;
}
int ma_precond_amg_level(ma_precond_t* M, int32_t level, ma_csr_t** A, ma_csr_t** P, ma_csr_t** R) {
  MA_REQUIRE(M && M->kind == 5, MA_ERR_INVALID, "not an AMG preconditioner");
  MA_REQUIRE(level >= 0 && (size_t)level < M->lv.size(), MA_ERR_INVALID, "level %d of %d", (int)level, (int)M->lv.size());
  const AmgLevelDev& L = M->lv[(size_t)level];
  if (A) *A = L.A;
  if (P) *P = L.P;
  if (R) *R = L.R;
  return MA_OK;
}
// IluPreconditioner::from_csr(matrix) (math-solvers/src/preconditioners/ilu.rs:36-140): ILU(0) on the matrix' own pattern, the
// factorisation on the HOST with the reference's loops (its row-k lookups included: first the entry right of the diagonal, then a
// scan of row k), L and U split into two device operators; apply = the two triangular solves, level-scheduled. Pivots below 1e-30
// are skipped by the factorisation as in the reference; the solves skip a row whose u_ii is below 1e-15 in modulus (the reference:
// 1e-30, and it still subtracts the row's sum there).
// the in-place ILU(0) factorisation of ilu.rs:55-98 on the host (ilu_parallel.rs:397-449 runs the same arithmetic with plain scans for
// its row-k lookups): rp / col / val come back holding pattern and factor values
static int ilu0_factor_host(ma_csr_t* csr, int64_t* n_out, std::vector<int64_t>& rp, std::vector<int64_t>& col, std::vector<cplx>& val) {
  int64_t n = 0, nnz = 0, nc = 0;
  int rc = ma_csr_num_rows(csr, &n, &nnz); if (rc) return rc;
  rc = ma_csr_num_cols(csr, &nc); if (rc) return rc;
  MA_REQUIRE(nc == n, MA_ERR_INVALID, "ILU(0) needs a square operator");
  rp.assign((size_t)n + 1, 0); col.assign((size_t)std::max<int64_t>(nnz, 1), 0);
  std::vector<ma_c64> v0((size_t)std::max<int64_t>(nnz, 1));
  rc = ma_csr_get(csr, rp.data(), col.data(), v0.data()); if (rc) return rc;
  val.assign((size_t)std::max<int64_t>(nnz, 1), cplx(0.0, 0.0));
  for (int64_t q = 0; q < nnz; ++q) val[(size_t)q] = cplx(v0[(size_t)q].re, v0[(size_t)q].im);
  const int64_t none = -1;
  std::vector<int64_t> diag((size_t)n, none);
  for (int64_t i = 0; i < n; ++i)
    for (int64_t idx = rp[(size_t)i]; idx < rp[(size_t)i + 1]; ++idx) if (col[(size_t)idx] == i) { diag[(size_t)i] = idx; break; }
  for (int64_t i = 0; i < n; ++i)                                              // :55-98
    for (int64_t idx = rp[(size_t)i]; idx < rp[(size_t)i + 1]; ++idx) {
      const int64_t k = col[(size_t)idx];
      if (k >= i) break;
      const int64_t ukk = diag[(size_t)k];
      if (ukk == none) continue;
      const cplx u_kk = val[(size_t)ukk];
      if (std::abs(u_kk) < 1e-30) continue;
      const cplx l_ik = val[(size_t)idx] * (std::conj(u_kk) / std::norm(u_kk));   // * u_kk.inv()
      val[(size_t)idx] = l_ik;
      for (int64_t jx = rp[(size_t)i]; jx < rp[(size_t)i + 1]; ++jx) {
        const int64_t j = col[(size_t)jx];
        if (j <= k) continue;
        const int64_t first = diag[(size_t)k] + 1;
        if (first < rp[(size_t)k + 1] && col[(size_t)first] == j) val[(size_t)jx] = val[(size_t)jx] - l_ik * val[(size_t)first];
        else
          for (int64_t sx = rp[(size_t)k] + 1; sx < rp[(size_t)k + 1]; ++sx)
            if (col[(size_t)sx] == j) { val[(size_t)jx] = val[(size_t)jx] - l_ik * val[(size_t)sx]; break; }
      }
    }
  *n_out = n;
  return MA_OK;
}
int ma_precond_create_ilu0(ma_csr_t* csr, ma_precond_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(csr, MA_ERR_INVALID, "csr is NULL");
  int64_t n = 0; std::vector<int64_t> rp, col; std::vector<cplx> val;
  int rc = ilu0_factor_host(csr, &n, rp, col, val); if (rc) return rc;
  std::vector<int64_t> lrp(1, 0), lci, urp(1, 0), uci; std::vector<ma_c64> lv, uv;   // :100-126
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t idx = rp[(size_t)i]; idx < rp[(size_t)i + 1]; ++idx) {
      const int64_t j = col[(size_t)idx]; const cplx z = val[(size_t)idx];
      if (j < i) { lci.push_back(j); lv.push_back(ma_c64{z.real(), z.imag()}); }
      else { uci.push_back(j); uv.push_back(ma_c64{z.real(), z.imag()}); }
    }
    lrp.push_back((int64_t)lci.size()); urp.push_back((int64_t)uci.size());
  }
  ma_precond* M = new (std::nothrow) ma_precond(); MA_REQUIRE(M, MA_ERR_NOMEM, "host allocation failed");
  M->kind = 6; M->n = n;
  int dev = 0;
  rc = ma_csr_device(csr, &dev);
  M->device = dev;
  static const int64_t zero_i = 0; static const ma_c64 zero_c = {0.0, 0.0};
  if (!rc) rc = ma_csr_create(n, lrp.data(), lci.empty() ? &zero_i : lci.data(), lv.empty() ? &zero_c : lv.data(), dev, &M->ilu_l);
  if (!rc) rc = ma_csr_create(n, urp.data(), uci.empty() ? &zero_i : uci.data(), uv.empty() ? &zero_c : uv.data(), dev, &M->ilu_u);
  if (!rc && hipMalloc(&M->d_tmp, sizeof(c64) * (size_t)std::max<int64_t>(n, 1)) != hipSuccess) { set_error("ILU workspace"); rc = MA_ERR_NOMEM; }
  if (rc) { ma_precond_destroy(M); return rc; }
  *out = M;
  return MA_OK;
}
// IluFixedPointPreconditioner::from_csr(matrix, iterations) (ilu_parallel.rs:397-495) and its apply (:510-590): the ILU(0) factors on
// the matrix' own pattern, then x = D^-1 r and `iterations` times x <- D^-1 (r - (L + U_off) x) with D = diag(U). That is the Jacobi
// sweep (omega = 1, from zero) of the matrix F that holds L, D and U_off in one pattern -- the factorisation's in-place result --
// run iterations + 1 times: F goes up as an operator of its own and the device Jacobi smoother does the rest.
// (The smoother takes 1 / u_ii = 1 where |u_ii| <= 1e-15; the reference: 1e-30.)
int ma_precond_create_ilu_fixed_point(ma_csr_t* csr, int32_t iterations, ma_precond_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(csr && iterations >= 0, MA_ERR_INVALID, "csr is NULL or iterations < 0");
  int64_t n = 0; std::vector<int64_t> rp, col; std::vector<cplx> val;
  int rc = ilu0_factor_host(csr, &n, rp, col, val); if (rc) return rc;
  std::vector<ma_c64> fv(val.size());
  for (size_t q = 0; q < val.size(); ++q) fv[q] = ma_c64{val[q].real(), val[q].imag()};
  int dev = 0; rc = ma_csr_device(csr, &dev); if (rc) return rc;
  ma_csr* F = nullptr;
  rc = ma_csr_create(n, rp.data(), col.data(), fv.data(), dev, &F); if (rc) return rc;
  rc = ma_precond_create_jacobi(F, 1.0, iterations + 1, out);
  if (rc) { (void)ma_csr_destroy(F); return rc; }
  (*out)->amg_owned.push_back(F);
  return MA_OK;
}
// AdditiveSchwarzPreconditioner::from_csr(matrix, num_subdomains, overlap) (math-solvers/src/preconditioners/schwarz.rs:84-145):
// contiguous index blocks (:90-100), each grown `overlap` times along the matrix graph (extend_partition, :196-229), weights 1 / (number of
// subdomains holding the row) (:112-128), per subdomain the local matrix and its ILU(0) (build_subdomain / ilu_factorize, :231-352).
// On the device the subdomains are stacked: E gathers r into the stacked vector, the block-diagonal matrix of the local matrices is
// one operator whose ILU(0) IS the subdomains' ILU(0)s (no entry couples two blocks) and whose level-scheduled solves run all
// subdomains at once, and the weighted transpose of E adds the local solutions back in subdomain order (apply, :394-408).
int ma_precond_create_schwarz(ma_csr_t* csr, int32_t num_subdomains, int32_t overlap, ma_precond_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL"); *out = nullptr;
  MA_REQUIRE(csr && overlap >= 0, MA_ERR_INVALID, "csr is NULL or overlap < 0");
  int64_t n = 0, nnz = 0, nc = 0; int dev = 0;
  int rc = ma_csr_num_rows(csr, &n, &nnz); if (rc) return rc;
  rc = ma_csr_num_cols(csr, &nc); if (rc) return rc;
  rc = ma_csr_device(csr, &dev); if (rc) return rc;
  MA_REQUIRE(nc == n && n > 0, MA_ERR_INVALID, "additive Schwarz needs a square, non-empty operator");
  std::vector<int64_t> rp((size_t)n + 1), col((size_t)std::max<int64_t>(nnz, 1)); std::vector<ma_c64> val((size_t)std::max<int64_t>(nnz, 1));
  rc = ma_csr_get(csr, rp.data(), col.data(), val.data()); if (rc) return rc;
  const int64_t ns = std::min<int64_t>(std::max<int64_t>(num_subdomains, 1), n);
  const int64_t base = n / ns, rem = n % ns;
  std::vector<int64_t> e_rp(1, 0), e_ci, b_rp(1, 0), b_ci; std::vector<ma_c64> e_v, b_v;
  std::vector<std::vector<int64_t>> cols_of((size_t)n);           // stacked rows that hold global row g, in subdomain order
  std::vector<char> in((size_t)n); std::vector<int64_t> frontier, next, g2l((size_t)n);
  int64_t start = 0, mn = n + 1, mx = 0;
  for (int64_t s_ = 0; s_ < ns; ++s_) {
    const int64_t size = base + (s_ < rem ? 1 : 0);
    std::fill(in.begin(), in.end(), 0);
    frontier.clear();
    for (int64_t i = start; i < start + size; ++i) { in[(size_t)i] = 1; frontier.push_back(i); }
    start += size;
    for (int o = 0; o < overlap; ++o) {                            // extend_partition
      next.clear();
      for (int64_t idx : frontier)
        for (int64_t q = rp[(size_t)idx]; q < rp[(size_t)idx + 1]; ++q) { const int64_t nb = col[(size_t)q]; if (nb != idx && !in[(size_t)nb]) { in[(size_t)nb] = 1; next.push_back(nb); } }
      frontier.swap(next);
    }
    const int64_t off = (int64_t)e_ci.size();
    int64_t ln = 0;
    for (int64_t g = 0; g < n; ++g) if (in[(size_t)g]) { g2l[(size_t)g] = off + ln; ++ln; e_ci.push_back(g); e_v.push_back(ma_c64{1.0, 0.0}); e_rp.push_back((int64_t)e_ci.size()); cols_of[(size_t)g].push_back(off + ln - 1); }
    for (int64_t g = 0; g < n; ++g) {                              // build_subdomain: the rows and columns of the subdomain, in global order
      if (!in[(size_t)g]) continue;
      for (int64_t q = rp[(size_t)g]; q < rp[(size_t)g + 1]; ++q) { const int64_t c = col[(size_t)q]; if (in[(size_t)c]) { b_ci.push_back(g2l[(size_t)c]); b_v.push_back(val[(size_t)q]); } }
      b_rp.push_back((int64_t)b_ci.size());
    }
    mn = std::min(mn, ln); mx = std::max(mx, ln);
  }
  const int64_t N = (int64_t)e_ci.size();
  std::vector<int64_t> t_rp(1, 0), t_ci; std::vector<ma_c64> t_v;
  for (int64_t g = 0; g < n; ++g) {
    const size_t c = cols_of[(size_t)g].size();
    const double w = c > 0 ? 1.0 / (double)c : 1.0;
    for (int64_t e : cols_of[(size_t)g]) { t_ci.push_back(e); t_v.push_back(ma_c64{w, 0.0}); }
    t_rp.push_back((int64_t)t_ci.size());
  }
  ma_precond* M = new (std::nothrow) ma_precond(); MA_REQUIRE(M, MA_ERR_NOMEM, "host allocation failed");
  M->kind = 7; M->n = n; M->device = dev;
  M->sch_stats[0] = ns; M->sch_stats[1] = mn; M->sch_stats[2] = mx; M->sch_stats[3] = N;
  static const int64_t zero_i = 0; static const ma_c64 zero_c = {0.0, 0.0};
  ma_csr* B = nullptr;
  rc = ma_csr_create_rect(N, n, e_rp.data(), e_ci.data(), e_v.data(), dev, &M->sch_E);
  if (!rc) rc = ma_csr_create_rect(n, N, t_rp.data(), t_ci.data(), t_v.data(), dev, &M->sch_Et);
  if (!rc) rc = ma_csr_create(N, b_rp.data(), b_ci.empty() ? &zero_i : b_ci.data(), b_v.empty() ? &zero_c : b_v.data(), dev, &B);
  if (!rc) rc = ma_precond_create_ilu0(B, &M->sch_inner);
  if (B) (void)ma_csr_destroy(B);                                   // the factors are operators of their own
  if (!rc && hipMalloc(&M->sch_a, sizeof(c64) * (size_t)N) != hipSuccess) { set_error("Schwarz workspace"); rc = MA_ERR_NOMEM; }
  if (!rc && hipMalloc(&M->sch_b, sizeof(c64) * (size_t)N) != hipSuccess) { set_error("Schwarz workspace"); rc = MA_ERR_NOMEM; }
  if (rc) { ma_precond_destroy(M); return rc; }
  *out = M; return MA_OK;
}
// AdditiveSchwarzPreconditioner::stats (schwarz.rs:148-170): subdomains, smallest, largest, mean size
int ma_precond_schwarz_stats(ma_precond_t* M, int64_t* num_subdomains, int64_t* min_size, int64_t* max_size, double* avg_size) {
  MA_REQUIRE(M && M->kind == 7, MA_ERR_INVALID, "not an additive Schwarz preconditioner");
  if (num_subdomains) *num_subdomains = M->sch_stats[0];
  if (min_size) *min_size = M->sch_stats[1];
  if (max_size) *max_size = M->sch_stats[2];
  if (avg_size) *avg_size = (double)M->sch_stats[3] / (double)M->sch_stats[0];
  return MA_OK;
}
// z = M^-1 r on device vectors (z and r distinct)
// AmgPreconditioner::apply (amg.rs:1068-1103) after z = 0: the cycle (V; W = the V-cycle twice; F = a second V-cycle on the residual)
static int amg_apply_body(ma_precond* M, c64* z, const c64* r, hipStream_t st, bool amg_lazy) {
  int rc = amg_v_cycle(M, 0, z, r, st, amg_lazy);
  if (!rc && M->amg_cycle == 1) rc = amg_v_cycle(M, 0, z, r, st);             // W: the V-cycle twice (:1084-1087)
  if (!rc && M->amg_cycle == 2) {                                              // F: a second V-cycle on the residual (:1088-1094)
    c64* res = M->d_tmp; c64* corr = M->d_tmp + M->n;
    rc = ma_csr_residual_dev(M->lv[0].A, z, r, res, st);
    if (!rc && !amg_lazy && hipMemsetAsync(corr, 0, sizeof(c64) * (size_t)M->n, st) != hipSuccess) { set_error("AMG: clearing the correction failed"); rc = MA_ERR_HIP; }
    if (!rc) rc = amg_v_cycle(M, 0, corr, res, st, amg_lazy);
    if (!rc) rc = op_launch_axpby(M->n, 1.0, 0.0, z, 1.0, 0.0, corr, z, st);
  }
  return rc;
}
int ma_precond_apply_dev(ma_precond_t* M, const void* d_r, void* d_z, void* stream) {
  MA_REQUIRE(M && d_r && d_z, MA_ERR_INVALID, "NULL argument");
  if (M->kind == 0) { MA_HIP(hipMemcpyAsync(d_z, d_r, sizeof(c64) * (size_t)M->n, hipMemcpyDeviceToDevice, (hipStream_t)stream)); return MA_OK; }
  if (M->kind == 4) return op_launch_cmul(M->n, M->d_invdiag, (const c64*)d_r, (c64*)d_z, (hipStream_t)stream);
  if (M->kind == 7) {                                    // AdditiveSchwarzPreconditioner::apply (schwarz.rs:394-408): gather, local solves, weighted scatter-add
    int rc = ma_csr_spmv_dev(M->sch_E, d_r, M->sch_a, stream);
    if (!rc) rc = ma_precond_apply_dev(M->sch_inner, M->sch_a, M->sch_b, stream);
    if (!rc) rc = ma_csr_spmv_dev(M->sch_Et, M->sch_b, d_z, stream);
    return rc;
  }
  if (M->kind == 6) {                                    // IluPreconditioner::apply (ilu.rs:143-175): L y = r forward, U z = y backward
    // a forward sweep over a matrix without upper entries IS the forward substitution (and a backward sweep over one without lower
    // entries the backward substitution): the level-scheduled Gauss-Seidel sweeps of the two factors, amg.rs' form sum * diag.inv()
    // U: mode 2 -- x_i = y_i - sum always, times u_ii^-1 only where |u_ii| > 1e-30 (ilu.rs:154-170): nothing of what the caller's
    // d_z held before survives into the result
    int rc = ma_csr_gauss_seidel_sweep_dev(M->ilu_l, M->d_tmp, d_r, 1, 0, stream);
    if (!rc) rc = ma_csr_gauss_seidel_sweep_dev(M->ilu_u, d_z, M->d_tmp, 2, 1, stream);
    return rc;
  }
  const bool amg_lazy = M->kind == 5 && amg_first_sweep_writes(M, 0);          // z = 0 is then written by the cycle's first sweep
  // (round 3 measured the cycle as ONE graph replay, 1.39 ms against 1.06 ms eager, and the coarse levels in one workgroup, 1.08-1.40 ms:
  // profiles/r03_amg_cycle_variants.json; both are gone)
  if (!amg_lazy) MA_HIP(hipMemsetAsync(d_z, 0, sizeof(c64) * (size_t)M->n, (hipStream_t)stream));
  if (M->kind == 5) return amg_apply_body(M, (c64*)d_z, (const c64*)d_r, (hipStream_t)stream, amg_lazy);
  if (M->kind == 1) return ma_csr_jacobi_dev(M->csr, d_z, d_r, M->omega, M->sweeps, M->d_tmp, stream);
  if (M->kind == 3) return ma_csr_sym_gauss_seidel_dev(M->csr, d_z, d_r, M->sweeps, stream);
  return ma_csr_l1jacobi_dev(M->csr, d_z, d_r, M->sweeps, M->d_tmp, stream);
}

// host-buffer convenience form of apply (tests, small systems)
int ma_precond_apply(ma_precond_t* M, const ma_c64* r_host, ma_c64* z_host) {
  MA_REQUIRE(M && r_host && z_host, MA_ERR_INVALID, "NULL argument");
  int rc = use_device(M->device); if (rc) return rc;
  c64 *d_r = nullptr, *d_z = nullptr; size_t bytes = sizeof(c64) * (size_t)M->n;
  if (hipMalloc(&d_r, bytes) != hipSuccess || hipMalloc(&d_z, bytes) != hipSuccess) { if (d_r) (void)hipFree(d_r); set_error("preconditioner vectors: out of device memory"); return MA_ERR_NOMEM; }
  rc = MA_OK;
  if (hipMemcpy(d_r, r_host, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = MA_ERR_HIP;
  if (!rc) rc = ma_precond_apply_dev(M, d_r, d_z, nullptr);
  if (!rc && hipMemcpy(z_host, d_z, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = MA_ERR_HIP;
  (void)hipFree(d_r); (void)hipFree(d_z);
  if (!rc) rc = spin_error_check("ma_precond_apply");
  return rc;
}

// ------------------------------------------------------------------ GMRES(m), gmres.rs:105-277 (and :282-585 with a preconditioner)
// b, x0 (may be NULL), x_out: host vectors of n entries. info = {iterations, restarts, converged, residual}.
// Non-convergence is not an error (gmres.rs:270-276): the flag is returned with MA_OK.
static int gmres_impl(ma_op_t* o, ma_precond_t* Mp, const ma_c64* b_host, const ma_c64* x0_host, int32_t restart, int32_t max_iterations, double tol,
                      ma_c64* x_out, ma_gmres_info_t* info) {
  MA_REQUIRE(o && b_host && x_out && info, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(!Mp || Mp->n == o->n, MA_ERR_DIM, "preconditioner and operator sizes differ");
  MA_REQUIRE(restart >= 1 && max_iterations >= 0, MA_ERR_INVALID, "restart must be >= 1");
  MA_HIP(hipSetDevice(o->device));
  const long long n = o->n; const int m = restart;
  hipStream_t st = nullptr;
  c64 *V = nullptr, *w = nullptr, *x = nullptr, *b = nullptr, *scal = nullptr, *partial = nullptr, *t = nullptr; void* mgs_slots = nullptr; unsigned* mgs_err = nullptr;
  auto cleanup = [&]() { void* p[] = {V, w, x, b, scal, partial, t, mgs_slots, mgs_err}; for (void* q : p) if (q) (void)hipFree(q); };
#define GM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return MA_ERR_HIP; } } while (0)
#define GM_RC(call) do { int rc_ = (call); if (rc_) { cleanup(); return rc_; } } while (0)
  GM_HIP(hipMalloc(&V, sizeof(c64) * (size_t)n * (size_t)(m + 1)));
  GM_HIP(hipMalloc(&w, sizeof(c64) * (size_t)n));
  GM_HIP(hipMalloc(&x, sizeof(c64) * (size_t)n));
  GM_HIP(hipMalloc(&b, sizeof(c64) * (size_t)n));
  GM_HIP(hipMalloc(&scal, sizeof(c64) * (size_t)(m + 5)));
  GM_HIP(hipMalloc(&partial, sizeof(c64) * 256));
  GM_HIP(hipMalloc(&t, sizeof(c64) * (size_t)n));
  // the Gram-Schmidt step as one launch (op_launch_gmres_mgs); MA_GMRES_FUSED_MGS=0: an inner product and an update kernel per basis vector
  const char* efuse = getenv("MA_GMRES_FUSED_MGS");
  bool fused_mgs = !(efuse && atoi(efuse) == 0);
  if (fused_mgs) {
    GM_HIP(hipMalloc(&mgs_slots, (size_t)op_mgs_slot_bytes(m)));
    GM_HIP(hipMalloc(&mgs_err, sizeof(unsigned)));
    GM_HIP(hipMemset(mgs_err, 0, sizeof(unsigned)));
  }
  GM_HIP(hipMemcpy(b, b_host, sizeof(c64) * (size_t)n, hipMemcpyHostToDevice));
  if (x0_host) GM_HIP(hipMemcpy(x, x0_host, sizeof(c64) * (size_t)n, hipMemcpyHostToDevice));
  else GM_HIP(hipMemset(x, 0, sizeof(c64) * (size_t)n));
  auto norm_of = [&](const c64* v, double* out) -> int {
    int rc = op_launch_dot(n, v, nullptr, 1, partial, scal, st); if (rc) return rc;
    c64 h; MA_HIP(hipMemcpy(&h, scal, sizeof(c64), hipMemcpyDeviceToHost)); *out = h.re; return MA_OK;
  };
  // left preconditioning measures everything in the M^-1 norm: b_norm = ||M^-1 b|| (gmres.rs:299-300)
  double b_norm = 0.0;
  if (Mp) { GM_RC(ma_precond_apply_dev(Mp, b, w, st)); GM_RC(norm_of(w, &b_norm)); }
  else GM_RC(norm_of(b, &b_norm));
  info->iterations = 0; info->restarts = 0; info->converged = 1; info->residual = 0.0;
  if (b_norm < 1e-15) { GM_HIP(hipMemcpy(x_out, x, sizeof(c64) * (size_t)n, hipMemcpyDeviceToHost)); cleanup(); return MA_OK; }

  std::vector<cplx> H((size_t)(m + 1) * m), cs(m), sn(m), g(m + 1), y(m), hcol(m + 3);
  auto Hh = [&](int i, int j) -> cplx& { return H[(size_t)i * m + j]; };
  auto givens = [](cplx a, cplx bb, cplx* c, cplx* s) {
    if (std::abs(bb) < 1e-30) { *c = 1.0; *s = 0.0; return; }
    if (std::abs(a) < 1e-30) { *c = 0.0; *s = 1.0; return; }
    const double r = std::sqrt(std::norm(a) + std::norm(bb));
    *c = a * (1.0 / r); *s = bb * (1.0 / r);
  };
  auto solve_upper = [&](int k) {
    for (int i = k - 1; i >= 0; --i) {
      cplx sum = g[i];
      for (int q = i + 1; q < k; ++q) sum -= Hh(i, q) * y[q];
      y[i] = std::abs(Hh(i, i)) > 1e-30 ? sum * (1.0 / Hh(i, i)) : cplx(0.0, 0.0);
    }
  };
  auto update_x = [&](int k) -> int {                        // x += sum_i y_i v_i in one launch: the coefficients go up negated (multi_axpy subtracts)
    std::vector<c64> neg((size_t)std::max(k, 1));
    for (int i = 0; i < k; ++i) neg[(size_t)i] = c64{-y[i].real(), -y[i].imag()};
    MA_HIP(hipMemcpy(scal + 1, neg.data(), sizeof(c64) * (size_t)k, hipMemcpyHostToDevice));
    return op_launch_multi_axpy(n, k, scal + 1, V, x, nullptr, nullptr, st);
  };
  int total = 0, restarts = 0; bool done = false;
  for (int outer = 0; outer < max_iterations && !done; ++outer) {
    GM_RC(ma_op_apply_dev(o, x, w, st));
    if (Mp) { GM_RC(op_launch_axpby(n, 1.0, 0.0, b, -1.0, 0.0, w, t, st)); GM_RC(ma_precond_apply_dev(Mp, t, V, st)); }   // r = M^-1 (b - A x)
    else GM_RC(op_launch_axpby(n, 1.0, 0.0, b, -1.0, 0.0, w, V, st));     // r = b - A x into v[0]
    double beta = 0.0; GM_RC(norm_of(V, &beta));
    double rel = beta / b_norm;
    if (rel < tol) { info->iterations = total; info->restarts = restarts; info->residual = rel; info->converged = 1; done = true; break; }
    GM_RC(op_launch_axpby(n, 1.0 / beta, 0.0, V, 0.0, 0.0, nullptr, V, st));
    std::fill(H.begin(), H.end(), cplx(0.0, 0.0)); std::fill(g.begin(), g.end(), cplx(0.0, 0.0));
    g[0] = beta;
    bool inner_conv = false, finished = false;
    for (int j = 0; j < m; ++j) {
      total += 1;
      if (Mp) { GM_RC(ma_op_apply_dev(o, V + (size_t)j * n, t, st)); GM_RC(ma_precond_apply_dev(Mp, t, w, st)); }   // w = M^-1 A v_j
      else GM_RC(ma_op_apply_dev(o, V + (size_t)j * n, w, st));
      bool stepped = false;
      if (fused_mgs) {                                                      // modified Gram-Schmidt: h_0..h_j, w, |w| in one launch
        const int frc = op_launch_gmres_mgs(n, V, j, w, mgs_slots, scal + 1, mgs_err, st);
        if (frc == MA_OK) stepped = true;
        else if (frc == MA_ERR_UNSUPPORTED) fused_mgs = false;              // vector too long for the register-resident form
        else { cleanup(); return frc; }
      }
      if (!stepped) {
        for (int i = 0; i <= j; ++i) {                                      // the same step as separate kernels, scalars stay on the device
          GM_RC(op_launch_dot(n, V + (size_t)i * n, w, 0, partial, scal + 1 + i, st));
          GM_RC(op_launch_axpy_dev(n, scal + 1 + i, -1.0, V + (size_t)i * n, w, st));
        }
        GM_RC(op_launch_dot(n, w, nullptr, 1, partial, scal + 2 + j, st));
      }
      GM_HIP(hipMemcpy(hcol.data(), scal + 1, sizeof(c64) * (size_t)(j + 2 + (stepped ? 1 : 0)), hipMemcpyDeviceToHost));   // one sync per inner step
      if (stepped && hcol[j + 2].real() != 0.0) {            // the one-launch step gave up a wait (its flag rides behind |w|): stop at once
        set_error("a Gram-Schmidt step was abandoned at its exchange (iteration %d)", total);
        (void)spin_error_check("ma_gmres");                   // clears the device-wide word; the text above is what the caller reads
        set_error("a Gram-Schmidt step was abandoned at its exchange (iteration %d)", total);
        cleanup(); return MA_ERR_HIP;
      }
      for (int i = 0; i <= j; ++i) Hh(i, j) = hcol[i];
      const double wn = hcol[j + 1].real();
      Hh(j + 1, j) = wn;
      if (wn < 1e-14) inner_conv = true;
      else GM_RC(op_launch_axpby(n, Mp ? 1.0 / wn : 1.0 + (1.0 / wn - 1.0), 0.0, w, 0.0, 0.0, nullptr, V + (size_t)(j + 1) * n, st));   // w/|w| (:366) or w + (1/|w| - 1) w (:198-201)
      for (int i = 0; i < j; ++i) {
        const cplx t = std::conj(cs[i]) * Hh(i, j) + std::conj(sn[i]) * Hh(i + 1, j);
        Hh(i + 1, j) = cplx(0.0, 0.0) - sn[i] * Hh(i, j) + cs[i] * Hh(i + 1, j);
        Hh(i, j) = t;
      }
      cplx c, s; givens(Hh(j, j), Hh(j + 1, j), &c, &s);
      cs[j] = c; sn[j] = s;
      Hh(j, j) = std::conj(c) * Hh(j, j) + std::conj(s) * Hh(j + 1, j);
      Hh(j + 1, j) = 0.0;
      const cplx t = std::conj(c) * g[j] + std::conj(s) * g[j + 1];
      g[j + 1] = cplx(0.0, 0.0) - s * g[j] + c * g[j + 1];
      g[j] = t;
      rel = std::abs(g[j + 1]) / b_norm;
      if (rel < tol || inner_conv) {
        solve_upper(j + 1);
        GM_RC(update_x(j + 1));
        info->iterations = total; info->restarts = restarts; info->residual = rel; info->converged = 1;
        finished = true; done = true; break;
      }
    }
    if (finished) break;
    solve_upper(m);
    GM_RC(update_x(m));
    restarts += 1;
  }
  if (!done) {
    GM_RC(ma_op_apply_dev(o, x, w, st));
    GM_RC(op_launch_axpby(n, 1.0, 0.0, b, -1.0, 0.0, w, w, st));
    if (Mp) { GM_RC(ma_precond_apply_dev(Mp, w, t, st)); GM_HIP(hipMemcpyAsync(w, t, sizeof(c64) * (size_t)n, hipMemcpyDeviceToDevice, st)); }
    double rn = 0.0; GM_RC(norm_of(w, &rn));
    info->iterations = total; info->restarts = restarts; info->residual = rn / b_norm; info->converged = 0;
  }
  if (mgs_err) { unsigned ew = 0; GM_HIP(hipMemcpy(&ew, mgs_err, sizeof(unsigned), hipMemcpyDeviceToHost)); if (ew) { set_error("a Gram-Schmidt step was abandoned at its exchange"); cleanup(); return MA_ERR_HIP; } }
  // a flag-driven sweep inside the preconditioner (symmetric Gauss-Seidel, ILU, Schwarz, AMG smoothers) that abandoned a wait left
  // sentinel NaNs in its result: never return that with MA_OK
  GM_RC(spin_error_check("ma_gmres"));
  GM_HIP(hipMemcpy(x_out, x, sizeof(c64) * (size_t)n, hipMemcpyDeviceToHost));
  cleanup();
#undef GM_HIP
#undef GM_RC
  return MA_OK;
}

// ------------------------------------------------------------------ gmres_pipelined (iterative/gmres_pipelined.rs:18-250)
// p-GMRES: with the auxiliary basis Z = M^-1 A V the inner products of step j (z_j against v_0..v_j, classical Gram-Schmidt) do
// not depend on q = M^-1 A z_j, so the reference runs the two concurrently (rayon::join, :106-119). Here: the operator (and
// preconditioner) on the caller's stream, the j + 1 inner products on a second stream, joined by events; the scalars stay on the
// device, one host synchronisation per inner step (the column of H for the Givens update), as in ma_gmres.
static int gmres_pipelined_impl(ma_op_t* o, ma_precond_t* Mp, const ma_c64* b_host, const ma_c64* x0_host, int32_t restart, int32_t max_iterations, double tol,
                                ma_c64* x_out, ma_gmres_info_t* info) {
  MA_REQUIRE(o && b_host && x_out && info, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(!Mp || Mp->n == o->n, MA_ERR_DIM, "preconditioner and operator sizes differ");
  MA_REQUIRE(restart >= 1 && max_iterations >= 0, MA_ERR_INVALID, "restart must be >= 1");
  MA_HIP(hipSetDevice(o->device));
  const long long n = o->n; const int m = restart;
  hipStream_t st = nullptr, s2 = nullptr;
  hipEvent_t ev_z = nullptr, ev_h = nullptr;
  c64 *V = nullptr, *Z = nullptr, *x = nullptr, *b = nullptr, *scal = nullptr, *partial = nullptr, *partial2 = nullptr, *t = nullptr, *q = nullptr, *vn = nullptr;
  auto cleanup = [&]() {
    void* p[] = {V, Z, x, b, scal, partial, partial2, t, q, vn}; for (void* r : p) if (r) (void)hipFree(r);
    if (ev_z) (void)hipEventDestroy(ev_z); if (ev_h) (void)hipEventDestroy(ev_h); if (s2) (void)hipStreamDestroy(s2);
  };
#define GM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return MA_ERR_HIP; } } while (0)
#define GM_RC(call) do { int rc_ = (call); if (rc_) { cleanup(); return rc_; } } while (0)
  GM_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  GM_HIP(hipEventCreateWithFlags(&ev_z, hipEventDisableTiming)); GM_HIP(hipEventCreateWithFlags(&ev_h, hipEventDisableTiming));
  GM_HIP(hipMalloc(&V, sizeof(c64) * (size_t)n * (size_t)(m + 1)));
  GM_HIP(hipMalloc(&Z, sizeof(c64) * (size_t)n * (size_t)(m + 1)));
  for (c64** p : {&x, &b, &t, &q, &vn}) GM_HIP(hipMalloc(p, sizeof(c64) * (size_t)n));
  GM_HIP(hipMalloc(&scal, sizeof(c64) * (size_t)(m + 4)));
  GM_HIP(hipMalloc(&partial, sizeof(c64) * 256)); GM_HIP(hipMalloc(&partial2, sizeof(c64) * 256 * (size_t)(m + 1)));   // partial2: one row of 256 per basis vector (op_launch_multi_dot)
  GM_HIP(hipMemcpy(b, b_host, sizeof(c64) * (size_t)n, hipMemcpyHostToDevice));
  if (x0_host) GM_HIP(hipMemcpy(x, x0_host, sizeof(c64) * (size_t)n, hipMemcpyHostToDevice));
  else GM_HIP(hipMemset(x, 0, sizeof(c64) * (size_t)n));
  auto norm_of = [&](const c64* v, double* out) -> int {
    int rc = op_launch_dot(n, v, nullptr, 1, partial, scal, st); if (rc) return rc;
    c64 h; MA_HIP(hipMemcpy(&h, scal, sizeof(c64), hipMemcpyDeviceToHost)); *out = h.re; return MA_OK;
  };
  auto precond = [&](const c64* in, c64* out) -> int {        // M^-1 in -> out (identity: a copy)
    if (Mp) return ma_precond_apply_dev(Mp, in, out, st);
    MA_HIP(hipMemcpyAsync(out, in, sizeof(c64) * (size_t)n, hipMemcpyDeviceToDevice, st)); return MA_OK;
  };
  auto residual_into = [&](c64* out) -> int {                // out = M^-1 (b - A x)   (:40-48, :72-76)
    int rc = ma_op_apply_dev(o, x, t, st); if (rc) return rc;
    rc = op_launch_axpby(n, 1.0, 0.0, b, -1.0, 0.0, t, t, st); if (rc) return rc;
    return precond(t, out);
  };
  double b_norm = 0.0;
  GM_RC(residual_into(q)); GM_RC(norm_of(q, &b_norm));
  info->iterations = 0; info->restarts = 0; info->converged = 1; info->residual = 0.0;
  if (b_norm < 1e-15) { GM_HIP(hipMemcpy(x_out, x, sizeof(c64) * (size_t)n, hipMemcpyDeviceToHost)); cleanup(); return MA_OK; }

  std::vector<cplx> H((size_t)(m + 1) * m), cs(m), sn(m), g(m + 1), y(m), hcol(m + 3);
  auto Hh = [&](int i, int j) -> cplx& { return H[(size_t)i * m + j]; };
  auto givens = [](cplx a, cplx bb, cplx* c, cplx* s) {
    if (std::abs(bb) < 1e-30) { *c = 1.0; *s = 0.0; return; }
    if (std::abs(a) < 1e-30) { *c = 0.0; *s = 1.0; return; }
    const double r = std::sqrt(std::norm(a) + std::norm(bb));
    *c = a * (1.0 / r); *s = bb * (1.0 / r);
  };
  auto solve_upper = [&](int k) {
    for (int i = k - 1; i >= 0; --i) {
      cplx sum = g[i];
      for (int p = i + 1; p < k; ++p) sum -= Hh(i, p) * y[p];
      y[i] = std::abs(Hh(i, i)) > 1e-30 ? sum * (1.0 / Hh(i, i)) : cplx(0.0, 0.0);
    }
  };
  auto update_x = [&](int k) -> int {                        // x += sum_i y_i v_i in one launch: the coefficients go up negated (multi_axpy subtracts)
    std::vector<c64> neg((size_t)std::max(k, 1));
    for (int i = 0; i < k; ++i) neg[(size_t)i] = c64{-y[i].real(), -y[i].imag()};
    MA_HIP(hipMemcpy(scal + 1, neg.data(), sizeof(c64) * (size_t)k, hipMemcpyHostToDevice));
    return op_launch_multi_axpy(n, k, scal + 1, V, x, nullptr, nullptr, st);
  };
  int total = 0, restarts = 0; bool done = false;
  for (int outer = 0; outer < max_iterations && !done; ++outer) {
    GM_RC(residual_into(V));
    double beta = 0.0; GM_RC(norm_of(V, &beta));
    double rel = beta / b_norm;
    if (rel < tol) { info->iterations = total; info->restarts = restarts; info->residual = rel; info->converged = 1; done = true; break; }
    GM_RC(op_launch_axpby(n, 1.0 / beta, 0.0, V, 0.0, 0.0, nullptr, V, st));                       // v0 = r / beta (:93)
    GM_RC(ma_op_apply_dev(o, V, t, st)); GM_RC(precond(t, Z));                                    // z0 = M^-1 A v0 (:96-97)
    std::fill(H.begin(), H.end(), cplx(0.0, 0.0)); std::fill(g.begin(), g.end(), cplx(0.0, 0.0));
    g[0] = beta;
    int nv = 1; bool inner_conv = false, finished = false;
    for (int j = 0; j < m; ++j) {
      total += 1;
      const c64* zj = Z + (size_t)j * n;
      GM_HIP(hipEventRecord(ev_z, st)); GM_HIP(hipStreamWaitEvent(s2, ev_z, 0));
      GM_RC(ma_op_apply_dev(o, zj, t, st)); GM_RC(precond(t, q));                                  // q = M^-1 A z_j        | concurrently
      GM_RC(op_launch_multi_dot(n, V, j + 1, zj, partial2, scal + 1, s2));                        // h_ij = <v_i, z_j>, i <= j | (two launches)
      GM_HIP(hipEventRecord(ev_h, s2)); GM_HIP(hipStreamWaitEvent(st, ev_h, 0));
      GM_HIP(hipMemcpyAsync(vn, zj, sizeof(c64) * (size_t)n, hipMemcpyDeviceToDevice, st));       // :129-136
      GM_RC(op_launch_multi_axpy(n, j + 1, scal + 1, V, vn, Z, q, st));                           // vn -= h_ij v_i, q -= h_ij z_i (one launch)
      GM_RC(op_launch_dot(n, vn, nullptr, 1, partial, scal + 2 + j, st));                          // :139
      GM_HIP(hipMemcpy(hcol.data(), scal + 1, sizeof(c64) * (size_t)(j + 2), hipMemcpyDeviceToHost));
      for (int i = 0; i <= j; ++i) Hh(i, j) = hcol[i];
      const double wn = hcol[j + 1].real();
      Hh(j + 1, j) = wn;
      if (wn < 1e-14) inner_conv = true;
      else {
        GM_RC(op_launch_axpby(n, 1.0 / wn, 0.0, vn, 0.0, 0.0, nullptr, V + (size_t)nv * n, st));
        GM_RC(op_launch_axpby(n, 1.0 / wn, 0.0, q, 0.0, 0.0, nullptr, Z + (size_t)nv * n, st));
        nv += 1;
      }
      for (int i = 0; i < j; ++i) {
        const cplx tt = std::conj(cs[i]) * Hh(i, j) + std::conj(sn[i]) * Hh(i + 1, j);
        Hh(i + 1, j) = cplx(0.0, 0.0) - sn[i] * Hh(i, j) + cs[i] * Hh(i + 1, j);
        Hh(i, j) = tt;
      }
      cplx c, s_; givens(Hh(j, j), Hh(j + 1, j), &c, &s_);
      cs[j] = c; sn[j] = s_;
      Hh(j, j) = std::conj(c) * Hh(j, j) + std::conj(s_) * Hh(j + 1, j);
      Hh(j + 1, j) = 0.0;
      const cplx tt = std::conj(c) * g[j] + std::conj(s_) * g[j + 1];
      g[j + 1] = cplx(0.0, 0.0) - s_ * g[j] + c * g[j + 1];
      g[j] = tt;
      rel = std::abs(g[j + 1]) / b_norm;
      if (rel < tol || inner_conv) {
        solve_upper(j + 1);
        GM_RC(update_x(j + 1));
        info->iterations = total; info->restarts = restarts; info->residual = rel; info->converged = 1;
        finished = true; done = true; break;
      }
    }
    if (finished) break;
    solve_upper(m);
    GM_RC(update_x(m));
    restarts += 1;
  }
  if (!done) {
    GM_RC(residual_into(q));
    double rn = 0.0; GM_RC(norm_of(q, &rn));
    info->iterations = total; info->restarts = restarts; info->residual = rn / b_norm; info->converged = 0;
  }
  GM_RC(spin_error_check("ma_gmres_pipelined"));            // a sweep of the preconditioner that abandoned a wait: never MA_OK
  GM_HIP(hipMemcpy(x_out, x, sizeof(c64) * (size_t)n, hipMemcpyDeviceToHost));
  cleanup();
#undef GM_HIP
#undef GM_RC
  return MA_OK;
}
// gmres_pipelined(operator, precond, b, x0, config): M may be NULL (IdentityPreconditioner)
int ma_gmres_pipelined(ma_op_t* o, ma_precond_t* M, const ma_c64* b, const ma_c64* x0, int32_t restart, int32_t max_iterations, double tol,
                       ma_c64* x_out, ma_gmres_info_t* info) {
  return gmres_pipelined_impl(o, M, b, x0, restart, max_iterations, tol, x_out, info);
}

int ma_gmres(ma_op_t* o, const ma_c64* b, const ma_c64* x0, int32_t restart, int32_t max_iterations, double tol, ma_c64* x_out, ma_gmres_info_t* info) {
  return gmres_impl(o, nullptr, b, x0, restart, max_iterations, tol, x_out, info);
}
// gmres_preconditioned / gmres_preconditioned_with_guess (gmres.rs:282-585)
int ma_gmres_preconditioned(ma_op_t* o, ma_precond_t* M, const ma_c64* b, const ma_c64* x0, int32_t restart, int32_t max_iterations, double tol,
                            ma_c64* x_out, ma_gmres_info_t* info) {
  MA_REQUIRE(M, MA_ERR_INVALID, "preconditioner is NULL");
  return gmres_impl(o, M, b, x0, restart, max_iterations, tol, x_out, info);
}

}  // extern "C"

// ---------------------------------------------------------------- the other Krylov solvers of math-solvers/src/iterative: bicgstab.rs:46-182, cgs.rs:46-139,
// cg.rs:49-138. Same operator boundary as GMRES (ma_op_apply_dev), vectors resident, the scalars of a step (inner products <x, y> = sum conj(x_i) y_i and
// norms, blas_helpers.rs:21-50) come back to the host, where the reference takes its branches: breakdown below 1e-30, b_norm below 1e-15 -> x = 0 converged,
// relative residual against ||b||. kind 0 BiCGSTAB, 1 CGS, 2 CG. info->restarts is unused (0).
static int krylov_impl(int kind, ma_op_t* o, const ma_c64* b_host, int32_t max_iterations, double tol, ma_c64* x_out, ma_gmres_info_t* info) {
  MA_REQUIRE(o && b_host && x_out && info, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(max_iterations >= 0, MA_ERR_INVALID, "max_iterations must be >= 0");
  MA_HIP(hipSetDevice(o->device));
  const long long n = o->n;
  hipStream_t st = nullptr;
  c64* buf = nullptr; c64* scal = nullptr; c64* partial = nullptr;
  auto cleanup = [&]() { if (buf) (void)hipFree(buf); if (scal) (void)hipFree(scal); if (partial) (void)hipFree(partial); };
#define KR_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s failed: %s", #call, hipGetErrorString(e_)); cleanup(); return MA_ERR_HIP; } } while (0)
#define KR_RC(call) do { int rc_ = (call); if (rc_) { cleanup(); return rc_; } } while (0)
  KR_HIP(hipMalloc(&buf, sizeof(c64) * (size_t)n * 9));
  KR_HIP(hipMalloc(&scal, sizeof(c64) * 4));
  KR_HIP(hipMalloc(&partial, sizeof(c64) * 256 * 2));
  c64 *x = buf, *r = buf + n, *r0 = buf + 2 * n, *p = buf + 3 * n, *v = buf + 4 * n, *sv = buf + 5 * n, *t = buf + 6 * n, *u = buf + 7 * n, *q = buf + 8 * n;
  KR_HIP(hipMemset(buf, 0, sizeof(c64) * (size_t)n * 9));
  KR_HIP(hipMemcpy(r, b_host, sizeof(c64) * (size_t)n, hipMemcpyHostToDevice));
  KR_HIP(hipMemcpy(r0, r, sizeof(c64) * (size_t)n, hipMemcpyDeviceToDevice));
  auto dot = [&](const c64* a, const c64* bb, cplx* out) -> int {              // <a, b> = sum conj(a_i) b_i
    int rc = op_launch_dot(n, a, bb, 0, partial, scal, st); if (rc) return rc;
    c64 h; MA_HIP(hipMemcpy(&h, scal, sizeof(c64), hipMemcpyDeviceToHost)); *out = cplx(h.re, h.im); return MA_OK;
  };
  auto norm = [&](const c64* a, double* out) -> int {
    int rc = op_launch_dot(n, a, nullptr, 1, partial, scal, st); if (rc) return rc;
    c64 h; MA_HIP(hipMemcpy(&h, scal, sizeof(c64), hipMemcpyDeviceToHost)); *out = h.re; return MA_OK;
  };
  // two inner products against one vector, <v0, y> and <v1, y> with v0, v1 adjacent in `buf`, in one exchange with the host (each reduced
  // exactly as `dot` reduces it)
  auto dot2 = [&](const c64* v01, const c64* yy, cplx* o0, cplx* o1) -> int {
    int rc = op_launch_multi_dot(n, v01, 2, yy, partial, scal, st); if (rc) return rc;
    c64 h[2]; MA_HIP(hipMemcpy(h, scal, 2 * sizeof(c64), hipMemcpyDeviceToHost)); *o0 = cplx(h[0].re, h[0].im); *o1 = cplx(h[1].re, h[1].im); return MA_OK;
  };
  auto axpby = [&](cplx a, const c64* xa, cplx bcoef, const c64* ya, c64* out) { return op_launch_axpby(n, a.real(), a.imag(), xa, bcoef.real(), bcoef.imag(), ya, out, st); };
  auto finish = [&](int iters, double res, bool conv) -> int {
    info->iterations = iters; info->restarts = 0; info->residual = res; info->converged = conv ? 1 : 0;
    hipError_t e = hipMemcpy(x_out, x, sizeof(c64) * (size_t)n, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) { set_error("copy back failed: %s", hipGetErrorString(e)); return MA_ERR_HIP; }
    return spin_error_check("Krylov solve");                // a sweep of a preconditioner that abandoned a wait: never MA_OK
  };
  double b_norm = 0.0, rn = 0.0;
  KR_RC(norm(r, &b_norm));
  if (b_norm < 1e-15) return finish(0, 0.0, true);
  const cplx one(1.0, 0.0);
  if (kind == 0) {                                                             // bicgstab.rs:46-182
    cplx rho = one, alpha = one, omega = one;
    cplx rho_next; KR_RC(dot(r0, r, &rho_next));                               // <r0, r> of the coming iteration (later ones arrive with |r|)
    for (int it = 0; it < max_iterations; ++it) {
      const cplx rho_new = rho_next;
      if (std::abs(rho_new) < 1e-30) { KR_RC(norm(r, &rn)); return finish(it, rn / b_norm, false); }
      const cplx beta = (rho_new / rho) * (alpha / omega);
      rho = rho_new;
      KR_RC(axpby(one, p, -omega, v, t));                                      // p = r + beta (p - omega v)
      KR_RC(axpby(one, r, beta, t, p));
      KR_RC(ma_op_apply_dev(o, p, v, st));
      cplx r0v; KR_RC(dot(r0, v, &r0v));
      if (std::abs(r0v) < 1e-30) { KR_RC(norm(r, &rn)); return finish(it, rn / b_norm, false); }
      alpha = rho / r0v;
      KR_RC(axpby(one, r, -alpha, v, sv));                                     // s = r - alpha v
      double s_norm; KR_RC(norm(sv, &s_norm));
      if (s_norm / b_norm < tol) { KR_RC(op_launch_axpy_host(n, alpha.real(), alpha.imag(), p, x, st)); return finish(it + 1, s_norm / b_norm, true); }
      KR_RC(ma_op_apply_dev(o, sv, t, st));
      cplx st_, tt; KR_RC(dot2(sv, t, &st_, &tt));                             // <s, t> and <t, t> together (s and t are adjacent); <t, s> = conj <s, t>
      if (std::abs(tt) < 1e-30) { KR_RC(norm(r, &rn)); return finish(it, rn / b_norm, false); }
      const cplx ts = std::conj(st_);
      omega = ts / tt;
      KR_RC(op_launch_axpy_host(n, alpha.real(), alpha.imag(), p, x, st));     // x = x + alpha p + omega s
      KR_RC(op_launch_axpy_host(n, omega.real(), omega.imag(), sv, x, st));
      KR_RC(axpby(one, sv, -omega, t, r));                                     // r = s - omega t
      cplx rr; KR_RC(dot2(r, r, &rr, &rho_next));                              // <r, r> and <r0, r> together (r and r0 are adjacent)
      rn = std::sqrt(rr.real());
      const double rel = rn / b_norm;
      if (rel < tol) return finish(it + 1, rel, true);
      if (std::abs(omega) < 1e-30) return finish(it + 1, rel, false);
    }
    KR_RC(norm(r, &rn));
    return finish(max_iterations, rn / b_norm, false);
  }
  if (kind == 1) {                                                             // cgs.rs:46-139
    cplx rho; KR_RC(dot(r0, r, &rho));
    KR_HIP(hipMemcpy(p, r, sizeof(c64) * (size_t)n, hipMemcpyDeviceToDevice));
    KR_HIP(hipMemcpy(u, r, sizeof(c64) * (size_t)n, hipMemcpyDeviceToDevice));
    for (int it = 0; it < max_iterations; ++it) {
      KR_RC(ma_op_apply_dev(o, p, v, st));
      cplx sigma; KR_RC(dot(r0, v, &sigma));
      if (std::abs(sigma) < 1e-30) { KR_RC(norm(r, &rn)); return finish(it, rn / b_norm, false); }
      const cplx alpha = rho / sigma;
      KR_RC(axpby(one, u, -alpha, v, q));                                      // q = u - alpha v
      KR_RC(axpby(one, u, one, q, sv));                                        // u + q
      KR_RC(ma_op_apply_dev(o, sv, t, st));                                    // w = A (u + q)
      KR_RC(op_launch_axpy_host(n, alpha.real(), alpha.imag(), sv, x, st));
      KR_RC(op_launch_axpy_host(n, -alpha.real(), -alpha.imag(), t, r, st));
      cplx rr, rho_new; KR_RC(dot2(r, r, &rr, &rho_new));                      // |r|^2 and <r0, r> together (r and r0 are adjacent)
      rn = std::sqrt(rr.real());
      const double rel = rn / b_norm;
      if (rel < tol) return finish(it + 1, rel, true);
      if (std::abs(rho) < 1e-30) return finish(it + 1, rel, false);
      const cplx beta = rho_new / rho;
      rho = rho_new;
      KR_RC(axpby(one, r, beta, q, u));                                        // u = r + beta q
      KR_RC(axpby(one, q, beta, p, sv));                                       // p = u + beta (q + beta p)
      KR_RC(axpby(one, u, beta, sv, p));
    }
    KR_RC(norm(r, &rn));
    return finish(max_iterations, rn / b_norm, false);
  }
  // cg.rs:49-138 (the reference's <r, r> and <p, q> are the conjugated inner products as well)
  KR_HIP(hipMemcpy(p, r, sizeof(c64) * (size_t)n, hipMemcpyDeviceToDevice));
  cplx rho; KR_RC(dot(r, r, &rho));
  for (int it = 0; it < max_iterations; ++it) {
    KR_RC(ma_op_apply_dev(o, p, q, st));
    cplx pq; KR_RC(dot(p, q, &pq));
    if (std::abs(pq) < 1e-30) { KR_RC(norm(r, &rn)); return finish(it, rn / b_norm, false); }
    const cplx alpha = rho / pq;
    KR_RC(op_launch_axpy_host(n, alpha.real(), alpha.imag(), p, x, st));
    KR_RC(op_launch_axpy_host(n, -alpha.real(), -alpha.imag(), q, r, st));
    KR_RC(norm(r, &rn));
    const double rel = rn / b_norm;
    if (rel < tol) return finish(it + 1, rel, true);
    cplx rho_new; KR_RC(dot(r, r, &rho_new));
    if (std::abs(rho) < 1e-30) return finish(it + 1, rel, false);
    const cplx beta = rho_new / rho;
    rho = rho_new;
    KR_RC(axpby(one, r, beta, p, p));                                          // p = r + beta p
  }
  KR_RC(norm(r, &rn));
  return finish(max_iterations, rn / b_norm, false);
#undef KR_HIP
#undef KR_RC
}

extern "C" {
int ma_bicgstab(ma_op_t* o, const ma_c64* b, int32_t max_iterations, double tol, ma_c64* x_out, ma_gmres_info_t* info) { return krylov_impl(0, o, b, max_iterations, tol, x_out, info); }
int ma_cgs(ma_op_t* o, const ma_c64* b, int32_t max_iterations, double tol, ma_c64* x_out, ma_gmres_info_t* info) { return krylov_impl(1, o, b, max_iterations, tol, x_out, info); }
int ma_cg(ma_op_t* o, const ma_c64* b, int32_t max_iterations, double tol, ma_c64* x_out, ma_gmres_info_t* info) { return krylov_impl(2, o, b, max_iterations, tol, x_out, info); }
}  // extern "C"
