// test_host_mirror.cpp — the reference's own unit tests for the seam functions, restated against
// the C++ host mirror (math-solvers/src/direct/lu.rs:163-240; math-bem/src/core/assembly/tbem.rs:545-598;
// math-bem/bin/qa_suite.rs:199-326 at ka = 0.2). Exit code 0 = all passed. Needs an MI355X.
#include <array>
#include <cstdio>
#include "../../math_audio_amd/host/math_audio.hpp"

using math_solvers::Complex64;
static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static void test_lu_solve_real() {                      // lu.rs:163-175
  std::vector<double> a = {4.0, 1.0, 1.0, 3.0}, b = {1.0, 2.0};
  auto x = math_solvers::lu_solve(a, 2, 2, b).expect("LU solve should succeed");
  for (int i = 0; i < 2; ++i) { Complex64 ax = a[2 * i] * x[0] + a[2 * i + 1] * x[1]; CHECK(std::abs(ax - b[i]) < 1e-10); }
}
static void test_lu_solve_complex() {                   // lu.rs:177-193
  std::vector<Complex64> a = {{4, 1}, {1, 0}, {1, 0}, {3, -1}}, b = {{1, 1}, {2, -1}};
  auto x = math_solvers::lu_solve(a, 2, 2, b).expect("LU solve should succeed");
  for (int i = 0; i < 2; ++i) CHECK(std::abs(a[2 * i] * x[0] + a[2 * i + 1] * x[1] - b[i]) < 1e-10);
}
static void test_lu_identity() {                        // lu.rs:195-206
  const int n = 5; std::vector<double> a(n * n, 0.0), b(n);
  for (int i = 0; i < n; ++i) { a[i * n + i] = 1.0; b[i] = i + 1.0; }
  auto x = math_solvers::lu_solve(a, n, n, b).expect("LU solve should succeed");
  for (int i = 0; i < n; ++i) CHECK(std::abs(x[i] - b[i]) < 1e-10);
}
static void test_lu_singular() {                        // lu.rs:208-216
  std::vector<double> a = {1.0, 2.0, 2.0, 4.0}, b = {1.0, 2.0};
  auto r = math_solvers::lu_solve(a, 2, 2, b);
  CHECK(r.is_err()); CHECK(r.err.kind == math_solvers::LuError::SingularMatrix);
}
static void test_lu_dimension_mismatch() {
  std::vector<double> a = {1.0, 0.0, 0.0, 1.0}, b = {1.0, 2.0, 3.0};
  auto r = math_solvers::lu_solve(a, 2, 2, b);
  CHECK(r.is_err()); CHECK(r.err.kind == math_solvers::LuError::DimensionMismatch);
}
static void test_lu_factorize_then_solve() {             // lu.rs:218-240 (factorize once, solve)
  std::vector<Complex64> a = {{4, 1}, {1, 0}, {2, 0}, {1, 0}, {3, -1}, {0, 1}, {2, 0}, {0, 1}, {5, 0}};
  auto f = math_solvers::lu_factorize(a, 3, 3);
  CHECK(f.is_ok());
  for (int t = 0; t < 2; ++t) {
    std::vector<Complex64> b = {{1.0 + t, 1}, {2, -1.0 * t}, {0.5, 0.25}};
    auto x = f.expect("factorization").solve(b).expect("solve");
    for (int i = 0; i < 3; ++i) CHECK(std::abs(a[3 * i] * x[0] + a[3 * i + 1] * x[1] + a[3 * i + 2] * x[2] - b[i]) < 1e-10);
  }
  CHECK(math_solvers::lu_factorize({1, 2, 2, 4}, 2, 2).err.kind == math_solvers::LuError::SingularMatrix);
}
static void test_tbem_diagonal_nonzero_and_qa() {       // tbem.rs:585-598 + qa_suite.rs:199-326 (Rayleigh, ka = 0.2)
  using namespace math_bem;
  const double radius = 0.1, c = 343.0, rho = 1.21, ka = 0.2;
  const double k = ka / radius, freq = k * c / (2.0 * 3.14159265358979323846);
  PhysicsParams physics(freq, c, rho, false);
  Mesh mesh = generate_icosphere_mesh(radius, 2);
  CHECK(mesh.elements.size() == 320); CHECK(mesh.num_nodes() == 162);
  auto beta = physics.burton_miller_beta_adaptive(radius).first;
  TbemSystem sys = build_tbem_system_with_beta(mesh.elements, mesh.nodes, physics, beta);
  CHECK(sys.num_dofs == 320);
  for (size_t i = 0; i < sys.num_dofs; ++i) CHECK(std::abs(sys.at(i, i)) > 1e-15);
  std::vector<double> centers, normals;
  for (auto& e : mesh.elements) for (int d = 0; d < 3; ++d) { centers.push_back(e.center[d]); normals.push_back(e.normal[d]); }
  auto rhs = IncidentField::plane_wave_z().compute_rhs_with_beta(centers, normals, physics, beta);
  for (size_t i = 0; i < rhs.size(); ++i) rhs[i] += sys.rhs[i];
  auto p = math_solvers::lu_solve(sys.matrix, sys.num_dofs, sys.num_dofs, rhs).expect("LU solve should succeed");
  // Rayleigh regime: the total surface pressure of a small rigid sphere is close to the incident plane wave
  // (first-order: p = p_inc (1 + 1.5 i k a cos(theta)) ), well inside the QA suite's 5 % L2 threshold.
  double num = 0.0, den = 0.0;
  for (size_t i = 0; i < p.size(); ++i) {
    const double z = mesh.elements[i].center[2], r = std::sqrt(centers[3 * i] * centers[3 * i] + centers[3 * i + 1] * centers[3 * i + 1] + z * z);
    const Complex64 approx = Complex64(1.0, 1.5 * physics.wave_number * r * (z / r));
    num += std::norm(p[i] - approx); den += std::norm(approx);
  }
  CHECK(std::sqrt(num / den) < 0.05);
  // error behaviour: an element that references a node outside the mesh is refused before anything is computed
  Mesh bad = mesh; bad.elements[0].connectivity[1] = 100000;
  bool threw = false;
  try { build_tbem_system_with_beta(bad.elements, bad.nodes, physics, beta); } catch (const BemError& e) { threw = e.status == MA_ERR_INVALID; }
  CHECK(threw);
}

// ---- math-solvers/src/sparse/csr.rs:659-760, preconditioners/diagonal.rs:106-145, iterative/gmres.rs:632-705
static void test_csr_from_dense_matvec_triplets() {
  using namespace math_solvers;
  std::vector<Complex64> d = {1, 0, 2, 0, 3, 0, 4, 0, 5};
  CsrMatrix csr = CsrMatrix::from_dense(d, 3, 3, 1e-15);
  CHECK(csr.num_rows() == 3 && csr.num_cols() == 3 && csr.nnz() == 5);
  CHECK(csr.get(0, 0).real() == 1.0 && csr.get(0, 2).real() == 2.0 && csr.get(1, 1).real() == 3.0 && csr.get(2, 0).real() == 4.0 && csr.get(2, 2).real() == 5.0);
  CsrMatrix m2 = CsrMatrix::from_dense({1, 2, 3, 4}, 2, 2, 1e-15);
  auto y = m2.matvec({Complex64(1, 0), Complex64(2, 0)});                    // [1 2; 3 4] [1 2]^T = [5 11]^T
  CHECK(std::abs(y[0] - Complex64(5, 0)) < 1e-10 && std::abs(y[1] - Complex64(11, 0)) < 1e-10);
  CsrMatrix t = CsrMatrix::from_triplets(3, 3, {{0, 0, {1, 0}}, {0, 2, {2, 0}}, {1, 1, {3, 0}}, {2, 0, {4, 0}}, {2, 2, {5, 0}}});
  CHECK(t.nnz() == 5 && t.get(0, 0).real() == 1.0 && t.get(1, 1).real() == 3.0);
  CsrMatrix dup = CsrMatrix::from_triplets(2, 2, {{0, 0, {1, 0}}, {0, 0, {2, 0}}, {1, 1, {3, 0}}});
  CHECK(dup.get(0, 0).real() == 3.0);                                        // 1 + 2 = 3
  auto yt = m2.apply_transpose({Complex64(1, 0), Complex64(2, 0)});          // [1 3; 2 4] [1 2]^T = [7 10]^T
  CHECK(std::abs(yt[0] - Complex64(7, 0)) < 1e-10 && std::abs(yt[1] - Complex64(10, 0)) < 1e-10);
}
static void test_diagonal_preconditioner() {
  using namespace math_solvers;
  CsrMatrix m = CsrMatrix::from_dense({4, 1, 1, 2}, 2, 2, 1e-15);
  auto p = DiagonalPreconditioner::from_csr(m);
  auto z = p.apply({Complex64(4, 0), Complex64(4, 0)});
  CHECK(std::abs(z[0] - Complex64(1, 0)) < 1e-10 && std::abs(z[1] - Complex64(2, 0)) < 1e-10);
}
static void test_gmres_simple_identity_preconditioned() {
  using namespace math_solvers;
  CsrMatrix a = CsrMatrix::from_dense({4, 1, 1, 3}, 2, 2, 1e-15);
  std::vector<Complex64> b = {{1, 0}, {2, 0}};
  GmresConfig config{100, 10, 1e-10, 0};
  auto sol = gmres(a, b, config);
  CHECK(sol.converged);
  auto ax = a.matvec(sol.x);
  CHECK(std::sqrt(std::norm(ax[0] - b[0]) + std::norm(ax[1] - b[1])) < 1e-8);
  const size_t n = 5;
  CsrMatrix id = CsrMatrix::identity(n);
  std::vector<Complex64> bi; for (size_t i = 1; i <= n; ++i) bi.push_back(Complex64((double)i, 0.0));
  auto s2 = gmres(id, bi, GmresConfig{10, 10, 1e-12, 0});
  CHECK(s2.converged && s2.iterations <= 2);
  double e = 0.0; for (size_t i = 0; i < n; ++i) e += std::norm(s2.x[i] - bi[i]);
  CHECK(std::sqrt(e) < 1e-10);
  auto p = DiagonalPreconditioner::from_csr(a);
  auto s3 = gmres_preconditioned(a, p, b, config);
  CHECK(s3.converged);
  auto ax3 = a.matvec(s3.x);
  CHECK(std::sqrt(std::norm(ax3[0] - b[0]) + std::norm(ax3[1] - b[1])) < 1e-8);
  DenseOperator dn({4, 1, 1, 3}, 2);                                         // DenseOperator, fmm_interface.rs:25-53
  auto s4 = gmres_with_guess(dn, b, &sol.x, config);
  CHECK(s4.converged && s4.iterations <= 1);
}

// ---- math-fem/src/multigrid/smoother.rs:192-237 (the reference's own smoother tests, on a 1D Laplacian given as COO triplets with
// the diagonal split in two) and amg.rs's symmetric Gauss-Seidel on the CSR form
static void test_fem_smoothers_reduce_residual() {
  using namespace math_fem;
  const size_t n = 40;
  std::vector<int64_t> r, c; std::vector<Complex64> v;
  for (size_t i = 0; i < n; ++i) {
    r.push_back((int64_t)i); c.push_back((int64_t)i); v.push_back(Complex64(1.0, 0.0));
    r.push_back((int64_t)i); c.push_back((int64_t)i); v.push_back(Complex64(1.0, 0.0));      // duplicates are summed (helmholtz.rs:22-33)
    if (i > 0) { r.push_back((int64_t)i); c.push_back((int64_t)i - 1); v.push_back(Complex64(-1.0, 0.0)); }
    if (i + 1 < n) { r.push_back((int64_t)i); c.push_back((int64_t)i + 1); v.push_back(Complex64(-1.0, 0.0)); }
  }
  HelmholtzMatrix m(n, r, c, v);
  std::vector<Complex64> b(n); for (size_t i = 0; i < n; ++i) b[i] = Complex64(std::sin(0.3 * (double)i), 0.0);
  auto norm = [](const std::vector<Complex64>& z) { double s = 0; for (auto& e : z) s += std::norm(e); return std::sqrt(s); };
  for (SmootherType t : {SmootherType::GaussSeidel, SmootherType::Jacobi, SmootherType::SymmetricGaussSeidel}) {
    std::vector<Complex64> x(n, Complex64(0.0, 0.0));
    const double r0 = norm(compute_residual(m, x, b));
    SmootherConfig cfg; cfg.smoother_type = t; if (t == SmootherType::Jacobi) { cfg.iterations = 5; cfg.omega = 0.6; }
    smooth(m, x, b, cfg);
    CHECK(norm(compute_residual(m, x, b)) < r0);
  }
  // one forward Gauss-Seidel sweep of the 1D Laplacian from x = 0 is the recurrence x_i = (b_i + x_{i-1}) / 2
  std::vector<Complex64> x(n, Complex64(0.0, 0.0));
  SmootherConfig one; one.iterations = 1;
  smooth(m, x, b, one);
  Complex64 prev(0.0, 0.0); double e = 0.0;
  for (size_t i = 0; i < n; ++i) { prev = (b[i] + prev) / 2.0; e = std::max(e, std::abs(x[i] - prev)); }
  CHECK(e < 1e-14);
  std::vector<std::tuple<size_t, size_t, Complex64>> trip;
  for (size_t k = 0; k < v.size(); ++k) trip.emplace_back((size_t)r[k], (size_t)c[k], v[k]);
  math_solvers::CsrMatrix a = math_solvers::CsrMatrix::from_triplets(n, n, trip);
  std::vector<Complex64> xs(n, Complex64(0.0, 0.0));
  const double r0 = norm(b);
  math_solvers::smooth_sym_gauss_seidel(a, xs, b, 2);
  auto ax = a.matvec(xs); for (size_t i = 0; i < n; ++i) ax[i] = b[i] - ax[i];
  CHECK(norm(ax) < r0);
}

// ---- round 2: AMG cycle + pipelined GMRES (amg.rs:981-1103, gmres_pipelined.rs:258-285), the operators over a mesh, the sweep
static void test_amg_and_pipelined_gmres() {
  using namespace math_solvers;
  // gmres_pipelined.rs:258-285: the 2 x 2 system
  CsrMatrix a2 = CsrMatrix::from_dense({4, 1, 1, 3}, 2, 2, 1e-15);
  std::vector<Complex64> b2 = {{1, 0}, {2, 0}};
  auto sp = gmres_pipelined(a2, nullptr, b2, nullptr, GmresConfig{100, 10, 1e-10, 0});
  CHECK(sp.converged);
  auto ax = a2.matvec(sp.x);
  CHECK(std::sqrt(std::norm(ax[0] - b2[0]) + std::norm(ax[1] - b2[1])) < 1e-8);
  // bicgstab.rs:190-219, cgs.rs:151-180, cg.rs:146-189
  for (int which = 0; which < 3; ++which) {
    KrylovConfig kc{100, 1e-10, 0};
    auto ks = which == 0 ? bicgstab(a2, b2, kc) : (which == 1 ? cgs(a2, b2, kc) : cg(a2, b2, kc));
    CHECK(ks.converged);
    auto kax = a2.matvec(ks.x);
    CHECK(std::sqrt(std::norm(kax[0] - b2[0]) + std::norm(kax[1] - b2[1])) < 1e-8);
  }
  {
    CsrMatrix id5 = CsrMatrix::identity(5);
    std::vector<Complex64> b5; for (int i = 1; i <= 5; ++i) b5.push_back(Complex64((double)i, 0.0));
    auto cs5 = cg(id5, b5, KrylovConfig{10, 1e-12, 0});
    double e5 = 0.0; for (size_t i = 0; i < 5; ++i) e5 += std::norm(cs5.x[i] - b5[i]);
    CHECK(cs5.converged && cs5.iterations <= 2 && std::sqrt(e5) < 1e-10);
  }
  {                                                          // ilu.rs:177-224: the 3 x 3 tridiagonal system
    CsrMatrix t3 = CsrMatrix::from_dense({4, -1, 0, -1, 4, -1, 0, -1, 4}, 3, 3, 1e-15);
    auto ilu = IluPreconditioner::from_csr(t3);
    {                                                      // ilu_parallel.rs / schwarz.rs through the mirror: each one preconditions GMRES to convergence
      std::vector<Complex64> bb(t3.num_rows()); for (size_t q = 0; q < bb.size(); ++q) bb[q] = Complex64(std::sin((double)q), 0.25);
      auto col = IluColoringPreconditioner::from_csr(t3);
      auto fp = IluFixedPointPreconditioner::from_csr_default(t3);
      auto sw = AdditiveSchwarzPreconditioner::from_csr(t3, 2, 1);
      CHECK(sw.stats().num_subdomains == 2 && sw.stats().min_size > 0 && sw.stats().max_size >= sw.stats().min_size);
      CHECK(gmres_preconditioned(t3, col, bb, GmresConfig{100, 20, 1e-10, 0}).converged);
      CHECK(gmres_preconditioned(t3, fp, bb, GmresConfig{100, 20, 1e-10, 0}).converged);
      CHECK(gmres_preconditioned(t3, sw, bb, GmresConfig{100, 20, 1e-10, 0}).converged);
    }
    std::vector<Complex64> r3 = {{1, 0}, {2, 0}, {3, 0}};
    auto chk = t3.matvec(ilu.apply(r3));
    CHECK(std::abs(chk[0] - r3[0]) < 0.5 && std::abs(chk[1] - r3[1]) < 0.5 && std::abs(chk[2] - r3[2]) < 0.5);
    auto sg = gmres_preconditioned(t3, ilu, r3, GmresConfig{50, 10, 1e-10, 0});
    CHECK(sg.converged);
  }
  // two-level hierarchy of the 1D Laplacian: aggregates of two, P piecewise constant, R = P^T, A_c = R A P
  const size_t n = 64, nc = n / 2;
  std::vector<std::tuple<size_t, size_t, Complex64>> ta, tp, tr, tc;
  for (size_t i = 0; i < n; ++i) {
    ta.emplace_back(i, i, Complex64(2.0, 0.1));
    if (i > 0) ta.emplace_back(i, i - 1, Complex64(-1.0, 0.0));
    if (i + 1 < n) ta.emplace_back(i, i + 1, Complex64(-1.0, 0.0));
    tp.emplace_back(i, i / 2, Complex64(1.0, 0.0)); tr.emplace_back(i / 2, i, Complex64(1.0, 0.0));
  }
  for (size_t I = 0; I < nc; ++I) {                        // R A P of the tridiagonal matrix: (2 (2 + 0.1i) - 2) on the diagonal, -1 beside it
    tc.emplace_back(I, I, Complex64(2.0, 0.2));
    if (I > 0) tc.emplace_back(I, I - 1, Complex64(-1.0, 0.0));
    if (I + 1 < nc) tc.emplace_back(I, I + 1, Complex64(-1.0, 0.0));
  }
  CsrMatrix A = CsrMatrix::from_triplets(n, n, ta), P = CsrMatrix::from_triplets(n, nc, tp), R = CsrMatrix::from_triplets(nc, n, tr), Ac = CsrMatrix::from_triplets(nc, nc, tc);
  AmgConfig cfg; cfg.num_pre_smooth = 2; cfg.num_post_smooth = 2;
  AmgPreconditioner amg({AmgLevel{&A, &P, &R}, AmgLevel{&Ac, nullptr, nullptr}}, cfg);
  std::vector<Complex64> b(n); for (size_t i = 0; i < n; ++i) b[i] = Complex64(std::sin(0.2 * (double)i), std::cos(0.1 * (double)i));
  auto norm = [](const std::vector<Complex64>& z) { double s = 0; for (auto& e : z) s += std::norm(e); return std::sqrt(s); };
  auto z = amg.apply(b);                                   // one V-cycle from zero is a contraction: ||b - A z|| < ||b||
  auto az = A.matvec(z); for (size_t i = 0; i < n; ++i) az[i] = b[i] - az[i];
  CHECK(norm(az) < norm(b));
  auto s_amg = gmres_preconditioned(A, amg, b, GmresConfig{200, 30, 1e-10, 0});
  auto s_plain = gmres(A, b, GmresConfig{200, 30, 1e-10, 0});
  CHECK(s_amg.converged && s_plain.converged && s_amg.iterations < s_plain.iterations);
  auto s_pp = gmres_pipelined(A, &amg, b, nullptr, GmresConfig{200, 30, 1e-10, 0});
  CHECK(s_pp.converged);
  {                                                        // AmgPreconditioner::from_csr: amg.rs' own tests (:1158-1266) through the mirror
    std::vector<std::tuple<size_t, size_t, Complex64>> tl;
    const size_t m = 100;
    for (size_t i = 0; i < m; ++i) { tl.emplace_back(i, i, Complex64(2.0, 0.0)); if (i > 0) tl.emplace_back(i, i - 1, Complex64(-1.0, 0.0)); if (i + 1 < m) tl.emplace_back(i, i + 1, Complex64(-1.0, 0.0)); }
    CsrMatrix Lap = CsrMatrix::from_triplets(m, m, tl);
    auto built = AmgPreconditioner::from_csr(Lap, AmgConfig());
    CHECK(built.num_levels() >= 2 && built.grid_complexity() >= 1.0 && built.operator_complexity() >= 1.0 && built.setup_time_ms() >= 0.0);
    auto dg = built.diagnostics();
    CHECK(dg.level_dofs.size() == dg.num_levels && dg.level_nnz.size() == dg.num_levels && dg.level_dofs[0] == m && dg.level_nnz[0] == 3 * m - 2);
    AmgConfig pm; pm.coarsening = AmgCoarsening::Pmis;
    CHECK(AmgPreconditioner::from_csr(Lap, pm).num_levels() >= 2);
    std::vector<Complex64> rb(m), xx(m, Complex64(0.0, 0.0));
    for (size_t i = 0; i < m; ++i) rb[i] = Complex64(std::sin((double)i), 0.0);
    const double r0 = norm(rb);
    for (int it = 0; it < 10; ++it) {
      auto ax = Lap.matvec(xx); std::vector<Complex64> rr(m); for (size_t i = 0; i < m; ++i) rr[i] = rb[i] - ax[i];
      auto zz = built.apply(rr); for (size_t i = 0; i < m; ++i) xx[i] += zz[i];
    }
    auto ax = Lap.matvec(xx); std::vector<Complex64> rr(m); for (size_t i = 0; i < m; ++i) rr[i] = rb[i] - ax[i];
    CHECK(norm(rr) < 0.1 * r0);
    auto s_b = gmres_preconditioned(Lap, AmgPreconditioner::from_csr(Lap, AmgConfig::for_parallel()), rb, GmresConfig{200, 30, 1e-10, 0});
    CHECK(s_b.converged);
  }
  auto r = A.matvec(s_pp.x); for (size_t i = 0; i < n; ++i) r[i] = b[i] - r[i];
  CHECK(norm(r) < 1e-7 * norm(b));
}
static void test_mesh_operators_and_sweep() {
  using namespace math_bem;
  Mesh mesh = generate_icosphere_mesh(0.1, 2);             // 320 panels
  const size_t n = mesh.elements.size();
  const double c0 = 343.0, f1 = 500.0;
  PhysicsParams ph(f1, c0, 1.21, false);
  const Complex64 beta = ph.burton_miller_beta_scaled(4.0);
  TbemSystem sys = build_tbem_system_with_beta(mesh.elements, mesh.nodes, ph, beta);
  std::vector<Complex64> x(n); for (size_t i = 0; i < n; ++i) x[i] = Complex64(std::cos(0.37 * (double)i), std::sin(0.11 * (double)i));
  auto dense = [&](const std::vector<Complex64>& v) { std::vector<Complex64> y(n); for (size_t i = 0; i < n; ++i) { Complex64 s(0, 0); for (size_t j = 0; j < n; ++j) s += sys.matrix[i * n + j] * v[j]; y[i] = s; } return y; };
  auto norm = [](const std::vector<Complex64>& z) { double s = 0; for (auto& e : z) s += std::norm(e); return std::sqrt(s); };
  auto diff = [&](const std::vector<Complex64>& a, const std::vector<Complex64>& b) { std::vector<Complex64> d(a.size()); for (size_t i = 0; i < a.size(); ++i) d[i] = a[i] - b[i]; return norm(d); };
  BemPlan plan(mesh.elements, mesh.nodes);
  const auto yd = dense(x);
  TbemOperator op(plan, ph, beta);                         // matrix-free: equals A x of the stored matrix
  CHECK(diff(op.apply(x), yd) < 1e-10 * norm(yd));
  TbemOperator op1(mesh.elements, mesh.nodes, ph, beta, {0});   // the sharded form with one shard
  CHECK(op1.num_shards() == 1 && diff(op1.apply(x), yd) < 1e-10 * norm(yd));
  // SLFMM over grid clusters: <A x, z> = <x, A^T z>, near matrix with a non-zero diagonal (slfmm.rs:790-877)
  std::map<std::array<long, 3>, std::vector<size_t>> cells;
  const double cell = 0.07;
  for (size_t e = 0; e < n; ++e) { std::array<long, 3> k; for (int d = 0; d < 3; ++d) k[d] = (long)std::floor((mesh.elements[e].center[d] + 0.2) / cell); cells[k].push_back(e); }
  std::vector<Cluster> cl; std::vector<std::array<long, 3>> keys;
  for (auto& kv : cells) { Cluster c; for (int d = 0; d < 3; ++d) c.center[d] = -0.2 + ((double)kv.first[d] + 0.5) * cell; c.element_indices = kv.second; cl.push_back(c); keys.push_back(kv.first); }
  for (size_t a = 0; a < cl.size(); ++a) for (size_t b = 0; b < cl.size(); ++b) {
    if (a == b) continue;
    bool nb = true; for (int d = 0; d < 3; ++d) nb = nb && std::labs(keys[a][d] - keys[b][d]) <= 1;
    (nb ? cl[a].near_clusters : cl[a].far_clusters).push_back(b);
  }
  SlfmmSystem fmm(plan, cl, ph, 4, 8, 6);
  std::vector<Complex64> z(n); for (size_t i = 0; i < n; ++i) z[i] = Complex64(std::sin(0.23 * (double)i), 0.5);
  auto Ax = fmm.matvec(x), Atz = fmm.matvec_transpose(z);
  Complex64 l(0, 0), r(0, 0); for (size_t i = 0; i < n; ++i) { l += Ax[i] * z[i]; r += x[i] * Atz[i]; }
  CHECK(std::abs(l - r) < 1e-10 * std::abs(l));
  auto nearm = fmm.extract_near_field_matrix();
  bool diag_ok = true; for (size_t i = 0; i < n; ++i) diag_ok = diag_ok && std::abs(nearm[i * n + i]) > 0.0;
  CHECK(diag_ok);
  // MLFMM (mlfmm.rs:1267-1312): the tree's root holds every element, the system has n dofs, matvec returns n finite entries,
  // and a tree of ONE level (n <= target) is the near field alone: the dense coefficient matrix without the free term
  {
    ClusterTree tree(mesh.elements, mesh.nodes, 20, ph);
    CHECK(tree.num_levels() >= 2 && tree.level(0).clusters.size() == 1 && tree.level(0).clusters[0].element_indices.size() == n);
    MlfmmSystem ml(plan, tree, ph);
    auto ym = ml.matvec(x);
    bool fin = ym.size() == n; for (auto& v : ym) fin = fin && std::isfinite(v.real()) && std::isfinite(v.imag());
    CHECK(fin);
    bool threw = false; try { ml.apply_transpose(x); } catch (const math_solvers::SolverError& e) { threw = e.status == MA_ERR_UNSUPPORTED; }
    CHECK(threw);
    ClusterTree one(mesh.elements, mesh.nodes, 1000, ph);
    CHECK(one.num_levels() == 1);
    MlfmmSystem m1(plan, one, ph);
    // ... which is the single-level operator over ONE cluster minus the free term slfmm.rs adds and mlfmm.rs does not (gamma / 2)
    Cluster all; all.center = {0.0, 0.0, 0.0}; for (size_t e = 0; e < n; ++e) all.element_indices.push_back(e);
    SlfmmSystem s1(plan, {all}, ph, 4, 8, 5);
    auto yref = s1.matvec(x);
    for (size_t i = 0; i < n; ++i) yref[i] -= 0.5 * x[i];
    auto y1 = m1.matvec(x);
    CHECK(diff(y1, yref) < 1e-10 * norm(yref));
  }
  // the sweep: two frequencies in one call against assemble + RHS + lu_solve per frequency
  const std::vector<double> freqs = {300.0, 700.0};
  std::vector<int32_t> status;
  auto sols = solve_frequency_sweep(mesh.elements, mesh.nodes, freqs, c0, 4.0, IncidentField::plane_wave_z(), {0}, &status);
  std::vector<double> centers, normals;
  for (auto& e : mesh.elements) { for (int d = 0; d < 3; ++d) { centers.push_back(e.center[d]); normals.push_back(e.normal[d]); } }
  for (size_t f = 0; f < freqs.size(); ++f) {
    PhysicsParams pf(freqs[f], c0, 1.21, false);
    const Complex64 bf = pf.burton_miller_beta_scaled(4.0);
    TbemSystem sf = build_tbem_system_with_beta(mesh.elements, mesh.nodes, pf, bf);
    auto rhs = IncidentField::plane_wave_z().compute_rhs_with_beta(centers, normals, pf, bf);
    for (size_t i = 0; i < n; ++i) rhs[i] += sf.rhs[i];
    auto ref = math_solvers::lu_solve(sf.matrix, n, n, rhs);
    CHECK(ref.is_ok() && status[f] == MA_OK && diff(sols[f], ref.value) < 1e-9 * norm(ref.value));
  }
  // round 4: the loop behind a handle that lives beside the mesh (ma_bem_sweep_t): two sweeps through ONE handle, the second shorter.
  // The same frequencies through the one-call form: the same bits (same kernels on the same data); a sweep with another number of
  // slots may assemble one, two or three systems per pass over the quadrature points, which moves the last bits of the far entries
  {
    FrequencySweep sw(mesh.elements, mesh.nodes, 4);
    CHECK(sw.num_dofs() == n);
    const std::vector<double> f4 = {300.0, 700.0, 1100.0, 250.0};
    std::vector<int32_t> st4;
    auto s4 = sw.solve(f4, c0, 4.0, IncidentField::plane_wave_z(), &st4);
    auto s4_ref = solve_frequency_sweep(mesh.elements, mesh.nodes, f4, c0, 4.0, IncidentField::plane_wave_z(), {0});
    for (size_t f = 0; f < f4.size(); ++f) CHECK(st4[f] == MA_OK && s4[f] == s4_ref[f]);
    CHECK(diff(s4[0], sols[0]) < 1e-12 * norm(sols[0]) && diff(s4[1], sols[1]) < 1e-12 * norm(sols[1]));
    auto s2 = sw.solve(freqs, c0, 4.0, IncidentField::plane_wave_z());
    CHECK(s2.size() == 2 && s2[0] == s4[0] && s2[1] == s4[1]);          // the same handle, the same systems: the same bits
    bool threw = false;
    try { sw.solve({100.0, 200.0, 300.0, 400.0, 500.0}, c0, 4.0, IncidentField::plane_wave_z()); } catch (const BemError& e) { threw = e.status == MA_ERR_INVALID; }
    CHECK(threw);                                            // more frequencies than the handle was made for
  }
}

int main() {
  int n = 0;
  if (ma_device_count(&n) != MA_OK || n <= 0) { std::printf("no HIP device: the host mirror has no CPU fallback\n"); return 77; }
  test_lu_solve_real(); test_lu_solve_complex(); test_lu_identity(); test_lu_singular(); test_lu_dimension_mismatch(); test_lu_factorize_then_solve();
  test_tbem_diagonal_nonzero_and_qa();
  test_csr_from_dense_matvec_triplets(); test_diagonal_preconditioner(); test_gmres_simple_identity_preconditioned();
  test_fem_smoothers_reduce_residual();
  test_amg_and_pipelined_gmres();
  test_mesh_operators_and_sweep();
  std::printf(failures ? "%d check(s) failed\n" : "host mirror: all checks passed\n", failures);
  return failures ? 1 : 0;
}
