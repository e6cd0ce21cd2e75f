// math_audio.hpp — C++ host mirror of the reference's Rust API for the hot path, on top of the C-ABI.
//
// The reference is compiled code (Rust); this image has no Rust toolchain, so the host side above
// include/mathaudio_hip.h is C++ with the reference's names, argument meaning and error behaviour:
//   math_bem::PhysicsParams / Element / Mesh / generate_icosphere_mesh / build_tbem_system_with_beta /
//            IncidentField::compute_rhs_with_beta        (math-bem/src/core/{types,mesh/generators,assembly/tbem,incident}.rs)
//   math_solvers::lu_solve -> Result (LuError)            (math-solvers/src/direct/lu.rs)
//   math_solvers::CsrMatrix / LinearOperator / DenseOperator / DiagonalPreconditioner / GmresConfig / GmresSolution /
//            gmres / gmres_with_guess / gmres_preconditioned   (math-solvers/src/{sparse/csr,traits,iterative/gmres,preconditioners/diagonal}.rs)
//   math_solvers::AmgPreconditioner (V/W/F cycle over a hierarchy the caller built) / gmres_pipelined   (preconditioners/amg.rs, iterative/gmres_pipelined.rs)
//   math_bem::Cluster / SlfmmSystem (build_slfmm_system + its LinearOperator) / ClusterTree + MlfmmSystem (build_cluster_tree, build_mlfmm_system) / TbemOperator (matrix-free, one or several GPUs) /
//            solve_frequency_sweep (the `for freq` loop of bin/room_simulator_bem.rs:329 as one call, one or several GPUs) / FrequencySweep (the same behind a reusable handle)
// Header-only; links against libmathaudio_hip.so. No CPU fallback: errors come back as exceptions or
// Result values carrying the C status code.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <complex>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#include "../../include/mathaudio_hip.h"

namespace math_bem {

using Complex64 = std::complex<double>;
static_assert(sizeof(Complex64) == sizeof(ma_c64), "Complex64 layout");

struct BemError : std::runtime_error {
  int status;
  BemError(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

// types.rs:16-219
struct PhysicsParams {
  double speed_of_sound, density, frequency, wave_number, omega, wave_length, harmonic_factor, pressure_factor, tau;
  PhysicsParams(double f, double c, double rho, bool is_internal) {
    const double PI = 3.14159265358979323846;
    omega = 2.0 * PI * f; wave_number = omega / c; wave_length = c / f; harmonic_factor = 1.0;
    tau = is_internal ? -1.0 : 1.0; speed_of_sound = c; density = rho; frequency = f; pressure_factor = rho * omega * harmonic_factor;
  }
  double gamma() const { return 1.0; }
  Complex64 burton_miller_beta() const { return tau > 0.0 ? Complex64(0.0, harmonic_factor / wave_number) : Complex64(0.0, 0.0); }
  Complex64 burton_miller_beta_scaled(double s) const { return tau > 0.0 ? Complex64(0.0, harmonic_factor * s / wave_number) : Complex64(0.0, 0.0); }
  std::pair<Complex64, double> burton_miller_beta_adaptive(double radius) const {
    if (tau <= 0.0) return {Complex64(0.0, 0.0), 1.0};
    const double ka = wave_number * radius;
    const double scale = ka < 0.5 ? 1.0 : (ka < 1.2 ? 4.0 : (ka < 1.8 ? 8.0 : 16.0));
    return {Complex64(0.0, harmonic_factor * scale / wave_number), scale};
  }
};

enum class ElementType { Tri3, Quad4 };
enum class ElementProperty { Surface = 0, MidFace = 1, Evaluation = 2 };
struct BoundaryCondition {
  enum Kind { Velocity, Pressure, Other } kind = Velocity;
  std::vector<Complex64> values{Complex64(0.0, 0.0)};
};

// types.rs:330-351
struct Element {
  std::vector<size_t> connectivity;
  ElementType element_type = ElementType::Tri3;
  ElementProperty property = ElementProperty::Surface;
  double normal[3] = {0, 0, 0};
  double center[3] = {0, 0, 0};
  double area = 0.0;
  BoundaryCondition boundary_condition;
  std::vector<size_t> dof_addresses;
};

struct Mesh {
  std::vector<double> nodes;          // n_nodes x 3 row-major (Array2<f64>)
  std::vector<Element> elements;
  size_t num_nodes() const { return nodes.size() / 3; }
};

// generators.rs:513-602
inline void compute_element_geometry(Element& e, const std::vector<double>& nodes) {
  const size_t n = e.connectivity.size();
  double c[3] = {0, 0, 0};
  for (size_t a = 0; a < n; ++a) for (int d = 0; d < 3; ++d) c[d] += nodes[3 * e.connectivity[a] + d];
  for (int d = 0; d < 3; ++d) c[d] /= (double)n;
  double v1[3], v2[3];
  if (n == 3) { for (int d = 0; d < 3; ++d) { v1[d] = nodes[3 * e.connectivity[1] + d] - nodes[3 * e.connectivity[0] + d]; v2[d] = nodes[3 * e.connectivity[2] + d] - nodes[3 * e.connectivity[0] + d]; } }
  else { for (int d = 0; d < 3; ++d) { v1[d] = nodes[3 * e.connectivity[2] + d] - nodes[3 * e.connectivity[0] + d]; v2[d] = nodes[3 * e.connectivity[3] + d] - nodes[3 * e.connectivity[1] + d]; } }
  const double cr[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
  const double len = std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
  e.area = len / 2.0;
  if (len > 1e-15) for (int d = 0; d < 3; ++d) e.normal[d] = cr[d] / len;
  if (e.normal[0] * c[0] + e.normal[1] * c[1] + e.normal[2] * c[2] < 0.0) for (int d = 0; d < 3; ++d) e.normal[d] = -e.normal[d];
  for (int d = 0; d < 3; ++d) e.center[d] = c[d];
}

// generators.rs:110-228
inline Mesh generate_icosphere_mesh(double radius, size_t subdivisions) {
  const double phi = (1.0 + std::sqrt(5.0)) / 2.0;
  std::vector<std::array<double, 3>> v = {{-1, phi, 0}, {1, phi, 0}, {-1, -phi, 0}, {1, -phi, 0}, {0, -1, phi}, {0, 1, phi},
                                          {0, -1, -phi}, {0, 1, -phi}, {phi, 0, -1}, {phi, 0, 1}, {-phi, 0, -1}, {-phi, 0, 1}};
  for (auto& p : v) { const double l = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); p[0] /= l; p[1] /= l; p[2] /= l; }
  std::vector<std::array<size_t, 3>> f = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                                          {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8}, {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
  for (size_t s = 0; s < subdivisions; ++s) {
    std::map<std::pair<size_t, size_t>, size_t> cache;
    auto mid = [&](size_t a, size_t b) {
      auto key = a < b ? std::make_pair(a, b) : std::make_pair(b, a);
      auto it = cache.find(key);
      if (it != cache.end()) return it->second;
      std::array<double, 3> m = {(v[a][0] + v[b][0]) / 2.0, (v[a][1] + v[b][1]) / 2.0, (v[a][2] + v[b][2]) / 2.0};
      const double l = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
      v.push_back({m[0] / l, m[1] / l, m[2] / l});
      return cache[key] = v.size() - 1;
    };
    std::vector<std::array<size_t, 3>> nf;
    for (auto& t : f) {
      const size_t m01 = mid(t[0], t[1]), m12 = mid(t[1], t[2]), m20 = mid(t[2], t[0]);
      nf.push_back({t[0], m01, m20}); nf.push_back({t[1], m12, m01}); nf.push_back({t[2], m20, m12}); nf.push_back({m01, m12, m20});
    }
    f.swap(nf);
  }
  Mesh m;
  for (auto& p : v) { m.nodes.push_back(p[0] * radius); m.nodes.push_back(p[1] * radius); m.nodes.push_back(p[2] * radius); }
  for (size_t i = 0; i < f.size(); ++i) {
    Element e; e.connectivity = {f[i][0], f[i][1], f[i][2]}; e.dof_addresses = {i};
    compute_element_geometry(e, m.nodes);
    m.elements.push_back(e);
  }
  return m;
}

// tbem.rs:13-20
struct TbemSystem {
  std::vector<Complex64> matrix;      // num_dofs x num_dofs row-major
  std::vector<Complex64> rhs;
  size_t num_dofs = 0;
  Complex64& at(size_t i, size_t j) { return matrix[i * num_dofs + j]; }
};

namespace detail {
struct Flat {
  std::vector<int32_t> conn, dof, bclen;
  std::vector<double> center, normal, area;
  std::vector<uint8_t> bct, eval;
  std::vector<Complex64> bcv;
  ma_mesh_t c{};
};
inline void flatten(const std::vector<Element>& el, const std::vector<double>& nodes, Flat& F) {
  const size_t n = el.size();
  F.conn.assign(4 * n, -1); F.dof.resize(n); F.bclen.assign(n, 1); F.center.resize(3 * n); F.normal.resize(3 * n); F.area.resize(n);
  F.bct.assign(n, 0); F.eval.assign(n, 0); F.bcv.assign(4 * n, Complex64(0.0, 0.0));
  for (size_t e = 0; e < n; ++e) {
    for (size_t a = 0; a < el[e].connectivity.size() && a < 4; ++a) F.conn[4 * e + a] = (int32_t)el[e].connectivity[a];
    for (int d = 0; d < 3; ++d) { F.center[3 * e + d] = el[e].center[d]; F.normal[3 * e + d] = el[e].normal[d]; }
    F.area[e] = el[e].area; F.dof[e] = (int32_t)el[e].dof_addresses.at(0);
    F.eval[e] = el[e].property == ElementProperty::Evaluation;
    const auto& bc = el[e].boundary_condition;
    F.bct[e] = bc.kind == BoundaryCondition::Velocity ? 0 : (bc.kind == BoundaryCondition::Pressure ? 1 : 2);
    F.bclen[e] = (int32_t)(bc.values.empty() ? 1 : bc.values.size());
    for (size_t a = 0; a < bc.values.size() && a < 4; ++a) F.bcv[4 * e + a] = bc.values[a];
  }
  F.c.n_nodes = (int32_t)(nodes.size() / 3); F.c.nodes = nodes.data(); F.c.n_elem = (int32_t)n; F.c.conn = F.conn.data();
  F.c.center = F.center.data(); F.c.normal = F.normal.data(); F.c.area = F.area.data(); F.c.dof = F.dof.data();
  F.c.bc_type = F.bct.data(); F.c.bc_values = reinterpret_cast<const ma_c64*>(F.bcv.data()); F.c.bc_len = F.bclen.data(); F.c.is_eval = F.eval.data();
}
inline ma_physics_t phys(const PhysicsParams& p) { return ma_physics_t{p.wave_number, p.harmonic_factor, p.tau, p.gamma()}; }
}  // namespace detail

// tbem.rs:96-101 — same arguments, same result; throws BemError with the C status on failure
inline TbemSystem build_tbem_system_with_beta(const std::vector<Element>& elements, const std::vector<double>& nodes,
                                              const PhysicsParams& physics, Complex64 beta) {
  detail::Flat F; detail::flatten(elements, nodes, F);
  size_t nd = 0; for (auto& e : elements) nd += e.property != ElementProperty::Evaluation;
  TbemSystem s; s.num_dofs = nd; s.matrix.assign(nd * nd, Complex64(0.0, 0.0)); s.rhs.assign(nd, Complex64(0.0, 0.0));
  const ma_physics_t ph = detail::phys(physics);
  const int rc = ma_bem_assemble_tbem(&F.c, &ph, beta.real(), beta.imag(), reinterpret_cast<ma_c64*>(s.matrix.data()), reinterpret_cast<ma_c64*>(s.rhs.data()));
  if (rc != MA_OK) throw BemError(rc, ma_last_error_string());
  return s;
}
inline TbemSystem build_tbem_system(const std::vector<Element>& el, const std::vector<double>& nodes, const PhysicsParams& p) {
  return build_tbem_system_with_beta(el, nodes, p, p.burton_miller_beta());                    // tbem.rs:45-51
}
inline TbemSystem build_tbem_system_scaled(const std::vector<Element>& el, const std::vector<double>& nodes, const PhysicsParams& p, double scale) {
  return build_tbem_system_with_beta(el, nodes, p, p.burton_miller_beta_scaled(scale));        // tbem.rs:85-93
}

// incident.rs:17-39, 317-342
struct IncidentField {
  int kind = 0; double v[3] = {0, 0, 1}; Complex64 amp{1.0, 0.0};
  static IncidentField plane_wave_z() { return IncidentField{}; }
  static IncidentField plane_wave(const double d[3], double a) {
    IncidentField f; const double l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (l > 1e-10) { f.v[0] = d[0] / l; f.v[1] = d[1] / l; f.v[2] = d[2] / l; } else { f.v[0] = 0; f.v[1] = 0; f.v[2] = -1; }
    f.amp = Complex64(a, 0.0); return f;
  }
  static IncidentField point_source(const double p[3], double s) { IncidentField f; f.kind = 1; f.v[0] = p[0]; f.v[1] = p[1]; f.v[2] = p[2]; f.amp = Complex64(s, 0.0); return f; }
  std::vector<Complex64> compute_rhs_with_beta(const std::vector<double>& centers, const std::vector<double>& normals,
                                               const PhysicsParams& physics, Complex64 beta) const {
    const int n = (int)(centers.size() / 3);
    std::vector<Complex64> rhs(n);
    const ma_physics_t ph = detail::phys(physics);
    const int rc = ma_bem_incident_rhs(n, centers.data(), normals.data(), &ph, beta.real(), beta.imag(), kind, v, amp.real(), amp.imag(),
                                       reinterpret_cast<ma_c64*>(rhs.data()));
    if (rc != MA_OK) throw BemError(rc, ma_last_error_string());
    return rhs;
  }
};

}  // namespace math_bem

namespace math_solvers {

using Complex64 = std::complex<double>;

// lu.rs:16-21
struct LuError {
  enum Kind { SingularMatrix, DimensionMismatch, Backend } kind;
  size_t expected = 0, got = 0;
  std::string text;
};

template <class T>
struct Result {
  bool ok; T value; LuError err;
  bool is_ok() const { return ok; }
  bool is_err() const { return !ok; }
  const T& expect(const char* msg) const { if (!ok) throw std::runtime_error(std::string(msg) + ": " + err.text); return value; }
};

// lu.rs:142-153: a is n x n row-major (ndarray C order), b has n entries
inline Result<std::vector<Complex64>> lu_solve(const std::vector<Complex64>& a, size_t nrows, size_t ncols, const std::vector<Complex64>& b) {
  Result<std::vector<Complex64>> r{false, {}, {}};
  if (nrows != ncols || a.size() != nrows * ncols) { r.err = {LuError::DimensionMismatch, nrows, ncols, "Matrix dimensions mismatch"}; return r; }
  if (b.size() != nrows) { r.err = {LuError::DimensionMismatch, nrows, b.size(), "Matrix dimensions mismatch"}; return r; }
  std::vector<Complex64> x(b.size());
  const int rc = ma_lu_solve((int32_t)nrows, reinterpret_cast<const ma_c64*>(a.data()), reinterpret_cast<const ma_c64*>(b.data()), reinterpret_cast<ma_c64*>(x.data()));
  if (rc == MA_OK) { r.ok = true; r.value.swap(x); return r; }
  if (rc == MA_ERR_SINGULAR) r.err = {LuError::SingularMatrix, 0, 0, "Matrix is singular or nearly singular"};
  else if (rc == MA_ERR_DIM) r.err = {LuError::DimensionMismatch, nrows, b.size(), ma_last_error_string()};
  else r.err = {LuError::Backend, 0, 0, ma_last_error_string()};
  return r;
}
inline Result<std::vector<Complex64>> lu_solve(const std::vector<double>& a, size_t n, size_t m, const std::vector<double>& b) {
  std::vector<Complex64> ac(a.begin(), a.end()), bc(b.begin(), b.end());
  return lu_solve(ac, n, m, bc);
}

// lu.rs:25-78, 83-137: the factors stay on the device behind the object
class LuFactorization {
 public:
  size_t n = 0;
  LuFactorization(LuFactorization&& o) noexcept : n(o.n), h_(o.h_) { o.h_ = nullptr; }
  LuFactorization(const LuFactorization&) = delete;
  ~LuFactorization() { if (h_) ma_lu_factorization_destroy(h_); }
  Result<std::vector<Complex64>> solve(const std::vector<Complex64>& b) const {
    Result<std::vector<Complex64>> r{false, {}, {}};
    if (b.size() != n) { r.err = {LuError::DimensionMismatch, n, b.size(), "Matrix dimensions mismatch"}; return r; }
    std::vector<Complex64> x(n);
    const int rc = ma_lu_factorization_solve(h_, reinterpret_cast<const ma_c64*>(b.data()), reinterpret_cast<ma_c64*>(x.data()));
    if (rc == MA_OK) { r.ok = true; r.value.swap(x); } else r.err = {LuError::Backend, 0, 0, ma_last_error_string()};
    return r;
  }
  friend Result<LuFactorization> lu_factorize(const std::vector<Complex64>& a, size_t nrows, size_t ncols);
 private:
  LuFactorization() = default;
  ma_lu_factorization_t* h_ = nullptr;
};
template <>
struct Result<LuFactorization> {
  bool ok = false; std::unique_ptr<LuFactorization> value; LuError err{};
  bool is_ok() const { return ok; }
  bool is_err() const { return !ok; }
  LuFactorization& expect(const char* msg) { if (!ok) throw std::runtime_error(std::string(msg) + ": " + err.text); return *value; }
};
inline Result<LuFactorization> lu_factorize(const std::vector<Complex64>& a, size_t nrows, size_t ncols) {
  Result<LuFactorization> r;
  if (nrows != ncols || a.size() != nrows * ncols) { r.err = {LuError::DimensionMismatch, nrows, ncols, "Matrix dimensions mismatch"}; return r; }
  ma_lu_factorization_t* h = nullptr;
  const int rc = ma_lu_factorize((int32_t)nrows, reinterpret_cast<const ma_c64*>(a.data()), &h);
  if (rc == MA_ERR_SINGULAR) { r.err = {LuError::SingularMatrix, 0, 0, "Matrix is singular or nearly singular"}; return r; }
  if (rc != MA_OK) { r.err = {LuError::Backend, 0, 0, ma_last_error_string()}; return r; }
  r.value.reset(new LuFactorization()); r.value->n = nrows; r.value->h_ = h; r.ok = true;
  return r;
}

// ---------------------------------------------------------------- sparse/csr.rs, traits.rs, iterative/gmres.rs
struct SolverError : std::runtime_error {
  int status;
  SolverError(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};
inline void solver_check(int rc) { if (rc != MA_OK) throw SolverError(rc, ma_last_error_string()); }

// traits.rs:316-364: the operator boundary. `handle()` is the device operator GMRES iterates on.
struct LinearOperator {
  virtual ~LinearOperator() = default;
  virtual size_t num_rows() const = 0;
  virtual size_t num_cols() const = 0;
  virtual ma_op_t* handle() const = 0;
  std::vector<Complex64> apply(const std::vector<Complex64>& x) const {
    std::vector<Complex64> y(num_rows());
    solver_check(ma_op_apply(handle(), reinterpret_cast<const ma_c64*>(x.data()), reinterpret_cast<ma_c64*>(y.data())));
    return y;
  }
  std::vector<Complex64> apply_transpose(const std::vector<Complex64>& x) const {
    std::vector<Complex64> y(num_cols());
    solver_check(ma_op_apply_transpose(handle(), reinterpret_cast<const ma_c64*>(x.data()), reinterpret_cast<ma_c64*>(y.data())));
    return y;
  }
  std::vector<Complex64> apply_hermitian(const std::vector<Complex64>& x) const {
    std::vector<Complex64> y(num_cols());
    solver_check(ma_op_apply_hermitian(handle(), reinterpret_cast<const ma_c64*>(x.data()), reinterpret_cast<ma_c64*>(y.data())));
    return y;
  }
};

// csr.rs:21-33. Square operators only on the device path (what the FEM and BEM callers build).
class CsrMatrix : public LinearOperator {
 public:
  size_t num_rows_ = 0, num_cols_ = 0;
  std::vector<int64_t> row_ptrs, col_indices;
  std::vector<Complex64> values;

  // csr.rs:69-99
  static CsrMatrix from_raw_parts(size_t nr, size_t nc, std::vector<int64_t> rp, std::vector<int64_t> ci, std::vector<Complex64> v) {
    CsrMatrix m; m.num_rows_ = nr; m.num_cols_ = nc; m.row_ptrs = std::move(rp); m.col_indices = std::move(ci); m.values = std::move(v);
    return m;
  }
  // csr.rs:135-205: sorted by (row, col), duplicates summed, zeros kept
  static CsrMatrix from_triplets(size_t nr, size_t nc, std::vector<std::tuple<size_t, size_t, Complex64>> t) {
    std::stable_sort(t.begin(), t.end(), [](const auto& a, const auto& b) { return std::get<0>(a) != std::get<0>(b) ? std::get<0>(a) < std::get<0>(b) : std::get<1>(a) < std::get<1>(b); });
    CsrMatrix m; m.num_rows_ = nr; m.num_cols_ = nc; m.row_ptrs.assign(nr + 1, 0);
    for (size_t i = 0; i < t.size(); ++i) {
      const size_t r = std::get<0>(t[i]), c = std::get<1>(t[i]);
      if (i > 0 && std::get<0>(t[i - 1]) == r && std::get<1>(t[i - 1]) == c) { m.values.back() += std::get<2>(t[i]); continue; }
      m.col_indices.push_back((int64_t)c); m.values.push_back(std::get<2>(t[i])); m.row_ptrs[r + 1]++;
    }
    for (size_t r = 0; r < nr; ++r) m.row_ptrs[r + 1] += m.row_ptrs[r];
    return m;
  }
  // csr.rs:101-133: entries with |a| > threshold
  static CsrMatrix from_dense(const std::vector<Complex64>& a, size_t nr, size_t nc, double threshold) {
    CsrMatrix m; m.num_rows_ = nr; m.num_cols_ = nc; m.row_ptrs.assign(nr + 1, 0);
    for (size_t r = 0; r < nr; ++r) {
      for (size_t c = 0; c < nc; ++c) if (std::abs(a[r * nc + c]) > threshold) { m.col_indices.push_back((int64_t)c); m.values.push_back(a[r * nc + c]); }
      m.row_ptrs[r + 1] = (int64_t)m.values.size();
    }
    return m;
  }
  static CsrMatrix identity(size_t n) {
    CsrMatrix m; m.num_rows_ = m.num_cols_ = n; m.row_ptrs.resize(n + 1);
    for (size_t i = 0; i <= n; ++i) m.row_ptrs[i] = (int64_t)i;
    for (size_t i = 0; i < n; ++i) { m.col_indices.push_back((int64_t)i); m.values.push_back(Complex64(1.0, 0.0)); }
    return m;
  }
  size_t nnz() const { return values.size(); }
  Complex64 get(size_t i, size_t j) const {                       // csr.rs:207-220
    for (int64_t t = row_ptrs[i]; t < row_ptrs[i + 1]; ++t) if ((size_t)col_indices[(size_t)t] == j) return values[(size_t)t];
    return Complex64(0.0, 0.0);
  }
  std::vector<Complex64> matvec(const std::vector<Complex64>& x) const { return apply(x); }   // csr.rs:240-292, on the device
  size_t num_rows() const override { return num_rows_; }
  size_t num_cols() const override { return num_cols_; }
  ma_csr_t* csr_handle() const { ensure(); return csr_; }
  ma_op_t* handle() const override { ensure(); return op_; }
  CsrMatrix() = default;
  CsrMatrix(CsrMatrix&& o) noexcept { *this = std::move(o); }
  CsrMatrix& operator=(CsrMatrix&& o) noexcept {
    release(); num_rows_ = o.num_rows_; num_cols_ = o.num_cols_; row_ptrs = std::move(o.row_ptrs); col_indices = std::move(o.col_indices);
    values = std::move(o.values); csr_ = o.csr_; op_ = o.op_; o.csr_ = nullptr; o.op_ = nullptr; return *this;
  }
  CsrMatrix(const CsrMatrix&) = delete;
  CsrMatrix& operator=(const CsrMatrix&) = delete;
  ~CsrMatrix() override { release(); }

 private:
  mutable ma_csr_t* csr_ = nullptr;
  mutable ma_op_t* op_ = nullptr;
  void ensure() const {
    if (op_) return;
    if (num_rows_ != num_cols_) throw SolverError(MA_ERR_UNSUPPORTED, "the device path takes square operators");
    solver_check(ma_csr_create((int64_t)num_rows_, row_ptrs.data(), col_indices.data(), reinterpret_cast<const ma_c64*>(values.data()), 0, &csr_));
    solver_check(ma_op_create_csr(csr_, &op_));
  }
  void release() { if (op_) ma_op_destroy(op_); if (csr_) ma_csr_destroy(csr_); op_ = nullptr; csr_ = nullptr; }
};

// math-bem/src/core/solver/fmm_interface.rs:25-53
class DenseOperator : public LinearOperator {
 public:
  DenseOperator(const std::vector<Complex64>& a, size_t n) : n_(n) {
    if (a.size() != n * n) throw SolverError(MA_ERR_DIM, "DenseOperator: matrix is not n x n");
    solver_check(ma_op_create_dense((int64_t)n, reinterpret_cast<const ma_c64*>(a.data()), 0, &op_));
  }
  ~DenseOperator() override { if (op_) ma_op_destroy(op_); }
  DenseOperator(const DenseOperator&) = delete;
  DenseOperator& operator=(const DenseOperator&) = delete;
  size_t num_rows() const override { return n_; }
  size_t num_cols() const override { return n_; }
  ma_op_t* handle() const override { return op_; }
 private:
  size_t n_; ma_op_t* op_ = nullptr;
};

// traits.rs:370-375 and preconditioners/diagonal.rs:20-75
struct Preconditioner {
  virtual ~Preconditioner() = default;
  virtual ma_precond_t* handle() const = 0;
  std::vector<Complex64> apply(const std::vector<Complex64>& r) const {
    std::vector<Complex64> z(r.size());
    solver_check(ma_precond_apply(handle(), reinterpret_cast<const ma_c64*>(r.data()), reinterpret_cast<ma_c64*>(z.data())));
    return z;
  }
};
class DiagonalPreconditioner : public Preconditioner {
 public:
  static DiagonalPreconditioner from_csr(const CsrMatrix& m) { DiagonalPreconditioner p; solver_check(ma_precond_create_jacobi(m.csr_handle(), 1.0, 1, &p.h_)); return p; }
  DiagonalPreconditioner(DiagonalPreconditioner&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  ~DiagonalPreconditioner() override { if (h_) ma_precond_destroy(h_); }
  ma_precond_t* handle() const override { return h_; }
 private:
  DiagonalPreconditioner() = default;
  ma_precond_t* h_ = nullptr;
};

// gmres.rs:14-85
struct GmresConfig {
  size_t max_iterations = 100, restart = 30;
  double tolerance = 1e-6;
  size_t print_interval = 0;
  static GmresConfig for_small_problems() { return GmresConfig{50, 50, 1e-8, 0}; }
  static GmresConfig with_restart(size_t r) { GmresConfig c; c.restart = r; return c; }
};
struct GmresSolution {
  std::vector<Complex64> x;
  size_t iterations = 0, restarts = 0;
  double residual = 0.0;
  bool converged = false;
};
namespace detail {
inline GmresSolution run_gmres(const LinearOperator& op, const Preconditioner* m, const std::vector<Complex64>& b, const std::vector<Complex64>* x0, const GmresConfig& c) {
  GmresSolution s; s.x.resize(b.size());
  ma_gmres_info_t info{};
  const ma_c64* g = x0 ? reinterpret_cast<const ma_c64*>(x0->data()) : nullptr;
  const int rc = m ? ma_gmres_preconditioned(op.handle(), m->handle(), reinterpret_cast<const ma_c64*>(b.data()), g, (int32_t)c.restart, (int32_t)c.max_iterations, c.tolerance,
                                             reinterpret_cast<ma_c64*>(s.x.data()), &info)
                   : ma_gmres(op.handle(), reinterpret_cast<const ma_c64*>(b.data()), g, (int32_t)c.restart, (int32_t)c.max_iterations, c.tolerance,
                              reinterpret_cast<ma_c64*>(s.x.data()), &info);
  solver_check(rc);
  s.iterations = (size_t)info.iterations; s.restarts = (size_t)info.restarts; s.residual = info.residual; s.converged = info.converged != 0;
  return s;
}
}  // namespace detail
// gmres.rs:96-103, 105-277, 282-292, 434-585
inline GmresSolution gmres(const LinearOperator& a, const std::vector<Complex64>& b, const GmresConfig& c) { return detail::run_gmres(a, nullptr, b, nullptr, c); }
inline GmresSolution gmres_with_guess(const LinearOperator& a, const std::vector<Complex64>& b, const std::vector<Complex64>* x0, const GmresConfig& c) { return detail::run_gmres(a, nullptr, b, x0, c); }
inline GmresSolution gmres_preconditioned(const LinearOperator& a, const Preconditioner& m, const std::vector<Complex64>& b, const GmresConfig& c) { return detail::run_gmres(a, &m, b, nullptr, c); }
inline GmresSolution gmres_preconditioned_with_guess(const LinearOperator& a, const Preconditioner& m, const std::vector<Complex64>& b, const std::vector<Complex64>* x0, const GmresConfig& c) {
  return detail::run_gmres(a, &m, b, x0, c);
}

// preconditioners/amg.rs:41-218 (enums, AmgConfig and its presets), 276-372 (from_csr: hierarchy built on the host inside the library),
// 981-1103 (v_cycle, apply: on the device). A hierarchy of the caller's can be given instead: levels[l] = {A_l, P_l (n_l x n_{l+1}),
// R_l (n_{l+1} x n_l)}, P and R empty on the coarsest level.
enum class AmgCoarsening { RugeStuben = 0, Pmis = 1, Hmis = 2 };
enum class AmgInterpolation { Standard = 0, Extended = 1, Direct = 2 };
enum class AmgSmoother { Jacobi = 0, L1Jacobi = 1, SymmetricGaussSeidel = 2, Chebyshev = 3 };
enum class AmgCycle { V = 0, W = 1, F = 2 };
struct AmgConfig {
  AmgCoarsening coarsening = AmgCoarsening::RugeStuben;
  AmgInterpolation interpolation = AmgInterpolation::Standard;
  AmgSmoother smoother = AmgSmoother::Jacobi;
  AmgCycle cycle = AmgCycle::V;
  double strong_threshold = 0.25;
  size_t max_levels = 25, coarse_size = 50, num_pre_smooth = 1, num_post_smooth = 1;
  double jacobi_weight = 0.6667, trunc_factor = 0.0;
  size_t max_interp_elements = 4, aggressive_coarsening_levels = 0;
  static AmgConfig for_bem() { AmgConfig c; c.strong_threshold = 0.5; c.coarsening = AmgCoarsening::Pmis; c.smoother = AmgSmoother::L1Jacobi; c.max_interp_elements = 6; return c; }
  static AmgConfig for_fem() { AmgConfig c; c.smoother = AmgSmoother::SymmetricGaussSeidel; return c; }
  static AmgConfig for_parallel() { AmgConfig c; c.coarsening = AmgCoarsening::Pmis; c.jacobi_weight = 0.8; c.num_pre_smooth = 2; c.num_post_smooth = 2; return c; }
  static AmgConfig for_difficult_problems() {
    AmgConfig c; c.interpolation = AmgInterpolation::Extended; c.smoother = AmgSmoother::SymmetricGaussSeidel; c.max_interp_elements = 8; c.num_pre_smooth = 2; c.num_post_smooth = 2;
    return c;
  }
  ma_amg_config_t c_config() const {
    return ma_amg_config_t{(int32_t)coarsening, (int32_t)interpolation, (int32_t)smoother, (int32_t)cycle, strong_threshold, (int32_t)max_levels, (int32_t)coarse_size,
                           (int32_t)num_pre_smooth, (int32_t)num_post_smooth, jacobi_weight, trunc_factor, (int32_t)max_interp_elements, (int32_t)aggressive_coarsening_levels};
  }
};
struct AmgDiagnostics { size_t num_levels = 0; double grid_complexity = 1.0, operator_complexity = 1.0, setup_time_ms = 0.0; std::vector<size_t> level_dofs, level_nnz; };
struct AmgLevel { const CsrMatrix* matrix = nullptr; const CsrMatrix* prolongation = nullptr; const CsrMatrix* restriction = nullptr; };
class AmgPreconditioner : public Preconditioner {
 public:
  AmgPreconditioner(const std::vector<AmgLevel>& levels, const AmgConfig& c) {
    if (levels.empty()) throw SolverError(MA_ERR_INVALID, "AmgPreconditioner: no levels");
    std::vector<ma_csr_t*> a, p, r;
    for (size_t l = 0; l < levels.size(); ++l) {
      if (!levels[l].matrix) throw SolverError(MA_ERR_INVALID, "AmgPreconditioner: level without a matrix");
      a.push_back(levels[l].matrix->csr_handle());
      if (l + 1 < levels.size()) {
        if (!levels[l].prolongation || !levels[l].restriction) throw SolverError(MA_ERR_INVALID, "AmgPreconditioner: level without P / R");
        p.push_back(rect(*levels[l].prolongation)); r.push_back(rect(*levels[l].restriction));
      }
    }
    solver_check(ma_precond_create_amg((int32_t)levels.size(), a.data(), p.data(), r.data(), c.smoother == AmgSmoother::Chebyshev ? 0 : (int32_t)c.smoother, c.jacobi_weight,
                                       (int32_t)c.num_pre_smooth, (int32_t)c.num_post_smooth, (int32_t)c.cycle, &h_));
  }
  // AmgPreconditioner::from_csr(&matrix, config), amg.rs:276-372
  static AmgPreconditioner from_csr(const CsrMatrix& m, const AmgConfig& c) {
    AmgPreconditioner p;
    const ma_amg_config_t cc = c.c_config();
    solver_check(ma_precond_create_amg_from_csr(m.csr_handle(), &cc, &p.h_));
    return p;
  }
  AmgPreconditioner(AmgPreconditioner&& o) noexcept : h_(o.h_), owned_(std::move(o.owned_)) { o.h_ = nullptr; o.owned_.clear(); }
  size_t num_levels() const { int32_t n = 0; solver_check(ma_precond_amg_info(h_, &n, nullptr, nullptr, nullptr)); return (size_t)n; }
  double grid_complexity() const { double v = 1.0; solver_check(ma_precond_amg_info(h_, nullptr, &v, nullptr, nullptr)); return v; }
  double operator_complexity() const { double v = 1.0; solver_check(ma_precond_amg_info(h_, nullptr, nullptr, &v, nullptr)); return v; }
  double setup_time_ms() const { double v = 0.0; solver_check(ma_precond_amg_info(h_, nullptr, nullptr, nullptr, &v)); return v; }
  AmgDiagnostics diagnostics() const {                      // amg.rs:1124-1133
    AmgDiagnostics d; int32_t n = 0;
    solver_check(ma_precond_amg_info(h_, &n, &d.grid_complexity, &d.operator_complexity, &d.setup_time_ms));
    d.num_levels = (size_t)n;
    for (int32_t l = 0; l < n; ++l) {
      ma_csr_t* a = nullptr; int64_t rows = 0, nnz = 0;
      solver_check(ma_precond_amg_level(h_, l, &a, nullptr, nullptr)); solver_check(ma_csr_num_rows(a, &rows, &nnz));
      d.level_dofs.push_back((size_t)rows); d.level_nnz.push_back((size_t)nnz);
    }
    return d;
  }
  ~AmgPreconditioner() override { if (h_) ma_precond_destroy(h_); for (ma_csr_t* q : owned_) ma_csr_destroy(q); }
  AmgPreconditioner(const AmgPreconditioner&) = delete;
  AmgPreconditioner& operator=(const AmgPreconditioner&) = delete;
  ma_precond_t* handle() const override { return h_; }
 private:
  AmgPreconditioner() = default;
  ma_csr_t* rect(const CsrMatrix& m) {
    ma_csr_t* q = nullptr;
    solver_check(ma_csr_create_rect((int64_t)m.num_rows_, (int64_t)m.num_cols_, m.row_ptrs.data(), m.col_indices.data(), reinterpret_cast<const ma_c64*>(m.values.data()), 0, &q));
    owned_.push_back(q);
    return q;
  }
  ma_precond_t* h_ = nullptr;
  std::vector<ma_csr_t*> owned_;
};

// preconditioners/ilu.rs: IluPreconditioner::from_csr(&matrix) (ILU(0), factorised on the host by the library, applied on the device)
class IluPreconditioner : public Preconditioner {
 public:
  static IluPreconditioner from_csr(const CsrMatrix& m) { IluPreconditioner p; solver_check(ma_precond_create_ilu0(m.csr_handle(), &p.h_)); return p; }
  IluPreconditioner(IluPreconditioner&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  ~IluPreconditioner() override { if (h_) ma_precond_destroy(h_); }
  ma_precond_t* handle() const override { return h_; }
 private:
  IluPreconditioner() = default;
  ma_precond_t* h_ = nullptr;
};
// preconditioners/ilu_parallel.rs: IluColoringPreconditioner::from_csr (:52: ILU(0) with level-scheduled solves -- the device apply of
// IluPreconditioner), IluFixedPointPreconditioner::from_csr(&matrix, iterations) / from_csr_default (:397, :492);
// preconditioners/schwarz.rs: AdditiveSchwarzPreconditioner::from_csr(&matrix, num_subdomains, overlap) (:84), stats() (:148)
class IluColoringPreconditioner : public Preconditioner {
 public:
  static IluColoringPreconditioner from_csr(const CsrMatrix& m) { IluColoringPreconditioner p; solver_check(ma_precond_create_ilu0(m.csr_handle(), &p.h_)); return p; }
  IluColoringPreconditioner(IluColoringPreconditioner&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  ~IluColoringPreconditioner() override { if (h_) ma_precond_destroy(h_); }
  ma_precond_t* handle() const override { return h_; }
 private:
  IluColoringPreconditioner() = default;
  ma_precond_t* h_ = nullptr;
};
class IluFixedPointPreconditioner : public Preconditioner {
 public:
  static IluFixedPointPreconditioner from_csr(const CsrMatrix& m, size_t iterations) {
    IluFixedPointPreconditioner p; solver_check(ma_precond_create_ilu_fixed_point(m.csr_handle(), (int32_t)iterations, &p.h_)); return p;
  }
  static IluFixedPointPreconditioner from_csr_default(const CsrMatrix& m) { return from_csr(m, 3); }
  IluFixedPointPreconditioner(IluFixedPointPreconditioner&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  ~IluFixedPointPreconditioner() override { if (h_) ma_precond_destroy(h_); }
  ma_precond_t* handle() const override { return h_; }
 private:
  IluFixedPointPreconditioner() = default;
  ma_precond_t* h_ = nullptr;
};
struct SchwarzStats { size_t num_subdomains = 0, min_size = 0, max_size = 0; double avg_size = 0.0; };
class AdditiveSchwarzPreconditioner : public Preconditioner {
 public:
  static AdditiveSchwarzPreconditioner from_csr(const CsrMatrix& m, size_t num_subdomains, size_t overlap) {
    AdditiveSchwarzPreconditioner p; solver_check(ma_precond_create_schwarz(m.csr_handle(), (int32_t)num_subdomains, (int32_t)overlap, &p.h_)); return p;
  }
  AdditiveSchwarzPreconditioner(AdditiveSchwarzPreconditioner&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  ~AdditiveSchwarzPreconditioner() override { if (h_) ma_precond_destroy(h_); }
  ma_precond_t* handle() const override { return h_; }
  SchwarzStats stats() const {
    int64_t a = 0, b = 0, c = 0; double d = 0.0;
    solver_check(ma_precond_schwarz_stats(h_, &a, &b, &c, &d));
    return SchwarzStats{(size_t)a, (size_t)b, (size_t)c, d};
  }
 private:
  AdditiveSchwarzPreconditioner() = default;
  ma_precond_t* h_ = nullptr;
};

// iterative/gmres_pipelined.rs:18-250: gmres_pipelined(operator, precond, b, x0, config); precond may be null (IdentityPreconditioner)
inline GmresSolution gmres_pipelined(const LinearOperator& a, const Preconditioner* m, const std::vector<Complex64>& b, const std::vector<Complex64>* x0, const GmresConfig& c) {
  GmresSolution s; s.x.resize(b.size());
  ma_gmres_info_t info{};
  solver_check(ma_gmres_pipelined(a.handle(), m ? m->handle() : nullptr, reinterpret_cast<const ma_c64*>(b.data()), x0 ? reinterpret_cast<const ma_c64*>(x0->data()) : nullptr,
                                  (int32_t)c.restart, (int32_t)c.max_iterations, c.tolerance, reinterpret_cast<ma_c64*>(s.x.data()), &info));
  s.iterations = (size_t)info.iterations; s.restarts = (size_t)info.restarts; s.residual = info.residual; s.converged = info.converged != 0;
  return s;
}

// iterative/bicgstab.rs:12-182, cgs.rs:12-139, cg.rs:12-138: BiCgstabConfig / CgsConfig / CgConfig {max_iterations 1000, tolerance 1e-6} and the
// solutions {x, iterations, residual, converged}
struct KrylovConfig { size_t max_iterations = 1000; double tolerance = 1e-6; size_t print_interval = 0; };
using BiCgstabConfig = KrylovConfig; using CgsConfig = KrylovConfig; using CgConfig = KrylovConfig;
struct KrylovSolution { std::vector<Complex64> x; size_t iterations = 0; double residual = 0.0; bool converged = false; };
using BiCgstabSolution = KrylovSolution; using CgsSolution = KrylovSolution; using CgSolution = KrylovSolution;
namespace detail {
typedef int (*krylov_fn)(ma_op_t*, const ma_c64*, int32_t, double, ma_c64*, ma_gmres_info_t*);
inline KrylovSolution run_krylov(krylov_fn fn, const LinearOperator& a, const std::vector<Complex64>& b, const KrylovConfig& c) {
  KrylovSolution s; s.x.resize(b.size());
  ma_gmres_info_t info{};
  solver_check(fn(a.handle(), reinterpret_cast<const ma_c64*>(b.data()), (int32_t)c.max_iterations, c.tolerance, reinterpret_cast<ma_c64*>(s.x.data()), &info));
  s.iterations = (size_t)info.iterations; s.residual = info.residual; s.converged = info.converged != 0;
  return s;
}
}  // namespace detail
inline BiCgstabSolution bicgstab(const LinearOperator& a, const std::vector<Complex64>& b, const BiCgstabConfig& c) { return detail::run_krylov(ma_bicgstab, a, b, c); }
inline CgsSolution cgs(const LinearOperator& a, const std::vector<Complex64>& b, const CgsConfig& c) { return detail::run_krylov(ma_cgs, a, b, c); }
inline CgSolution cg(const LinearOperator& a, const std::vector<Complex64>& b, const CgConfig& c) { return detail::run_krylov(ma_cg, a, b, c); }

// AmgPreconditioner's smoothers (preconditioners/amg.rs:855-884, 887-929, 932-978) on the device operator: x is updated in place
inline void smooth_jacobi(const CsrMatrix& a, std::vector<Complex64>& x, const std::vector<Complex64>& b, double omega, size_t num_sweeps) {
  solver_check(ma_csr_jacobi(a.csr_handle(), reinterpret_cast<ma_c64*>(x.data()), reinterpret_cast<const ma_c64*>(b.data()), omega, (int)num_sweeps));
}
inline void smooth_l1_jacobi(const CsrMatrix& a, std::vector<Complex64>& x, const std::vector<Complex64>& b, size_t num_sweeps) {
  solver_check(ma_csr_l1jacobi(a.csr_handle(), reinterpret_cast<ma_c64*>(x.data()), reinterpret_cast<const ma_c64*>(b.data()), (int)num_sweeps));
}
inline void smooth_sym_gauss_seidel(const CsrMatrix& a, std::vector<Complex64>& x, const std::vector<Complex64>& b, size_t num_sweeps) {
  solver_check(ma_csr_sym_gauss_seidel(a.csr_handle(), reinterpret_cast<ma_c64*>(x.data()), reinterpret_cast<const ma_c64*>(b.data()), (int)num_sweeps));
}

}  // namespace math_solvers

// ---- math-fem/src/multigrid/smoother.rs:12-68, 163-176 and assembly/helmholtz.rs:22-33: the COO system matrix and its smoothers
namespace math_fem {
using math_solvers::Complex64;
enum class SmootherType { GaussSeidel = 0, Jacobi = 1, SymmetricGaussSeidel = 2 };
struct SmootherConfig {
  SmootherType smoother_type = SmootherType::GaussSeidel;
  size_t iterations = 2;
  double omega = 2.0 / 3.0;
};
class HelmholtzMatrix {
 public:
  std::vector<int64_t> rows, cols;
  std::vector<Complex64> values;
  size_t dim = 0;
  HelmholtzMatrix(size_t n, std::vector<int64_t> r, std::vector<int64_t> c, std::vector<Complex64> v) : rows(std::move(r)), cols(std::move(c)), values(std::move(v)), dim(n) {}
  HelmholtzMatrix(const HelmholtzMatrix&) = delete;
  HelmholtzMatrix& operator=(const HelmholtzMatrix&) = delete;
  ~HelmholtzMatrix() { if (h_) ma_csr_destroy(h_); }
  ma_csr_t* handle() const {
    if (!h_) math_solvers::solver_check(ma_fem_matrix_create((int64_t)dim, (int64_t)values.size(), rows.data(), cols.data(), reinterpret_cast<const ma_c64*>(values.data()), 0, &h_));
    return h_;
  }
 private:
  mutable ma_csr_t* h_ = nullptr;
};
// smooth(matrix, x, b, config): x updated in place
inline void smooth(const HelmholtzMatrix& m, std::vector<Complex64>& x, const std::vector<Complex64>& b, const SmootherConfig& c) {
  math_solvers::solver_check(ma_fem_smooth(m.handle(), reinterpret_cast<ma_c64*>(x.data()), reinterpret_cast<const ma_c64*>(b.data()), (int)c.smoother_type, (int)c.iterations, c.omega));
}
inline std::vector<Complex64> compute_residual(const HelmholtzMatrix& m, const std::vector<Complex64>& x, const std::vector<Complex64>& b) {
  std::vector<Complex64> r(b.size());
  math_solvers::solver_check(ma_fem_residual(m.handle(), reinterpret_cast<const ma_c64*>(x.data()), reinterpret_cast<const ma_c64*>(b.data()), reinterpret_cast<ma_c64*>(r.data())));
  return r;
}
}  // namespace math_fem

// ---------------------------------------------------------------- operators over a mesh: matrix-free TBEM, SLFMM, the sweep
namespace math_bem {

// types.rs:445-488, the fields build_slfmm_system reads
struct Cluster {
  std::array<double, 3> center{};
  std::vector<size_t> element_indices, near_clusters, far_clusters;
};

// owns the device copy of a mesh (ma_bem_plan_t): what the operators below borrow
class BemPlan {
 public:
  BemPlan(const std::vector<Element>& elements, const std::vector<double>& nodes, int device = 0) {
    detail::flatten(elements, nodes, F_);
    const int rc = ma_bem_plan_create(&F_.c, device, &h_);
    if (rc != MA_OK) throw BemError(rc, ma_last_error_string());
  }
  ~BemPlan() { if (h_) ma_bem_plan_destroy(h_); }
  BemPlan(const BemPlan&) = delete;
  BemPlan& operator=(const BemPlan&) = delete;
  ma_bem_plan_t* handle() const { return h_; }
  const ma_mesh_t& mesh() const { return F_.c; }
  size_t num_dofs() const { int32_t n = 0; ma_bem_plan_num_dofs(h_, &n); return (size_t)n; }
 private:
  detail::Flat F_;
  ma_bem_plan_t* h_ = nullptr;
};

// the matrix-free TBEM operator (BASELINE configs[4]): y = A x of the dense Burton-Miller matrix without storing it; with
// several devices the collocation rows are sharded inside the library (ma_op_create_tbem_multi)
class TbemOperator : public math_solvers::LinearOperator {
 public:
  TbemOperator(const BemPlan& plan, const PhysicsParams& p, Complex64 beta) : n_(plan.num_dofs()) {
    const ma_physics_t ph = detail::phys(p);
    math_solvers::solver_check(ma_op_create_tbem(plan.handle(), &ph, beta.real(), beta.imag(), 0, (int32_t)n_, &op_));
  }
  TbemOperator(const std::vector<Element>& elements, const std::vector<double>& nodes, const PhysicsParams& p, Complex64 beta, const std::vector<int32_t>& devices) {
    detail::Flat F; detail::flatten(elements, nodes, F);
    const ma_physics_t ph = detail::phys(p);
    math_solvers::solver_check(ma_op_create_tbem_multi(&F.c, &ph, beta.real(), beta.imag(), devices.data(), (int32_t)devices.size(), &op_));
    int64_t n = 0; ma_op_num_rows(op_, &n); n_ = (size_t)n;
  }
  ~TbemOperator() override { if (op_) ma_op_destroy(op_); }
  TbemOperator(const TbemOperator&) = delete;
  TbemOperator& operator=(const TbemOperator&) = delete;
  size_t num_rows() const override { return n_; }
  size_t num_cols() const override { return n_; }
  ma_op_t* handle() const override { return op_; }
  size_t num_shards() const { int32_t s = 0; ma_op_num_shards(op_, &s, nullptr, nullptr); return (size_t)s; }
 private:
  size_t n_ = 0; ma_op_t* op_ = nullptr;
};

// build_slfmm_system(elements, nodes, clusters, physics, n_theta, n_phi, n_terms) -> SlfmmSystem (slfmm.rs:417-470) with its
// LinearOperator impl (matvec :150-257, matvec_transpose :262-376) and extract_near_field_matrix (:104-132)
class SlfmmSystem : public math_solvers::LinearOperator {
 public:
  SlfmmSystem(const BemPlan& plan, const std::vector<Cluster>& clusters, const PhysicsParams& p, size_t n_theta, size_t n_phi, size_t n_terms) : n_(plan.num_dofs()) {
    std::vector<double> center; std::vector<int32_t> ep{0}, ei, np_{0}, ni, fp{0}, fi;
    for (const Cluster& c : clusters) {
      center.insert(center.end(), c.center.begin(), c.center.end());
      for (size_t e : c.element_indices) ei.push_back((int32_t)e);
      for (size_t e : c.near_clusters) ni.push_back((int32_t)e);
      for (size_t e : c.far_clusters) fi.push_back((int32_t)e);
      ep.push_back((int32_t)ei.size()); np_.push_back((int32_t)ni.size()); fp.push_back((int32_t)fi.size());
    }
    const ma_clusters_t cl{(int32_t)clusters.size(), center.data(), ep.data(), ei.data(), np_.data(), ni.data(), fp.data(), fi.data()};
    const ma_physics_t ph = detail::phys(p);
    math_solvers::solver_check(ma_op_create_slfmm(plan.handle(), &cl, &ph, (int32_t)n_theta, (int32_t)n_phi, (int32_t)n_terms, &op_));
  }
  ~SlfmmSystem() override { if (op_) ma_op_destroy(op_); }
  SlfmmSystem(const SlfmmSystem&) = delete;
  SlfmmSystem& operator=(const SlfmmSystem&) = delete;
  std::vector<Complex64> matvec(const std::vector<Complex64>& x) const { return apply(x); }
  std::vector<Complex64> matvec_transpose(const std::vector<Complex64>& x) const { return apply_transpose(x); }
  std::vector<Complex64> extract_near_field_matrix() const {
    std::vector<Complex64> a(n_ * n_);
    math_solvers::solver_check(ma_op_slfmm_near_matrix(op_, reinterpret_cast<ma_c64*>(a.data())));
    return a;
  }
  size_t num_rows() const override { return n_; }
  size_t num_cols() const override { return n_; }
  ma_op_t* handle() const override { return op_; }
 private:
  size_t n_; ma_op_t* op_ = nullptr;
};

// build_cluster_tree(elements, target_elements_per_leaf, physics) -> Vec<ClusterLevel> (mlfmm.rs:979-1038) and
// build_mlfmm_system(elements, nodes, cluster_tree, physics) -> MlfmmSystem with MlfmmOperator's LinearOperator impl
// (mlfmm.rs:483-558, 128-460; fmm_interface.rs:98-135: apply_transpose is unimplemented!() there and throws here)
struct ClusterLevel {
  std::vector<Cluster> clusters;
  std::vector<double> radius;
  std::vector<std::vector<size_t>> sons; std::vector<long> father;
  size_t expansion_terms = 4, theta_points = 4, phi_points = 8;
};
class ClusterTree {
 public:
  ClusterTree(const std::vector<Element>& elements, const std::vector<double>& nodes, size_t target_elements_per_leaf, const PhysicsParams& physics) {
    detail::Flat F; detail::flatten(elements, nodes, F);
    const int rc = ma_cluster_tree_build(&F.c, (int32_t)target_elements_per_leaf, physics.wave_number, &h_);
    if (rc != MA_OK) throw BemError(rc, ma_last_error_string());
  }
  ~ClusterTree() { if (h_) ma_cluster_tree_destroy(h_); }
  ClusterTree(const ClusterTree&) = delete;
  ClusterTree& operator=(const ClusterTree&) = delete;
  ma_cluster_tree_t* handle() const { return h_; }
  size_t num_levels() const { int32_t n = 0; ma_cluster_tree_num_levels(h_, &n); return (size_t)n; }
  ClusterLevel level(size_t l) const {
    int32_t nc = 0, terms = 0, theta = 0, phi = 0; int64_t ne = 0, nn = 0, nf = 0, ns = 0;
    math_solvers::solver_check(ma_cluster_tree_level_info(h_, (int32_t)l, &nc, &terms, &theta, &phi, &ne, &nn, &nf, &ns));
    std::vector<double> c((size_t)nc * 3), r((size_t)nc);
    std::vector<int32_t> ep((size_t)nc + 1), ei((size_t)ne + 1), np_((size_t)nc + 1), ni((size_t)nn + 1), fp((size_t)nc + 1), fi((size_t)nf + 1), sp((size_t)nc + 1), si((size_t)ns + 1), fa((size_t)nc);
    math_solvers::solver_check(ma_cluster_tree_level_get(h_, (int32_t)l, c.data(), r.data(), ep.data(), ei.data(), np_.data(), ni.data(), fp.data(), fi.data(), sp.data(), si.data(), fa.data()));
    ClusterLevel L; L.expansion_terms = (size_t)terms; L.theta_points = (size_t)theta; L.phi_points = (size_t)phi;
    L.clusters.resize((size_t)nc); L.radius = r; L.sons.resize((size_t)nc); L.father.resize((size_t)nc);
    for (size_t q = 0; q < (size_t)nc; ++q) {
      for (int d = 0; d < 3; ++d) L.clusters[q].center[d] = c[3 * q + (size_t)d];
      L.clusters[q].element_indices.assign(ei.begin() + ep[q], ei.begin() + ep[q + 1]);
      L.clusters[q].near_clusters.assign(ni.begin() + np_[q], ni.begin() + np_[q + 1]);
      L.clusters[q].far_clusters.assign(fi.begin() + fp[q], fi.begin() + fp[q + 1]);
      L.sons[q].assign(si.begin() + sp[q], si.begin() + sp[q + 1]);
      L.father[q] = fa[q];
    }
    return L;
  }
 private:
  ma_cluster_tree_t* h_ = nullptr;
};
class MlfmmSystem : public math_solvers::LinearOperator {
 public:
  MlfmmSystem(const BemPlan& plan, const ClusterTree& tree, const PhysicsParams& p) : n_(plan.num_dofs()) {
    const ma_physics_t ph = detail::phys(p);
    math_solvers::solver_check(ma_op_create_mlfmm(plan.handle(), tree.handle(), &ph, &op_));
  }
  ~MlfmmSystem() override { if (op_) ma_op_destroy(op_); }
  MlfmmSystem(const MlfmmSystem&) = delete;
  MlfmmSystem& operator=(const MlfmmSystem&) = delete;
  std::vector<Complex64> matvec(const std::vector<Complex64>& x) const { return apply(x); }
  size_t num_rows() const override { return n_; }
  size_t num_cols() const override { return n_; }
  ma_op_t* handle() const override { return op_; }
 private:
  size_t n_; ma_op_t* op_ = nullptr;
};

// the `for freq in frequencies` loop of bin/room_simulator_bem.rs:329 (assemble, incident RHS, solve per frequency) as one call:
// beta = i beta_scale / k per frequency (BemSolver, bem_solver.rs:225, 366); devices.size() > 1 shards the frequencies
// (frequency f on devices[f mod ndev]). Returns the surface solutions, one vector per frequency; status[f] is MA_OK or MA_ERR_SINGULAR.
inline std::vector<std::vector<Complex64>> solve_frequency_sweep(const std::vector<Element>& elements, const std::vector<double>& nodes,
                                                                 const std::vector<double>& frequencies_hz, double speed_of_sound, double beta_scale,
                                                                 const IncidentField& incident, const std::vector<int32_t>& devices = {0},
                                                                 std::vector<int32_t>* status = nullptr, double harmonic_factor = 1.0, double tau = 1.0) {
  detail::Flat F; detail::flatten(elements, nodes, F);
  const size_t nf = frequencies_hz.size(), n = (size_t)F.c.n_elem;
  std::vector<Complex64> X(nf * n);
  std::vector<int32_t> st(nf, 0);
  const int rc = ma_bem_solve_sweep_multi(&F.c, devices.data(), (int32_t)devices.size(), (int32_t)nf, frequencies_hz.data(), speed_of_sound, harmonic_factor, tau,
                                          beta_scale, incident.kind, incident.v, incident.amp.real(), incident.amp.imag(), 3, reinterpret_cast<ma_c64*>(X.data()), st.data());
  if (rc != MA_OK) throw BemError(rc, ma_last_error_string());
  if (status) *status = st;
  std::vector<std::vector<Complex64>> out(nf);
  for (size_t f = 0; f < nf; ++f) out[f].assign(X.begin() + (std::ptrdiff_t)(f * n), X.begin() + (std::ptrdiff_t)((f + 1) * n));
  return out;
}

// The same loop behind a handle that a driver keeps beside its mesh (round 4; ma_bem_sweep_t): the device plan of the mesh, the LU plan,
// its streams, the systems in flight, the spares of the assembly-ahead and the parked solutions are made ONCE; every solve() is one
// ma_bem_sweep_run. room_simulator_bem.rs:243-256 builds the mesh once, :328-360 walks the frequencies -- once per source position.
class FrequencySweep {
 public:
  FrequencySweep(const std::vector<Element>& elements, const std::vector<double>& nodes, size_t max_frequencies, int device = 0, int slots = 3) {
    detail::Flat F; detail::flatten(elements, nodes, F);
    n_ = (size_t)F.c.n_elem;
    int rc = ma_bem_plan_create(&F.c, device, &plan_);
    if (rc == MA_OK) rc = ma_bem_sweep_create(plan_, slots, (int32_t)max_frequencies, &sweep_);
    if (rc != MA_OK) { const std::string msg = ma_last_error_string(); release(); throw BemError(rc, msg); }
  }
  FrequencySweep(const FrequencySweep&) = delete;
  FrequencySweep& operator=(const FrequencySweep&) = delete;
  ~FrequencySweep() { release(); }
  size_t num_dofs() const { return n_; }
  // surface solutions, one vector per frequency; status[f] is MA_OK or MA_ERR_SINGULAR
  std::vector<std::vector<Complex64>> solve(const std::vector<double>& frequencies_hz, double speed_of_sound, double beta_scale, const IncidentField& incident,
                                            std::vector<int32_t>* status = nullptr, double harmonic_factor = 1.0, double tau = 1.0) {
    const size_t nf = frequencies_hz.size();
    std::vector<Complex64> X(nf * n_);
    std::vector<int32_t> st(nf, 0);
    const int rc = ma_bem_sweep_run(sweep_, (int32_t)nf, frequencies_hz.data(), speed_of_sound, harmonic_factor, tau, beta_scale, incident.kind, incident.v,
                                    incident.amp.real(), incident.amp.imag(), reinterpret_cast<ma_c64*>(X.data()), st.data());
    if (rc != MA_OK && rc != MA_ERR_SINGULAR) throw BemError(rc, ma_last_error_string());
    if (status) *status = st;
    std::vector<std::vector<Complex64>> out(nf);
    for (size_t f = 0; f < nf; ++f) out[f].assign(X.begin() + (std::ptrdiff_t)(f * n_), X.begin() + (std::ptrdiff_t)((f + 1) * n_));
    return out;
  }
 private:
  void release() { if (sweep_) ma_bem_sweep_destroy(sweep_); if (plan_) ma_bem_plan_destroy(plan_); sweep_ = nullptr; plan_ = nullptr; }
  size_t n_ = 0; ma_bem_plan_t* plan_ = nullptr; ma_bem_sweep_t* sweep_ = nullptr;
};

}  // namespace math_bem
