// Host side of AmgPreconditioner::from_csr (math-solvers/src/preconditioners/amg.rs:276-372) and of the CSR algebra it calls
// (math-solvers/src/sparse/csr.rs). Setup stays on the host as in the reference: the loops below are the reference's, in
// its order, so that the coarse sets are identical and the values equal to the last bit of every sum; the product path then applies
// the cycle on the device (op_plan.hip). Nothing here is a fallback for a device step.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <thread>
#include "amg_setup.hpp"

namespace ma {
namespace {

inline double cnorm(c64 z) { return std::sqrt(z.re * z.re + z.im * z.im); }                 // traits.rs:94-96
inline c64 cinv(c64 z) { const double d = z.re * z.re + z.im * z.im; return c64{z.re / d, -z.im / d}; }   // traits.rs:159-162
inline c64 cmul(c64 a, c64 b) { return c64{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline c64 cadd(c64 a, c64 b) { return c64{a.re + b.re, a.im + b.im}; }
inline c64 cneg(c64 a) { return c64{0.0 - a.re, 0.0 - a.im}; }                             // T::zero() - x

struct Trip { int64_t r, c; c64 v; };

// CsrMatrix::from_triplets (csr.rs:135-205): stable sort by (row, col), equal positions accumulate in that order
HostCsr from_triplets(int64_t nr, int64_t nc, std::vector<Trip>& t) {
  HostCsr m; m.nr = nr; m.nc = nc; m.ptr.assign((size_t)nr + 1, 0);
  if (t.empty()) return m;
  auto before = [](const Trip& a, const Trip& b) { return a.r != b.r ? a.r < b.r : a.c < b.c; };
  if (!std::is_sorted(t.begin(), t.end(), before)) std::stable_sort(t.begin(), t.end(), before);   // a stable sort of a sorted list is the list
  m.col.reserve(t.size()); m.val.reserve(t.size());
  int64_t pr = -1, pc = -1;
  for (const Trip& e : t) {
    if (e.r == pr && e.c == pc) { m.val.back() = cadd(m.val.back(), e.v); continue; }
    m.val.push_back(e.v); m.col.push_back(e.c); m.ptr[(size_t)e.r + 1] += 1; pr = e.r; pc = e.c;
  }
  for (int64_t i = 0; i < nr; ++i) m.ptr[(size_t)i + 1] += m.ptr[(size_t)i];
  return m;
}

inline c64 get(const HostCsr& m, int64_t i, int64_t j) {                                   // csr.rs:340-347
  for (int64_t q = m.ptr[(size_t)i]; q < m.ptr[(size_t)i + 1]; ++q) if (m.col[(size_t)q] == j) return m.val[(size_t)q];
  return c64{0.0, 0.0};
}

// CsrMatrix::matmul (csr.rs:594-651): the rows are independent, blocks of rows run on host threads. Every row is formed as the
// reference forms it (products in storage order, stable sort by column, sums in that order, norm <= 1e-15 dropped); its from_triplets
// pass over the rows' triplets -- already in (row, column) order with one entry per position -- is the concatenation done here.
struct RowBlock { std::vector<int64_t> cnt, col; std::vector<c64> val; };
void matmul_rows(const HostCsr& a, const HostCsr& b, int64_t i0, int64_t i1, RowBlock& out) {
  std::vector<std::pair<int64_t, c64>> rd;
  out.cnt.assign((size_t)(i1 - i0), 0);
  for (int64_t i = i0; i < i1; ++i) {
    rd.clear();
    for (int64_t p = a.ptr[(size_t)i]; p < a.ptr[(size_t)i + 1]; ++p) {
      const int64_t k = a.col[(size_t)p]; const c64 aik = a.val[(size_t)p];
      for (int64_t q = b.ptr[(size_t)k]; q < b.ptr[(size_t)k + 1]; ++q) rd.push_back({b.col[(size_t)q], cmul(aik, b.val[(size_t)q])});
    }
    if (rd.empty()) continue;
    std::stable_sort(rd.begin(), rd.end(), [](const std::pair<int64_t, c64>& x, const std::pair<int64_t, c64>& y) { return x.first < y.first; });
    const size_t before = out.col.size();
    int64_t cj = rd[0].first; c64 cv = rd[0].second;
    for (size_t e = 1; e < rd.size(); ++e) {
      if (rd[e].first == cj) cv = cadd(cv, rd[e].second);
      else { if (cnorm(cv) > 1e-15) { out.col.push_back(cj); out.val.push_back(cv); } cj = rd[e].first; cv = rd[e].second; }
    }
    if (cnorm(cv) > 1e-15) { out.col.push_back(cj); out.val.push_back(cv); }
    out.cnt[(size_t)(i - i0)] = (int64_t)(out.col.size() - before);
  }
}
HostCsr matmul(const HostCsr& a, const HostCsr& b) {
  HostCsr m; m.nr = a.nr; m.nc = b.nc; m.ptr.assign((size_t)a.nr + 1, 0);
  if (a.nr == 0 || b.nc == 0 || a.val.empty() || b.val.empty()) return m;
  const int T = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, a.nr / 4096));
  std::vector<RowBlock> parts((size_t)T);
  if (T == 1) matmul_rows(a, b, 0, a.nr, parts[0]);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t]() { matmul_rows(a, b, a.nr * t / T, a.nr * (t + 1) / T, parts[(size_t)t]); });
    for (auto& x : th) x.join();
  }
  std::vector<int64_t> base((size_t)T + 1, 0);
  for (int t = 0; t < T; ++t) {
    const int64_t i0 = a.nr * t / T;
    for (size_t r = 0; r < parts[(size_t)t].cnt.size(); ++r) m.ptr[(size_t)i0 + r + 1] = parts[(size_t)t].cnt[r];
    base[(size_t)t + 1] = base[(size_t)t] + (int64_t)parts[(size_t)t].col.size();
  }
  for (int64_t i = 0; i < a.nr; ++i) m.ptr[(size_t)i + 1] += m.ptr[(size_t)i];
  m.col.resize((size_t)base[(size_t)T]); m.val.resize((size_t)base[(size_t)T]);
  auto place = [&](int t) {
    std::copy(parts[(size_t)t].col.begin(), parts[(size_t)t].col.end(), m.col.begin() + base[(size_t)t]);
    std::copy(parts[(size_t)t].val.begin(), parts[(size_t)t].val.end(), m.val.begin() + base[(size_t)t]);
  };
  if (T == 1) place(0);
  else { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(place, t); for (auto& x : th) x.join(); }
  return m;
}

HostCsr transpose(const HostCsr& m) {                                                       // amg.rs:810-822
  // from_triplets over (j, i, v) sorts by (j, i): with one entry per position -- what from_triplets itself hands over -- that is a
  // counting transpose, row j filled in ascending i
  HostCsr t; t.nr = m.nc; t.nc = m.nr; t.ptr.assign((size_t)m.nc + 1, 0); t.col.resize(m.val.size()); t.val.resize(m.val.size());
  for (int64_t c : m.col) t.ptr[(size_t)c + 1] += 1;
  for (int64_t j = 0; j < m.nc; ++j) t.ptr[(size_t)j + 1] += t.ptr[(size_t)j];
  std::vector<int64_t> cur(t.ptr.begin(), t.ptr.end() - 1);
  for (int64_t i = 0; i < m.nr; ++i)
    for (int64_t q = m.ptr[(size_t)i]; q < m.ptr[(size_t)i + 1]; ++q) { const int64_t pos = cur[(size_t)m.col[(size_t)q]]++; t.col[(size_t)pos] = i; t.val[(size_t)pos] = m.val[(size_t)q]; }
  return t;
}

// compute_strength_matrix (amg.rs:418-474), as CSR of column indices
void strength(const HostCsr& m, double theta, std::vector<int64_t>& sp, std::vector<int64_t>& sj) {
  sp.assign((size_t)m.nr + 1, 0);
  std::vector<char> strong(m.val.size(), 0);                 // per entry, rows on host threads; the lists are then read off in row order
  host_parallel_for(m.nr, 8192, [&](long long r0, long long r1) {
    for (int64_t i = r0; i < r1; ++i) {
      double mx = 0.0;
      for (int64_t q = m.ptr[(size_t)i]; q < m.ptr[(size_t)i + 1]; ++q) if (m.col[(size_t)q] != i) { const double nv = cnorm(m.val[(size_t)q]); if (nv > mx) mx = nv; }
      const double thr = theta * mx;
      int64_t cnt = 0;
      for (int64_t q = m.ptr[(size_t)i]; q < m.ptr[(size_t)i + 1]; ++q) if (m.col[(size_t)q] != i && cnorm(m.val[(size_t)q]) >= thr) { strong[(size_t)q] = 1; ++cnt; }
      sp[(size_t)i + 1] = cnt;
    }
  });
  for (int64_t i = 0; i < m.nr; ++i) sp[(size_t)i + 1] += sp[(size_t)i];
  sj.resize((size_t)sp[(size_t)m.nr]);
  host_parallel_for(m.nr, 8192, [&](long long r0, long long r1) {
    for (int64_t i = r0; i < r1; ++i) {
      int64_t o = sp[(size_t)i];
      for (int64_t q = m.ptr[(size_t)i]; q < m.ptr[(size_t)i + 1]; ++q) if (strong[(size_t)q]) sj[(size_t)o++] = m.col[(size_t)q];
    }
  });
}

enum : char { UNDECIDED = 0, COARSE = 1, FINE = 2 };

// coarsen_ruge_stuben (amg.rs:477-532). The reference orders the points once by decreasing lambda (a stable sort) and walks that
// order; the lambda updates inside the walk are not read again. Its scan over all j with strong[j].contains(i) is the transposed
// strength graph, used here directly: the same sets without the O(n^2) scan.
void coarsen_rs(int64_t n, const std::vector<int64_t>& sp, const std::vector<int64_t>& sj, std::vector<char>& pt) {
  pt.assign((size_t)n, UNDECIDED);
  std::vector<int64_t> lam((size_t)n, 0), tp((size_t)n + 1, 0), tj(sj.size());
  for (int64_t j : sj) { lam[(size_t)j] += 1; tp[(size_t)j + 1] += 1; }
  for (int64_t i = 0; i < n; ++i) tp[(size_t)i + 1] += tp[(size_t)i];
  { std::vector<int64_t> fill(tp.begin(), tp.end() - 1);
    for (int64_t i = 0; i < n; ++i) for (int64_t q = sp[(size_t)i]; q < sp[(size_t)i + 1]; ++q) tj[(size_t)fill[(size_t)sj[(size_t)q]]++] = i; }
  std::vector<int64_t> order((size_t)n);
  for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return lam[(size_t)a] > lam[(size_t)b]; });
  for (int64_t i : order) {
    if (pt[(size_t)i] != UNDECIDED) continue;
    pt[(size_t)i] = COARSE;
    for (int64_t q = tp[(size_t)i]; q < tp[(size_t)i + 1]; ++q) if (pt[(size_t)tj[(size_t)q]] == UNDECIDED) pt[(size_t)tj[(size_t)q]] = FINE;
  }
  for (char& p : pt) if (p == UNDECIDED) p = FINE;
}

// coarsen_pmis (amg.rs:535-642): every pass reads the state of the pass before (both builds of the reference do)
void coarsen_pmis(int64_t n, const std::vector<int64_t>& sp, const std::vector<int64_t>& sj, std::vector<char>& pt) {
  pt.assign((size_t)n, UNDECIDED);
  std::vector<double> w((size_t)n);
  for (int64_t i = 0; i < n; ++i) w[(size_t)i] = (double)(sp[(size_t)i + 1] - sp[(size_t)i]) + std::fmod((double)i * 0.0001, 0.001);
  bool changed = true; int it = 0;
  std::vector<char> old;
  while (changed && it < 100) {
    changed = false; ++it;
    old = pt;
    for (int64_t i = 0; i < n; ++i) {
      if (old[(size_t)i] != UNDECIDED) continue;
      bool is_max = true, has_c = false;
      for (int64_t q = sp[(size_t)i]; q < sp[(size_t)i + 1]; ++q) { const int64_t j = sj[(size_t)q]; if (old[(size_t)j] == UNDECIDED && w[(size_t)j] > w[(size_t)i]) { is_max = false; break; } }
      for (int64_t q = sp[(size_t)i]; q < sp[(size_t)i + 1]; ++q) if (old[(size_t)sj[(size_t)q]] == COARSE) { has_c = true; break; }
      if (has_c) { pt[(size_t)i] = FINE; changed = true; }
      else if (is_max) { pt[(size_t)i] = COARSE; changed = true; }
    }
  }
  for (char& p : pt) if (p == UNDECIDED) p = COARSE;
}

// build_interpolation (amg.rs:645-807); interpolation 0 Standard, 1 Extended, 2 Direct
HostCsr build_interpolation(const HostCsr& m, const std::vector<int64_t>& sp, const std::vector<int64_t>& sj, const std::vector<char>& pt,
                            const std::vector<int64_t>& c2f, const ma_amg_config_t& cfg) {
  const int64_t nf = m.nr;
  std::vector<int64_t> f2c((size_t)nf, -1);
  for (size_t c = 0; c < c2f.size(); ++c) f2c[(size_t)c2f[c]] = (int64_t)c;
  auto by_norm_desc = [](const std::pair<int64_t, c64>& a, const std::pair<int64_t, c64>& b) { return cnorm(a.second) > cnorm(b.second); };
  // the rows are independent: blocks of rows on host threads, their triplets joined in row order
  const int T = (int)std::min<int64_t>(host_threads(), std::max<int64_t>(1, nf / 8192));
  std::vector<std::vector<Trip>> parts((size_t)T);
  auto rows = [&](int64_t r0, int64_t r1, std::vector<Trip>& trip) {
  std::vector<int64_t> cn;
  std::vector<std::pair<int64_t, c64>> wts;
  for (int64_t i = r0; i < r1; ++i) {
    if (pt[(size_t)i] == COARSE) { trip.push_back({i, f2c[(size_t)i], c64{1.0, 0.0}}); continue; }
    if (pt[(size_t)i] != FINE) continue;
    const c64 aii = get(m, i, i);
    cn.clear();
    for (int64_t q = sp[(size_t)i]; q < sp[(size_t)i + 1]; ++q) if (pt[(size_t)sj[(size_t)q]] == COARSE) cn.push_back(sj[(size_t)q]);
    if (cn.empty()) continue;
    wts.clear();
    if (cfg.interpolation != 1) {
      c64 sw{0.0, 0.0};
      for (int64_t j : cn) {
        const c64 aij = get(m, i, j);
        if (cnorm(aii) > 1e-15) { const c64 w = cneg(cmul(aij, cinv(aii))); wts.push_back({f2c[(size_t)j], w}); sw = cadd(sw, w); }
      }
      if (cfg.interpolation == 0) {
        c64 weak{0.0, 0.0};
        for (int64_t q = m.ptr[(size_t)i]; q < m.ptr[(size_t)i + 1]; ++q) {
          const int64_t j = m.col[(size_t)q];
          if (j != i && std::find(cn.begin(), cn.end(), j) == cn.end()) weak = cadd(weak, m.val[(size_t)q]);
        }
        if (cnorm(sw) > 1e-15 && cnorm(weak) > 1e-15) {
          const c64 scale = cadd(c64{1.0, 0.0}, cmul(weak, cinv(cmul(aii, sw))));
          for (auto& e : wts) e.second = cmul(e.second, scale);
        }
      }
      if (cfg.trunc_factor > 0.0) {
        double mw = 0.0;
        for (const auto& e : wts) { const double nv = cnorm(e.second); if (nv > mw) mw = nv; }     // fold(0, max)
        const double thr = cfg.trunc_factor * mw;
        wts.erase(std::remove_if(wts.begin(), wts.end(), [&](const std::pair<int64_t, c64>& e) { return !(cnorm(e.second) >= thr); }), wts.end());
        if ((int64_t)wts.size() > (int64_t)cfg.max_interp_elements) { std::stable_sort(wts.begin(), wts.end(), by_norm_desc); wts.resize((size_t)cfg.max_interp_elements); }
      }
    } else {
      for (int64_t j : cn) {
        const c64 aij = get(m, i, j);
        if (cnorm(aii) > 1e-15) wts.push_back({f2c[(size_t)j], cneg(cmul(aij, cinv(aii)))});
      }
      for (int64_t q = sp[(size_t)i]; q < sp[(size_t)i + 1]; ++q) {
        const int64_t k = sj[(size_t)q];
        if (pt[(size_t)k] != FINE) continue;
        const c64 aik = get(m, i, k), akk = get(m, k, k);
        if (cnorm(akk) < 1e-15) continue;
        for (int64_t q2 = sp[(size_t)k]; q2 < sp[(size_t)k + 1]; ++q2) {
          const int64_t j = sj[(size_t)q2];
          if (pt[(size_t)j] != COARSE) continue;
          const c64 akj = get(m, k, j);
          const c64 w = cneg(cmul(cmul(aik, akj), cinv(cmul(aii, akk))));
          const int64_t cj = f2c[(size_t)j];
          bool found = false;
          for (auto& e : wts) if (e.first == cj) { e.second = cadd(e.second, w); found = true; break; }
          if (!found) wts.push_back({cj, w});
        }
      }
      if ((int64_t)wts.size() > (int64_t)cfg.max_interp_elements) { std::stable_sort(wts.begin(), wts.end(), by_norm_desc); wts.resize((size_t)cfg.max_interp_elements); }
    }
    for (const auto& e : wts) trip.push_back({i, e.first, e.second});
  }
  };
  if (T == 1) rows(0, nf, parts[0]);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([&, t]() { rows(nf * t / T, nf * (t + 1) / T, parts[(size_t)t]); });
    for (auto& x : th) x.join();
  }
  std::vector<Trip> trip;
  for (auto& v : parts) { trip.insert(trip.end(), v.begin(), v.end()); std::vector<Trip>().swap(v); }
  return from_triplets(nf, (int64_t)c2f.size(), trip);
}

}  // namespace

int amg_setup_host(const HostCsr& A, const ma_amg_config_t& cfg, std::vector<HostCsr>& As, std::vector<HostCsr>& Ps, std::vector<HostCsr>& Rs,
                   double* grid_complexity, double* operator_complexity) {
  As.clear(); Ps.clear(); Rs.clear();
  As.push_back(A);   // (the caller's copy; level 0 is not uploaded again)
  std::vector<int64_t> sp, sj, c2f;
  std::vector<char> pt;
#ifdef MA_DIAGNOSTICS
  const bool timing = getenv("MA_AMG_TIMING") != nullptr;    // diagnostic build only: phase times of the setup on stderr
#else
  const bool timing = false;
#endif
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  for (int l = 0; l + 1 < cfg.max_levels; ++l) {
    const HostCsr& cur = As.back();
    const int64_t n = cur.nr;
    if (n <= (int64_t)cfg.coarse_size) break;
    const auto t0 = now();
    strength(cur, cfg.strong_threshold, sp, sj);
    const auto t1 = now();
    if (cfg.coarsening == 0) coarsen_rs(n, sp, sj, pt); else coarsen_pmis(n, sp, sj, pt);
    c2f.clear();
    for (int64_t i = 0; i < n; ++i) if (pt[(size_t)i] == COARSE) c2f.push_back(i);
    if (c2f.empty() || (int64_t)c2f.size() >= n) break;
    const auto t2 = now();
    HostCsr P = build_interpolation(cur, sp, sj, pt, c2f, cfg);
    const auto t3 = now();
    HostCsr R = transpose(P);
    const auto t4 = now();
    HostCsr AP = matmul(cur, P);
    const auto t5 = now();
    HostCsr Ac = matmul(R, AP);                                                               // galerkin_product, amg.rs:825-828
    if (timing) fprintf(stderr, "[amg setup] level %d n %lld: strength %.0f coarsen %.0f interpolation %.0f transpose %.0f A*P %.0f R*(AP) %.0f ms\n", l, (long long)n,
                        ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t5, now()));
    Ps.push_back(std::move(P)); Rs.push_back(std::move(R)); As.push_back(std::move(Ac));
  }
  double td = 0.0, tn = 0.0;                                                                  // compute_complexities, amg.rs:837-853
  for (const HostCsr& a : As) { td += (double)a.nr; tn += (double)a.val.size(); }
  if (grid_complexity) *grid_complexity = td / (double)As[0].nr;
  if (operator_complexity) *operator_complexity = tn / (double)As[0].val.size();
  return MA_OK;
}

}  // namespace ma

extern "C" int ma_amg_config_preset(int32_t which, ma_amg_config_t* c) {
  MA_REQUIRE(c, MA_ERR_INVALID, "cfg is NULL");
  MA_REQUIRE(which >= 0 && which <= 4, MA_ERR_INVALID, "preset %d: 0 default, 1 for_bem, 2 for_fem, 3 for_parallel, 4 for_difficult_problems", (int)which);
  *c = ma_amg_config_t{0, 0, 0, 0, 0.25, 25, 50, 1, 1, 0.6667, 0.0, 4, 0};                     // amg.rs:148-167
  if (which == 1) { c->strong_threshold = 0.5; c->coarsening = 1; c->smoother = 1; c->max_interp_elements = 6; }
  if (which == 2) { c->strong_threshold = 0.25; c->coarsening = 0; c->smoother = 2; }
  if (which == 3) { c->coarsening = 1; c->smoother = 0; c->jacobi_weight = 0.8; c->num_pre_smooth = 2; c->num_post_smooth = 2; }
  if (which == 4) { c->coarsening = 0; c->interpolation = 1; c->smoother = 2; c->strong_threshold = 0.25; c->max_interp_elements = 8; c->num_pre_smooth = 2; c->num_post_smooth = 2; }
  return MA_OK;
}
