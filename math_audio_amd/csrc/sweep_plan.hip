// sweep_plan.hip — the frequency loop of the BEM drivers as one device-resident call
// (math-bem/bin/room_simulator_bem.rs:329-360 and BemSolver::solve, bem_solver.rs:355-480: per frequency
// PhysicsParams::new, beta = burton_miller_beta_scaled, build_tbem_system_with_beta, compute_rhs_with_beta, lu_solve).
// Built on the public entry points only: the systems go through the staged plan API as a pipeline -- `slots` of them in
// HBM at a time, slot s a quarter of a factorisation behind slot s-1 (see ma_lu_plan_stage_*) -- without a host
// synchronisation inside; the solutions are parked on the device and travel back once at the end.
//
// Multi-GPU: frequencies are independent, so ma_bem_solve_sweep_multi gives device d the
// frequencies f = d, d + ndev, ... : one host thread per device, each with its own BEM plan, LU plan, stream and buffers on
// ITS device; no data-path collective, the solutions land in the caller's X_out rows directly.
#include "ma_common.hpp"
#include <vector>
#include <algorithm>
#include <cmath>
#include <thread>
#include <string>
#include <chrono>

using namespace ma;

namespace {

struct SweepArgs {
  double speed_of_sound, harmonic_factor, tau, beta_scale; int incident_kind; const double* incident_vec3; double amp_re, amp_im; int32_t slots;
};

// frequencies first, first + stride, ... of frequencies_hz[0..n_freq) on the plan's own device; X_out / status are indexed by
// the GLOBAL frequency index
int sweep_on_plan_device(ma_bem_plan_t* plan, int32_t n_freq, const double* frequencies_hz, int32_t first, int32_t stride, const SweepArgs& a,
                         ma_c64* X_out, int32_t* status_or_null) {
  int32_t n = 0;
  int rc = ma_bem_plan_num_dofs(plan, &n);
  if (rc) return rc;
  int device = 0;
  if ((rc = ma_bem_plan_device(plan, &device))) return rc;
  // everything this call allocates lives on the PLAN's device, whatever device the calling thread had selected
  MA_HIP(hipSetDevice(device));
  std::vector<int> mine;
  for (int f = first; f < n_freq; f += stride) mine.push_back(f);
  const int n_mine = (int)mine.size();
  if (n_mine == 0) return MA_OK;
  int32_t slots = a.slots;
  if (slots < 1) slots = 3;
  if (slots > 4) slots = 4;
  if (slots > n_mine) slots = n_mine;
  ma_lu_plan_t* lu = nullptr;
  if ((rc = ma_lu_plan_create(n, device, &lu))) return rc;
  // the sweep's own stream: the plan's big-update stream when the plan splits the chip (that stream is masked to the update CUs,
  // and a stream more would be one hardware queue more: profiles/r03_lu_panel_experiments.md), a stream of its own otherwise
  hipStream_t st = nullptr;
  bool own_stream = false;
  { void* ms = nullptr; if (ma_lu_plan_main_stream(lu, &ms) == MA_OK && ms) st = (hipStream_t)ms; }
  if (!st) {
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { set_error("sweep: stream creation failed"); ma_lu_plan_destroy(lu); return MA_ERR_HIP; }
    own_stream = true;
  }
  std::vector<void*> dA((size_t)slots, nullptr), dx((size_t)slots, nullptr);
  auto cleanup = [&]() { for (void* p : dA) if (p) (void)hipFree(p); for (void* p : dx) if (p) (void)hipFree(p); ma_lu_plan_destroy(lu); if (own_stream) (void)hipStreamDestroy(st); };
  for (int s = 0; s < slots; ++s)
    if (hipMalloc(&dA[(size_t)s], sizeof(ma_c64) * (size_t)n * (size_t)n) != hipSuccess || hipMalloc(&dx[(size_t)s], sizeof(ma_c64) * (size_t)n) != hipSuccess) {
      set_error("sweep: %d systems of %d x %d do not fit device %d", slots, n, n, device);
      cleanup();
      return MA_ERR_NOMEM;
    }
  int worst = MA_OK;
  auto assemble = [&](int f, int s) -> int {
    const double freq = frequencies_hz[f];
    ma_physics_t ph;
    ph.wave_number = 2.0 * 3.14159265358979323846 * freq / a.speed_of_sound;     // PhysicsParams::new, types.rs:39-58
    ph.harmonic_factor = a.harmonic_factor; ph.tau = a.tau; ph.gamma = 1.0;
    const double bim = a.tau > 0.0 ? a.harmonic_factor * a.beta_scale / ph.wave_number : 0.0;   // burton_miller_beta_scaled, types.rs:144-150
    int r = ma_bem_plan_assemble_dev(plan, &ph, 0.0, bim, dA[(size_t)s], dx[(size_t)s], st);
    if (!r) r = ma_bem_plan_incident_rhs_dev(plan, &ph, 0.0, bim, a.incident_kind, a.incident_vec3, a.amp_re, a.amp_im, 1, dx[(size_t)s], st);
    return r;
  };
  // Assembly AHEAD, in PIECES, in the staged schedule. Two sets of (up to three) spare systems: while the slots consume one set --
  // a slot that begins a system swaps its matrix with the spare that holds it -- the other set's systems are assembled by
  // ma_bem_plan_assemble_multi_part_dev (their far pairs share one pass over the quadrature points) in pieces of the far pairs'
  // rows, one piece per round in the rounds just before a slot begins: there the sum of the slots' updates is smallest and the
  // stream would wait for the slots' panel chains. What has not been issued when a system is needed is issued then. Only when the
  // spares fit comfortably (MA_SWEEP_ASM_AHEAD=1: every system assembled when its slot begins).
  int ahead = 3;
  if (const char* ea = getenv("MA_SWEEP_ASM_AHEAD")) ahead = std::max(1, std::min(3, atoi(ea)));
  if (ahead > n_mine) ahead = std::max(1, (int)n_mine);
  int ppp = 4;                                                  // pieces per slot begin
  if (const char* ep = getenv("MA_SWEEP_ASM_PIECES")) ppp = std::max(1, std::min(16, atoi(ep)));
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || 2.0 * (double)ahead * 16.0 * (double)n * (double)n > 0.5 * (double)free_b) ahead = 1;
  }
  struct AsmSet { void* A[3] = {}; void* x[3] = {}; int first = -1, cnt = 0, part = 0; unsigned taken = 0; ma_physics_t ph[3]; double br[3], bi[3]; };
  AsmSet sets[2];
  auto free_spares = [&]() { for (auto& t : sets) for (int q = 0; q < 3; ++q) { if (t.A[q]) (void)hipFree(t.A[q]); if (t.x[q]) (void)hipFree(t.x[q]); t.A[q] = t.x[q] = nullptr; } };
  if (ahead > 1) {
    bool ok = true;
    for (auto& t : sets) for (int q = 0; q < ahead && ok; ++q)
      ok = hipMalloc(&t.A[q], sizeof(ma_c64) * (size_t)n * (size_t)n) == hipSuccess && hipMalloc(&t.x[q], sizeof(ma_c64) * (size_t)n) == hipSuccess;
    if (!ok) { free_spares(); ahead = 1; (void)hipGetLastError(); }
  }
  const int nparts = ppp * ahead;
  auto set_idle = [](const AsmSet& t) { return t.first < 0 || t.taken == (1u << t.cnt) - 1u; };
  auto start_job = [&](AsmSet& t, int first_i) {
    t.first = first_i; t.cnt = std::min(ahead, n_mine - first_i); t.part = 0; t.taken = 0;
    for (int q = 0; q < t.cnt; ++q) {
      const double freq = frequencies_hz[mine[(size_t)(first_i + q)]];
      t.ph[q].wave_number = 2.0 * 3.14159265358979323846 * freq / a.speed_of_sound;
      t.ph[q].harmonic_factor = a.harmonic_factor; t.ph[q].tau = a.tau; t.ph[q].gamma = 1.0;
      t.br[q] = 0.0; t.bi[q] = a.tau > 0.0 ? a.harmonic_factor * a.beta_scale / t.ph[q].wave_number : 0.0;
    }
  };
  auto issue_part = [&](AsmSet& t) -> int {
    int r = ma_bem_plan_assemble_multi_part_dev(plan, t.cnt, t.ph, t.br, t.bi, t.A, t.x, t.part, nparts, st);
    if (r) return r;
    if (++t.part == nparts)
      for (int q = 0; q < t.cnt && !r; ++q)
        r = ma_bem_plan_incident_rhs_dev(plan, &t.ph[q], 0.0, t.bi[q], a.incident_kind, a.incident_vec3, a.amp_re, a.amp_im, 1, t.x[q], st);
    return r;
  };
  // system of this device's i-th frequency into slot s
  auto take = [&](int i, int s) -> int {
    if (ahead <= 1) return assemble(mine[(size_t)i], s);
    AsmSet* t = nullptr;
    for (auto& c : sets) if (c.first >= 0 && i >= c.first && i < c.first + c.cnt && !(c.taken >> (i - c.first) & 1u)) t = &c;
    if (!t) {
      for (auto& c : sets) if (!t && set_idle(c)) t = &c;
      if (!t) { set_error("sweep: no spare system for frequency %d", i); return MA_ERR_INVALID; }
      start_job(*t, i);
      AsmSet& o = t == &sets[0] ? sets[1] : sets[0];
      if (set_idle(o) && i + ahead < n_mine) start_job(o, i + ahead);       // the set after this one: in pieces, from now on
    }
    int r = MA_OK;
    while (t->part < nparts && !r) r = issue_part(*t);                       // not finished in the gaps: the rest now
    if (r) return r;
    const int q = i - t->first;
    std::swap(dA[(size_t)s], t->A[q]); std::swap(dx[(size_t)s], t->x[q]);
    t->taken |= 1u << q;
    if (set_idle(*t)) {                                                      // free again: the systems after the other set's
      AsmSet& o = t == &sets[0] ? sets[1] : sets[0];
      const int nxt = (o.first >= 0 && o.first + o.cnt - 1 >= i) ? o.first + o.cnt : i + 1;
      if (nxt < n_mine) start_job(*t, nxt); else t->first = -1;
    }
    return MA_OK;
  };
  // after the updates of round r: one piece of the set being assembled, in the last ppp rounds before a slot begins
  auto assembly_tick = [&](int r, int spacing) -> int {
    if (ahead <= 1 || (r % spacing) < spacing - ppp) return MA_OK;
    for (auto& t : sets) if (t.first >= 0 && t.part < nparts && t.taken == 0) return issue_part(t);
    return MA_OK;
  };
  int32_t G = 0;
  ma_c64* dX = nullptr; int32_t* dinfo = nullptr;
  const bool staged = ma_lu_plan_num_blocks(lu, &G) == MA_OK && G > 0 && ma_lu_plan_stage_reset(lu, st) == MA_OK;
  if (!staged) { free_spares(); ahead = 1; }
  if (staged) {
    if (hipMalloc(&dX, sizeof(ma_c64) * (size_t)n_mine * (size_t)n) != hipSuccess || hipMalloc(&dinfo, sizeof(int32_t) * (size_t)n_mine) != hipSuccess) {
      if (dX) (void)hipFree(dX);
      set_error("sweep: the solutions of %d frequencies do not fit the device", n_mine);
      free_spares();
      cleanup();
      return MA_ERR_NOMEM;
    }
    std::vector<int> off((size_t)slots);
    int32_t spacing = std::max(1, (G + slots) / (slots + 1));
    (void)ma_lu_plan_stage_spacing(lu, slots, &spacing);                                          // what the plan's kernels were measured best with
    for (int s = 0; s < slots; ++s) off[(size_t)s] = s * spacing;
    for (int r = 0; !rc; ++r) {
      int32_t sl[4], bl[4]; int cnt = 0; bool live = false;
      for (int s = 0; s < slots && !rc; ++s) {
        const int lr = r - off[(size_t)s];
        if (lr < 0) { live = true; continue; }
        const int i = s + slots * (lr / G), g = lr % G;       // i: index into this device's frequencies
        if (i >= n_mine) continue;
        live = true;
        if (g == 0) {
          rc = take(i, s);
          if (!rc) rc = ma_lu_plan_stage_begin(lu, s, dA[(size_t)s], dx[(size_t)s], 1, st);
        }
        sl[cnt] = s; bl[cnt] = g; ++cnt;
      }
      if (!live || rc) break;
      if (cnt) rc = ma_lu_plan_stage_round(lu, cnt, sl, bl, st);
      for (int q = 0; q < cnt && !rc; ++q) {
        if (bl[q] != G - 1) continue;
        const int s = sl[q], i = s + slots * ((r - off[(size_t)s]) / G);
        rc = ma_lu_plan_stage_finish(lu, s, st);
        if (!rc && hipMemcpyAsync(dX + (size_t)i * (size_t)n, dx[(size_t)s], sizeof(ma_c64) * (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess) { set_error("sweep: parking a solution failed"); rc = MA_ERR_HIP; }
        if (!rc) rc = ma_lu_plan_stage_info_dev(lu, s, dinfo + i, st);
      }
      if (!rc) rc = assembly_tick(r, spacing);
    }
    if (!rc) {
      int stt = ma_lu_plan_status(lu, st);                     // synchronises; an abandoned panel (poisoned plan) surfaces here
      if (stt != MA_OK && stt != MA_ERR_SINGULAR) rc = stt;
    }
    if (!rc) {
      std::vector<int32_t> hinfo((size_t)n_mine);
      std::vector<ma_c64> hX;
      hipError_t e = hipMemcpy(hinfo.data(), dinfo, sizeof(int32_t) * (size_t)n_mine, hipMemcpyDeviceToHost);
      if (stride == 1 && first == 0) { if (e == hipSuccess) e = hipMemcpy(X_out, dX, sizeof(ma_c64) * (size_t)n_mine * (size_t)n, hipMemcpyDeviceToHost); }
      else for (int i = 0; i < n_mine && e == hipSuccess; ++i)
        e = hipMemcpy(X_out + (size_t)mine[(size_t)i] * (size_t)n, dX + (size_t)i * (size_t)n, sizeof(ma_c64) * (size_t)n, hipMemcpyDeviceToHost);
      if (e != hipSuccess) { set_error("sweep: copy back failed: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; }
      for (int i = 0; i < n_mine && !rc; ++i) {
        const int sf = hinfo[(size_t)i] ? MA_ERR_SINGULAR : MA_OK;
        if (status_or_null) status_or_null[mine[(size_t)i]] = sf;
        if (sf != MA_OK) worst = sf;
      }
    }
    if (rc) (void)hipDeviceSynchronize();                       // nothing may still be running on buffers that are about to go
    (void)hipFree(dX); (void)hipFree(dinfo);
    free_spares();
    cleanup();
    return rc ? rc : worst;
  }
  // look-ahead lanes switched off (MA_LU_LOOKAHEAD=0 / MA_LU_PANEL_OVERLAP=0): lock-step batches
  for (int i0 = 0; i0 < n_mine && !rc; i0 += slots) {
    const int cnt = std::min((int)slots, n_mine - i0);
    for (int s = 0; s < cnt && !rc; ++s) rc = assemble(mine[(size_t)(i0 + s)], s);
    if (rc) break;
    rc = ma_lu_plan_factor_solve_batch_dev(lu, cnt, dA.data(), dx.data(), 1, st);
    if (rc) break;
    int stt = ma_lu_plan_status(lu, st);
    if (stt != MA_OK && stt != MA_ERR_SINGULAR) { rc = stt; break; }
    for (int s = 0; s < cnt; ++s) {
      const int f = mine[(size_t)(i0 + s)];
      if (status_or_null) status_or_null[f] = stt;            // a singular member marks its whole batch; the caller may re-run those singly
      if (hipMemcpy(X_out + (size_t)f * (size_t)n, dx[(size_t)s], sizeof(ma_c64) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) { set_error("sweep: copy back failed"); rc = MA_ERR_HIP; }
    }
    if (stt != MA_OK) worst = stt;
  }
  if (rc) (void)hipDeviceSynchronize();
  cleanup();
  return rc ? rc : worst;
}

}  // namespace

extern "C" {

int ma_bem_solve_sweep(ma_bem_plan_t* plan, int32_t n_freq, const double* frequencies_hz, double speed_of_sound, double harmonic_factor, double tau,
                       double beta_scale, int incident_kind, const double* incident_vec3, double amp_re, double amp_im, int32_t slots,
                       ma_c64* X_out, int32_t* status_or_null) {
  MA_REQUIRE(plan && n_freq > 0 && frequencies_hz && incident_vec3 && X_out, MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(speed_of_sound > 0.0, MA_ERR_INVALID, "speed of sound must be positive");
  const SweepArgs a{speed_of_sound, harmonic_factor, tau, beta_scale, incident_kind, incident_vec3, amp_re, amp_im, slots};
  int prev = -1;
  const bool had = hipGetDevice(&prev) == hipSuccess;
  const int rc = sweep_on_plan_device(plan, n_freq, frequencies_hz, 0, 1, a, X_out, status_or_null);
  if (had) (void)hipSetDevice(prev);                            // the caller's current device is left as it was
  return rc;
}

// The sharding rule of the multi-device sweep, exposed so that callers, tests and bench.py agree on it: frequency f belongs
// to device slot f mod ndev (room_simulator_bem.rs:329's loop dealt round-robin).
int ma_sweep_owner(int32_t frequency_index, int32_t ndev) { return ndev > 0 ? frequency_index % ndev : 0; }

// the same with a per-device account for the caller (bench.py --inlib): device_seconds[d] = wall time of device d's sweep (its plan
// creation excluded), device_setup_seconds[d] = its plan creation, device_frequencies[d] = frequencies it solved; any may be NULL
int ma_bem_solve_sweep_multi_timed(const ma_mesh_t* mesh, const int32_t* devices, int32_t ndev, int32_t n_freq, const double* frequencies_hz, double speed_of_sound,
                                   double harmonic_factor, double tau, double beta_scale, int incident_kind, const double* incident_vec3, double amp_re, double amp_im,
                                   int32_t slots, ma_c64* X_out, int32_t* status_or_null, double* device_seconds, double* device_setup_seconds, int32_t* device_frequencies) {
  MA_REQUIRE(mesh && devices && ndev >= 1 && ndev <= 64 && n_freq > 0 && frequencies_hz && incident_vec3 && X_out, MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(speed_of_sound > 0.0, MA_ERR_INVALID, "speed of sound must be positive");
  int count = 0;
  int rc = ma_device_count(&count);
  if (rc) return rc;
  MA_REQUIRE(count > 0, MA_ERR_NO_DEVICE, "no gfx950 device visible");
  for (int d = 0; d < ndev; ++d) {
    MA_REQUIRE(devices[d] >= 0 && devices[d] < count, MA_ERR_INVALID, "device %d (entry %d) outside 0..%d", devices[d], d, count - 1);
    // (test hook MA_TEST_ALLOW_DUPLICATE_DEVICES=1: several host threads on one GPU, so that a one-GPU box exercises the sharding)
    for (int o = 0; o < d; ++o) MA_REQUIRE(devices[o] != devices[d] || getenv("MA_TEST_ALLOW_DUPLICATE_DEVICES"), MA_ERR_INVALID, "device %d listed twice", devices[d]);
  }
  const SweepArgs a{speed_of_sound, harmonic_factor, tau, beta_scale, incident_kind, incident_vec3, amp_re, amp_im, slots};
  std::vector<int> rcs((size_t)ndev, MA_OK);
  std::vector<std::string> texts((size_t)ndev);
  auto work = [&](int d) {
    // one host thread per device: its own plans, stream and buffers; errors are thread-local and carried back as text
    ma_bem_plan_t* plan = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    int r = ma_bem_plan_create(mesh, devices[d], &plan);
    const auto t1 = std::chrono::steady_clock::now();
    if (!r) r = sweep_on_plan_device(plan, n_freq, frequencies_hz, d, ndev, a, X_out, status_or_null);
    const auto t2 = std::chrono::steady_clock::now();
    if (r && r != MA_ERR_SINGULAR) texts[(size_t)d] = ma_last_error_string();
    if (plan) ma_bem_plan_destroy(plan);
    rcs[(size_t)d] = r;
    if (device_setup_seconds) device_setup_seconds[d] = std::chrono::duration<double>(t1 - t0).count();
    if (device_seconds) device_seconds[d] = std::chrono::duration<double>(t2 - t1).count();
    if (device_frequencies) { int c = 0; for (int f = d; f < n_freq; f += ndev) ++c; device_frequencies[d] = c; }
  };
  int prev = -1;
  const bool had = hipGetDevice(&prev) == hipSuccess;       // work(0) runs on the calling thread and selects devices[0]: the caller's device is restored below
  std::vector<std::thread> th;
  for (int d = 1; d < ndev; ++d) th.emplace_back(work, d);
  work(0);
  for (auto& t : th) t.join();
  if (had) (void)hipSetDevice(prev);
  int worst = MA_OK;
  for (int d = 0; d < ndev; ++d) {
    if (rcs[(size_t)d] == MA_OK) continue;
    if (rcs[(size_t)d] == MA_ERR_SINGULAR) { if (worst == MA_OK) worst = MA_ERR_SINGULAR; continue; }
    set_error("sweep on device %d: %s", devices[d], texts[(size_t)d].c_str());
    return rcs[(size_t)d];
  }
  return worst;
}

int ma_bem_solve_sweep_multi(const ma_mesh_t* mesh, const int32_t* devices, int32_t ndev, int32_t n_freq, const double* frequencies_hz, double speed_of_sound,
                             double harmonic_factor, double tau, double beta_scale, int incident_kind, const double* incident_vec3, double amp_re, double amp_im,
                             int32_t slots, ma_c64* X_out, int32_t* status_or_null) {
  return ma_bem_solve_sweep_multi_timed(mesh, devices, ndev, n_freq, frequencies_hz, speed_of_sound, harmonic_factor, tau, beta_scale, incident_kind, incident_vec3,
                                        amp_re, amp_im, slots, X_out, status_or_null, nullptr, nullptr, nullptr);
}

}  // extern "C"
