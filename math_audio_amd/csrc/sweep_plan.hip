// sweep_plan.hip — the frequency loop of the BEM drivers as one device-resident call
// (math-bem/bin/room_simulator_bem.rs:329-360 and BemSolver::solve, bem_solver.rs:355-480: per frequency
// PhysicsParams::new, beta = burton_miller_beta_scaled, build_tbem_system_with_beta, compute_rhs_with_beta, lu_solve).
// Built on the public entry points only: the systems go through the staged plan API as a pipeline -- `slots` of them in
// HBM at a time, slot s a third of a factorisation behind slot s-1 (see ma_lu_plan_stage_*) -- without a host
// synchronisation inside; the solutions are parked on the device and travel back once at the end.
//
// Round 4: the loop lives behind a HANDLE (ma_bem_sweep_t): LU plan, stream, the slots' systems, the spare systems of the
// assembly-ahead and the parked solutions are allocated once and reused by every run, so a caller that sweeps again and again
// (room_simulator_bem.rs runs one sweep per source position) pays no hipMalloc of 6-9 matrices of 1.6 GB per call, and
// bench.py times exactly this entry. ma_bem_solve_sweep is create + run + destroy.
//
// Multi-GPU: frequencies are independent, so ma_bem_solve_sweep_multi gives device d the
// frequencies f = d, d + ndev, ... : one host thread per device, each with its own BEM plan, sweep handle, LU plan, stream and
// buffers on ITS device; no data-path collective, the solutions land in the caller's X_out rows directly.
#include "ma_common.hpp"
#include <vector>
#include <algorithm>
#include <cmath>
#include <thread>
#include <string>
#include <chrono>
#include <new>

using namespace ma;

namespace {
struct SweepArgs {
  double speed_of_sound, harmonic_factor, tau, beta_scale; int incident_kind; const double* incident_vec3; double amp_re, amp_im;
};
struct AsmSet { void* A[3] = {}; void* x[3] = {}; int first = -1, cnt = 0, part = 0; unsigned taken = 0; ma_physics_t ph[3]; double br[3], bi[3]; };
}  // namespace

struct ma_bem_sweep {
  ma_bem_plan_t* plan = nullptr; int device = 0; int32_t n = 0; int32_t slots = 3; int32_t cap = 0;
  ma_lu_plan_t* lu = nullptr; hipStream_t st = nullptr; bool own_stream = false;
  std::vector<void*> dA, dx;
  int ahead = 1, ppp = 4;
  AsmSet sets[2];
  int32_t G = 0, spacing = 1; bool staged = false;
  ma_c64* dX = nullptr; int32_t* dinfo = nullptr;
  // timing of the last run (ma_bem_sweep_set_timing): events around the run and around every piece of assembly, on the sweep's stream
  bool timing = false;
  hipEvent_t ev_run[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> ev_asm; size_t ev_asm_used = 0;
  double last_wall_s = 0.0, last_device_ms = 0.0, last_asm_ms = 0.0; int last_asm_pieces = 0, last_n = 0;
  int last_redone = 0;                                     // frequencies of the last run that were solved a second time (their optimistic factorisation met a rejected panel)
  bool have_last = false;

  void free_spares() { for (auto& t : sets) for (int q = 0; q < 3; ++q) { if (t.A[q]) (void)hipFree(t.A[q]); if (t.x[q]) (void)hipFree(t.x[q]); t.A[q] = t.x[q] = nullptr; } }
  void release() {
    (void)hipSetDevice(device);
    if (st) (void)hipStreamSynchronize(st);
    for (void* p : dA) if (p) (void)hipFree(p);
    for (void* p : dx) if (p) (void)hipFree(p);
    dA.clear(); dx.clear();
    free_spares();
    if (dX) (void)hipFree(dX);
    if (dinfo) (void)hipFree(dinfo);
    dX = nullptr; dinfo = nullptr;
    for (hipEvent_t e : ev_asm) (void)hipEventDestroy(e);
    ev_asm.clear();
    for (auto& e : ev_run) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    if (lu) ma_lu_plan_destroy(lu);
    lu = nullptr;
    if (own_stream && st) (void)hipStreamDestroy(st);
    st = nullptr;
  }
  int asm_mark() {                                         // one event on the sweep's stream, from the pool
    if (!timing) return MA_OK;
    if (ev_asm_used >= ev_asm.size()) { hipEvent_t e; MA_HIP(hipEventCreate(&e)); ev_asm.push_back(e); }
    MA_HIP(hipEventRecord(ev_asm[ev_asm_used++], st));
    return MA_OK;
  }
};

namespace {

// The order in which the staged loop BEGINS the frequencies (pure: no device): slot s begins its j-th system, frequency s + slots j, at
// round s spacing + j blocks; within a round the slots are walked in order. order[q] = the frequency of the q-th begin.
int begin_order(int32_t blocks, int32_t slots, int32_t spacing, int32_t n_freq, std::vector<int32_t>& order) {
  order.clear();
  if (blocks < 1 || slots < 1 || spacing < 1 || n_freq < 0) return MA_ERR_INVALID;
  for (long long r = 0; (int32_t)order.size() < n_freq; ++r) {
    bool live = false;
    for (int s = 0; s < slots; ++s) {
      const long long lr = r - (long long)s * spacing;
      if (lr < 0) { live = true; continue; }
      const long long i = s + (long long)slots * (lr / blocks);
      if (i >= n_freq) continue;
      live = true;
      if (lr % blocks == 0) order.push_back((int32_t)i);
    }
    if (!live) break;
  }
  return MA_OK;
}
bool begins_in_order(int32_t blocks, int32_t slots, int32_t spacing) {
  std::vector<int32_t> o;
  if (begin_order(blocks, slots, spacing, 4 * slots, o)) return false;
  for (size_t q = 0; q < o.size(); ++q) if (o[q] != (int32_t)q) return false;
  return o.size() == (size_t)(4 * slots);
}

// the sweep's pivoting when the caller does not say: tournament (see ma_lu_plan_create_pivoting); MA_SWEEP_PIVOTING=partial|tournament overrides
int default_pivoting() {
  if (const char* e = getenv("MA_SWEEP_PIVOTING")) return (e[0] == 'p' || e[0] == 'P' || e[0] == '0') ? MA_LU_PIVOT_PARTIAL : MA_LU_PIVOT_TOURNAMENT;
  return MA_LU_PIVOT_TOURNAMENT;
}

int sweep_create(ma_bem_plan_t* plan, int32_t slots, int32_t max_frequencies, int32_t pivoting, ma_bem_sweep** out) {
  *out = nullptr;
  int32_t n = 0;
  int rc = ma_bem_plan_num_dofs(plan, &n);
  if (rc) return rc;
  int device = 0;
  if ((rc = ma_bem_plan_device(plan, &device))) return rc;
  // everything the handle allocates lives on the PLAN's device, whatever device the calling thread had selected
  MA_HIP(hipSetDevice(device));
  ma_bem_sweep* S = new (std::nothrow) ma_bem_sweep();
  MA_REQUIRE(S, MA_ERR_NOMEM, "host allocation failed");
  S->plan = plan; S->device = device; S->n = n; S->cap = max_frequencies;
  if (slots < 1) slots = 3;
  if (slots > 4) slots = 4;
  if (slots > max_frequencies) slots = max_frequencies;
  S->slots = slots;
  auto fail = [&](int code) { S->release(); delete S; return code; };
  if ((rc = ma_lu_plan_create_pivoting(n, device, pivoting, &S->lu))) return fail(rc);
  // The sweep runs the plan's speculative panels WITHOUT the fallback behind them (lu_spec.hip: Burton-Miller operators pass the check at
  // every panel) and solves a frequency that did meet a rejected panel again, in the verified mode, after the pipeline has drained
  // (sweep_run, "redo"). MA_SWEEP_SPECULATE=verified|off: the fallback in line / no speculation.
  {
    int mode = MA_LU_SPECULATE_OPTIMISTIC;
    if (const char* e = getenv("MA_SWEEP_SPECULATE")) mode = (e[0] == 'v' || e[0] == '1') ? MA_LU_SPECULATE_VERIFIED : ((e[0] == 'o' && e[1] == 'f') || e[0] == '0') ? MA_LU_SPECULATE_OFF : MA_LU_SPECULATE_OPTIMISTIC;
    int32_t cur = MA_LU_SPECULATE_OFF;
    if (ma_lu_plan_speculation(S->lu, &cur) == MA_OK && cur != MA_LU_SPECULATE_OFF && (rc = ma_lu_plan_set_speculation(S->lu, mode))) return fail(rc);
  }
  // the sweep's own stream: the plan's big-update stream when the plan splits the chip (that stream is masked to the update CUs,
  // and a stream more would be one hardware queue more: profiles/r03_lu_panel_experiments.md), a stream of its own otherwise
  { void* ms = nullptr; if (ma_lu_plan_main_stream(S->lu, &ms) == MA_OK && ms) S->st = (hipStream_t)ms; }
  if (!S->st) {
    if (hipStreamCreateWithFlags(&S->st, hipStreamNonBlocking) != hipSuccess) { set_error("sweep: stream creation failed"); return fail(MA_ERR_HIP); }
    S->own_stream = true;
  }
  S->dA.assign((size_t)slots, nullptr); S->dx.assign((size_t)slots, nullptr);
  for (int s = 0; s < slots; ++s)
    if (hipMalloc(&S->dA[(size_t)s], sizeof(ma_c64) * (size_t)n * (size_t)n) != hipSuccess || hipMalloc(&S->dx[(size_t)s], sizeof(ma_c64) * (size_t)n) != hipSuccess) {
      set_error("sweep: %d systems of %d x %d do not fit device %d", slots, n, n, device);
      return fail(MA_ERR_NOMEM);
    }
  S->staged = ma_lu_plan_num_blocks(S->lu, &S->G) == MA_OK && S->G > 0 && ma_lu_plan_stage_reset(S->lu, S->st) == MA_OK;
  if (S->staged) {
    S->spacing = std::max(1, (S->G + slots) / (slots + 1));
    (void)ma_lu_plan_stage_spacing(S->lu, slots, &S->spacing);                                   // what the plan's kernels were measured best with
  }
  // Assembly AHEAD, in PIECES, in the staged schedule. Two sets of (up to three) spare systems: while the slots consume one set --
  // a slot that begins a system swaps its matrix with the spare that holds it -- the other set's systems are assembled by
  // ma_bem_plan_assemble_multi_part_dev (their far pairs share one pass over the quadrature points) in pieces of the far pairs'
  // rows, one piece per round in the rounds just before a slot begins: there the sum of the slots' updates is smallest and the
  // stream would wait for the slots' panel chains. What has not been issued when a system is needed is issued then. Only when the
  // spares fit comfortably and when the slots take their systems in the order of the frequencies (checked per run, with the run's own
  // slot count: sweep_run): slot s begins its j-th system at round s * spacing + j * G, which is monotone in the frequency index only
  // while (slots - 1) * spacing < G -- a plan of one or two blocks (a few hundred rows) starts several slots in one round and a later
  // frequency before an earlier one; there every system is assembled when its slot begins.
  int ahead = std::min<int>(3, std::max<int>(1, max_frequencies));
  if (!S->staged || !begins_in_order(S->G, slots, S->spacing)) ahead = 1;    // ( <=> (slots - 1) spacing < blocks: tests/test_capi_cpu.py )
  {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || 2.0 * (double)ahead * 16.0 * (double)n * (double)n > 0.5 * (double)free_b) ahead = 1;
  }
  if (ahead > 1) {
    bool ok = true;
    for (auto& t : S->sets) for (int q = 0; q < ahead && ok; ++q)
      ok = hipMalloc(&t.A[q], sizeof(ma_c64) * (size_t)n * (size_t)n) == hipSuccess && hipMalloc(&t.x[q], sizeof(ma_c64) * (size_t)n) == hipSuccess;
    if (!ok) { S->free_spares(); ahead = 1; (void)hipGetLastError(); }
  }
  S->ahead = ahead;
  if (S->staged) {
    if (hipMalloc(&S->dX, sizeof(ma_c64) * (size_t)max_frequencies * (size_t)n) != hipSuccess || hipMalloc(&S->dinfo, sizeof(int32_t) * (size_t)max_frequencies) != hipSuccess) {
      set_error("sweep: the solutions of %d frequencies do not fit the device", max_frequencies);
      return fail(MA_ERR_NOMEM);
    }
  }
  if (hipEventCreate(&S->ev_run[0]) != hipSuccess || hipEventCreate(&S->ev_run[1]) != hipSuccess) { set_error("sweep: event creation failed"); return fail(MA_ERR_HIP); }
  *out = S;
  return MA_OK;
}

// frequencies first, first + stride, ... of frequencies_hz[0..n_freq) on the handle's device; X_out / status are indexed by
// the GLOBAL frequency index (X_out may be NULL: the solutions stay parked on the device, ma_bem_sweep_solutions_dev)
int sweep_run(ma_bem_sweep* S, int32_t n_freq, const double* frequencies_hz, int32_t first, int32_t stride, const SweepArgs& a, ma_c64* X_out, int32_t* status_or_null) {
  const auto wall0 = std::chrono::steady_clock::now();
  MA_HIP(hipSetDevice(S->device));
  ma_bem_plan_t* plan = S->plan; ma_lu_plan_t* lu = S->lu; hipStream_t st = S->st;
  const int32_t n = S->n;
  std::vector<int> mine;
  for (int f = first; f < n_freq; f += stride) mine.push_back(f);
  const int n_mine = (int)mine.size();
  S->have_last = false;
  if (n_mine == 0) return MA_OK;
  MA_REQUIRE(n_mine <= S->cap, MA_ERR_INVALID, "sweep: %d frequencies for a handle created for %d", n_mine, S->cap);
  MA_REQUIRE(X_out || S->staged, MA_ERR_INVALID, "sweep: without the staged schedule the solutions are not parked on the device; pass X_out");
  const int32_t slots = std::min<int32_t>(S->slots, n_mine);
  std::vector<void*>& dA = S->dA; std::vector<void*>& dx = S->dx;
  int rc = MA_OK, worst = MA_OK;
  S->ev_asm_used = 0;
  auto physics_of = [&](int f, ma_physics_t* ph, double* bim) {
    const double freq = frequencies_hz[f];
    ph->wave_number = 2.0 * 3.14159265358979323846 * freq / a.speed_of_sound;     // PhysicsParams::new, types.rs:39-58
    ph->harmonic_factor = a.harmonic_factor; ph->tau = a.tau; ph->gamma = 1.0;
    *bim = a.tau > 0.0 ? a.harmonic_factor * a.beta_scale / ph->wave_number : 0.0;   // burton_miller_beta_scaled, types.rs:144-150
  };
  auto assemble = [&](int f, int s) -> int {
    ma_physics_t ph; double bim;
    physics_of(f, &ph, &bim);
    int r = S->asm_mark();
    if (!r) r = ma_bem_plan_assemble_dev(plan, &ph, 0.0, bim, dA[(size_t)s], dx[(size_t)s], st);
    if (!r) r = ma_bem_plan_incident_rhs_dev(plan, &ph, 0.0, bim, a.incident_kind, a.incident_vec3, a.amp_re, a.amp_im, 1, dx[(size_t)s], st);
    if (!r) r = S->asm_mark();
    return r;
  };
  int32_t run_spacing = 1;
  if (S->staged) (void)ma_lu_plan_stage_spacing(lu, slots, &run_spacing);
  // (the order of the begins decides whether systems may be assembled ahead: with THIS run's slot count and spacing)
  const int ahead = (S->staged && begins_in_order(S->G, slots, run_spacing)) ? std::min(S->ahead, std::max(1, n_mine)) : 1;
  const int ppp = S->ppp;
  const int nparts = ppp * ahead;
  AsmSet* sets = S->sets;
  for (int q = 0; q < 2; ++q) { sets[q].first = -1; sets[q].cnt = 0; sets[q].part = 0; sets[q].taken = 0; }
  auto set_idle = [](const AsmSet& t) { return t.first < 0 || t.taken == (1u << t.cnt) - 1u; };
  auto start_job = [&](AsmSet& t, int first_i) {
    t.first = first_i; t.cnt = std::min(ahead, n_mine - first_i); t.part = 0; t.taken = 0;
    for (int q = 0; q < t.cnt; ++q) { physics_of(mine[(size_t)(first_i + q)], &t.ph[q], &t.bi[q]); t.br[q] = 0.0; }
  };
  auto issue_part = [&](AsmSet& t) -> int {
    int r = S->asm_mark();
    if (!r) r = ma_bem_plan_assemble_multi_part_dev(plan, t.cnt, t.ph, t.br, t.bi, t.A, t.x, t.part, nparts, st);
    if (r) return r;
    if (++t.part == nparts)
      for (int q = 0; q < t.cnt && !r; ++q)
        r = ma_bem_plan_incident_rhs_dev(plan, &t.ph[q], 0.0, t.bi[q], a.incident_kind, a.incident_vec3, a.amp_re, a.amp_im, 1, t.x[q], st);
    if (!r) r = S->asm_mark();
    return r;
  };
  // system of this device's i-th frequency into slot s
  auto take = [&](int i, int s) -> int {
    if (ahead <= 1) return assemble(mine[(size_t)i], s);
    AsmSet* t = nullptr;
    for (int c = 0; c < 2; ++c) if (sets[c].first >= 0 && i >= sets[c].first && i < sets[c].first + sets[c].cnt && !(sets[c].taken >> (i - sets[c].first) & 1u)) t = &sets[c];
    if (!t) {
      for (int c = 0; c < 2; ++c) if (!t && set_idle(sets[c])) t = &sets[c];
      if (!t) return assemble(mine[(size_t)i], s);                           // neither set holds it and neither is free: this one directly (the order of a tiny plan)
      start_job(*t, i);
      AsmSet& o = t == &sets[0] ? sets[1] : sets[0];
      if (set_idle(o) && i + t->cnt < n_mine) start_job(o, i + t->cnt);     // the set after this one: in pieces, from now on
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
    AsmSet* pick = nullptr;                                                  // the unfinished set that is needed first
    for (int c = 0; c < 2; ++c) if (sets[c].first >= 0 && sets[c].part < nparts && sets[c].taken == 0 && (!pick || sets[c].first < pick->first)) pick = &sets[c];
    return pick ? issue_part(*pick) : MA_OK;
  };
  if (S->staged) {
    const int32_t G = S->G;
    ma_c64* dX = S->dX; int32_t* dinfo = S->dinfo;
    rc = ma_lu_plan_stage_reset(lu, st);
    if (!rc && S->timing && hipEventRecord(S->ev_run[0], st) != hipSuccess) { set_error("sweep: event record failed"); rc = MA_ERR_HIP; }
    std::vector<int> off((size_t)slots);
    const int32_t spacing = run_spacing;
    for (int s = 0; s < slots; ++s) off[(size_t)s] = s * spacing;
    for (int r = 0; !rc; ++r) {
      int32_t sl[8], bl[8]; int cnt = 0; bool live = false;
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
        const int s = sl[q], i = s + slots * ((r - off[(size_t)s]) / G);
        if (bl[q] != G - 1) continue;
        rc = ma_lu_plan_stage_finish(lu, s, st);
        if (!rc && hipMemcpyAsync(dX + (size_t)i * (size_t)n, dx[(size_t)s], sizeof(ma_c64) * (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess) { set_error("sweep: parking a solution failed"); rc = MA_ERR_HIP; }
        if (!rc) rc = ma_lu_plan_stage_info_dev(lu, s, dinfo + i, st);
      }
      if (!rc) rc = assembly_tick(r, spacing);
    }
    if (!rc && S->timing && hipEventRecord(S->ev_run[1], st) != hipSuccess) { set_error("sweep: event record failed"); rc = MA_ERR_HIP; }
    if (!rc) {
      int stt = ma_lu_plan_status(lu, st);                     // synchronises; an abandoned panel (poisoned plan) surfaces here
      if (stt != MA_OK && stt != MA_ERR_SINGULAR && stt != MA_ERR_RETRY) rc = stt;
    }
    if (!rc) {
      std::vector<int32_t> hinfo((size_t)n_mine);
      hipError_t e = hipMemcpy(hinfo.data(), dinfo, sizeof(int32_t) * (size_t)n_mine, hipMemcpyDeviceToHost);
#ifdef MA_DIAGNOSTICS
      // diagnostic build only: pretend the i-th frequency of this device met a rejected panel (exercises the redo below)
      if (const char* et = getenv("MA_TEST_SWEEP_REJECT")) { const int i = atoi(et); if (i >= 0 && i < n_mine) hinfo[(size_t)i] = -1; }
#endif
      // redo: frequencies whose optimistic factorisation met a rejected panel (status -1), one at a time in the verified mode
      S->last_redone = 0;
      for (int i = 0; i < n_mine && e == hipSuccess && !rc; ++i) {
        if (hinfo[(size_t)i] >= 0) continue;
        int32_t mode = MA_LU_SPECULATE_OFF;
        rc = ma_lu_plan_speculation(lu, &mode);
        if (!rc) rc = ma_lu_plan_set_speculation(lu, MA_LU_SPECULATE_VERIFIED);
        if (!rc) rc = assemble(mine[(size_t)i], 0);
        if (!rc) rc = ma_lu_plan_factor_solve_dev(lu, dA[0], dx[0], 1, st);
        int stt = MA_OK;
        if (!rc) { stt = ma_lu_plan_status(lu, st); if (stt != MA_OK && stt != MA_ERR_SINGULAR) rc = stt; }
        if (!rc) { hinfo[(size_t)i] = stt == MA_ERR_SINGULAR ? 1 : 0; e = hipMemcpy(dX + (size_t)i * (size_t)n, dx[0], sizeof(ma_c64) * (size_t)n, hipMemcpyDeviceToDevice); }
        const int back = ma_lu_plan_set_speculation(lu, mode);
        if (!rc) rc = back;
        ++S->last_redone;
      }
      if (X_out) {
        if (stride == 1 && first == 0) { if (e == hipSuccess) e = hipMemcpy(X_out, dX, sizeof(ma_c64) * (size_t)n_mine * (size_t)n, hipMemcpyDeviceToHost); }
        else for (int i = 0; i < n_mine && e == hipSuccess; ++i)
          e = hipMemcpy(X_out + (size_t)mine[(size_t)i] * (size_t)n, dX + (size_t)i * (size_t)n, sizeof(ma_c64) * (size_t)n, hipMemcpyDeviceToHost);
      }
      if (e != hipSuccess) { set_error("sweep: copy back failed: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; }
      for (int i = 0; i < n_mine && !rc; ++i) {
        const int sf = hinfo[(size_t)i] ? MA_ERR_SINGULAR : MA_OK;
        if (status_or_null) status_or_null[mine[(size_t)i]] = sf;
        if (sf != MA_OK) worst = sf;
      }
    }
    if (rc) (void)hipDeviceSynchronize();                       // nothing may still be running on buffers that the caller may free next
    if (!rc) {
      S->last_wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
      S->last_n = n_mine; S->last_device_ms = 0.0; S->last_asm_ms = 0.0; S->last_asm_pieces = 0;
      if (S->timing) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, S->ev_run[0], S->ev_run[1]) == hipSuccess) S->last_device_ms = ms;
        for (size_t q = 0; q + 1 < S->ev_asm_used; q += 2)
          if (hipEventElapsedTime(&ms, S->ev_asm[q], S->ev_asm[q + 1]) == hipSuccess) { S->last_asm_ms += ms; ++S->last_asm_pieces; }
        (void)hipGetLastError();
      }
      S->have_last = true;
    }
    return rc ? rc : worst;
  }
  // look-ahead lanes switched off (MA_LU_LOOKAHEAD=0 / MA_LU_PANEL_OVERLAP=0): lock-step batches (always with the fallback in line)
  { int32_t mode = MA_LU_SPECULATE_OFF; if (ma_lu_plan_speculation(lu, &mode) == MA_OK && mode == MA_LU_SPECULATE_OPTIMISTIC) (void)ma_lu_plan_set_speculation(lu, MA_LU_SPECULATE_VERIFIED); }
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
  if (!rc) { S->last_wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count(); S->last_n = n_mine; S->have_last = true; }
  return rc ? rc : worst;
}

int check_args(int32_t n_freq, const double* frequencies_hz, double speed_of_sound, const double* incident_vec3) {
  MA_REQUIRE(n_freq > 0 && frequencies_hz && incident_vec3, MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(speed_of_sound > 0.0, MA_ERR_INVALID, "speed of sound must be positive");
  return MA_OK;
}

}  // namespace

extern "C" {

// the order in which a sweep of `slots` slots, `spacing` rounds apart, over plans of `blocks` blocks begins its frequencies (what decides
// whether systems may be assembled ahead: only when this is 0, 1, 2, ...); pure host arithmetic, exposed for the tests
int ma_sweep_begin_order(int32_t blocks, int32_t slots, int32_t spacing, int32_t n_freq, int32_t* order_out) {
  MA_REQUIRE(order_out || n_freq == 0, MA_ERR_INVALID, "NULL argument");
  std::vector<int32_t> o;
  const int rc = begin_order(blocks, slots, spacing, n_freq, o);
  MA_REQUIRE(rc == MA_OK && (int32_t)o.size() == n_freq, MA_ERR_INVALID, "blocks %d, slots %d, spacing %d, %d frequencies", blocks, slots, spacing, n_freq);
  for (int32_t q = 0; q < n_freq; ++q) order_out[q] = o[(size_t)q];
  return MA_OK;
}

int ma_bem_sweep_create(ma_bem_plan_t* plan, int32_t slots, int32_t max_frequencies, ma_bem_sweep_t** out) {
  return ma_bem_sweep_create_pivoting(plan, slots, max_frequencies, -1, out);
}

int ma_bem_sweep_create_pivoting(ma_bem_plan_t* plan, int32_t slots, int32_t max_frequencies, int32_t pivoting, ma_bem_sweep_t** out) {
  MA_REQUIRE(plan && out && max_frequencies > 0, MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(pivoting == -1 || pivoting == MA_LU_PIVOT_PARTIAL || pivoting == MA_LU_PIVOT_TOURNAMENT, MA_ERR_INVALID, "pivoting mode %d", pivoting);
  if (pivoting < 0) pivoting = default_pivoting();
  int prev = -1;
  const bool had = hipGetDevice(&prev) == hipSuccess;
  const int rc = sweep_create(plan, slots, max_frequencies, pivoting, out);
  if (had) (void)hipSetDevice(prev);
  return rc;
}

int ma_bem_sweep_destroy(ma_bem_sweep_t* sweep) {
  if (!sweep) return MA_OK;
  int prev = -1;
  const bool had = hipGetDevice(&prev) == hipSuccess;
  sweep->release();
  delete sweep;
  if (had) (void)hipSetDevice(prev);
  return MA_OK;
}

int ma_bem_sweep_run(ma_bem_sweep_t* sweep, int32_t n_freq, const double* frequencies_hz, double speed_of_sound, double harmonic_factor, double tau, double beta_scale,
                     int incident_kind, const double* incident_vec3, double amp_re, double amp_im, ma_c64* X_out_or_null, int32_t* status_or_null) {
  MA_REQUIRE(sweep, MA_ERR_INVALID, "NULL sweep");
  int rc = check_args(n_freq, frequencies_hz, speed_of_sound, incident_vec3);
  if (rc) return rc;
  const SweepArgs a{speed_of_sound, harmonic_factor, tau, beta_scale, incident_kind, incident_vec3, amp_re, amp_im};
  int prev = -1;
  const bool had = hipGetDevice(&prev) == hipSuccess;
  rc = sweep_run(sweep, n_freq, frequencies_hz, 0, 1, a, X_out_or_null, status_or_null);
  if (had) (void)hipSetDevice(prev);                            // the caller's current device is left as it was
  return rc;
}

int ma_bem_sweep_set_timing(ma_bem_sweep_t* sweep, int enable) {
  MA_REQUIRE(sweep, MA_ERR_INVALID, "NULL sweep");
  sweep->timing = enable != 0;
  int rc = ma_lu_plan_set_timing(sweep->lu, enable ? 2 : 0);     // 2: only the trailing-update launches are bracketed by events
  if (!rc && enable) {
    int prev = -1;
    const bool had = hipGetDevice(&prev) == hipSuccess;
    (void)hipSetDevice(sweep->device);
    rc = ma_lu_plan_reserve_events(sweep->lu, (int64_t)sweep->cap * 1400 + 64);
    while (!rc && sweep->ev_asm.size() < (size_t)sweep->cap * 16 + 16) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { set_error("sweep: event creation failed"); rc = MA_ERR_HIP; } else sweep->ev_asm.push_back(e); }
    if (had) (void)hipSetDevice(prev);
  }
  return rc;
}

int ma_bem_sweep_last_timing(ma_bem_sweep_t* sweep, double* out8) {
  MA_REQUIRE(sweep && out8, MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(sweep->have_last, MA_ERR_INVALID, "no completed run on this sweep handle");
  for (int i = 0; i < 8; ++i) out8[i] = 0.0;
  out8[0] = sweep->last_wall_s; out8[1] = sweep->last_device_ms; out8[2] = sweep->last_asm_ms; out8[3] = sweep->last_asm_pieces; out8[7] = sweep->last_n;
  if (sweep->timing && sweep->staged) {
    double t8[8];
    int rc = ma_lu_plan_last_timing(sweep->lu, t8);
    if (rc) return rc;
    out8[4] = t8[3];                                             // ms summed over the big trailing updates (events around every launch)
    double launches = 0.0, flops = 0.0;
    if ((rc = ma_lu_plan_last_big_update_stats(sweep->lu, &launches, &flops))) return rc;
    out8[5] = launches; out8[6] = flops;
  }
  return MA_OK;
}

int ma_bem_sweep_lu_plan(ma_bem_sweep_t* sweep, ma_lu_plan_t** plan) {
  MA_REQUIRE(sweep && plan, MA_ERR_INVALID, "bad argument");
  *plan = sweep->lu;
  return MA_OK;
}

int ma_bem_sweep_stream(ma_bem_sweep_t* sweep, void** stream) {
  MA_REQUIRE(sweep && stream, MA_ERR_INVALID, "bad argument");
  *stream = (void*)sweep->st;
  return MA_OK;
}

int ma_bem_sweep_info(ma_bem_sweep_t* sweep, int32_t* slots, int32_t* blocks, int32_t* spacing, int32_t* systems_ahead, int32_t* staged) {
  MA_REQUIRE(sweep, MA_ERR_INVALID, "NULL sweep");
  if (slots) *slots = sweep->slots;
  if (blocks) *blocks = sweep->G;
  if (spacing) *spacing = sweep->spacing;
  if (systems_ahead) *systems_ahead = sweep->ahead;
  if (staged) *staged = sweep->staged ? 1 : 0;
  return MA_OK;
}

int ma_bem_sweep_solutions_dev(ma_bem_sweep_t* sweep, void** d_X, int32_t* count) {
  MA_REQUIRE(sweep && d_X && count, MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(sweep->have_last && sweep->staged, MA_ERR_INVALID, "no parked solutions on this sweep handle");
  *d_X = (void*)sweep->dX; *count = sweep->last_n;
  return MA_OK;
}

// plumbing for callers that hold device pointers of the library (the parked solutions) and buffers of their own
int ma_device_copy(void* d_dst, const void* d_src, int64_t bytes, void* stream) {
  MA_REQUIRE(d_dst && d_src && bytes >= 0, MA_ERR_INVALID, "bad argument");
  MA_HIP(hipMemcpyAsync(d_dst, d_src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  MA_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MA_OK;
}

int ma_bem_solve_sweep(ma_bem_plan_t* plan, int32_t n_freq, const double* frequencies_hz, double speed_of_sound, double harmonic_factor, double tau,
                       double beta_scale, int incident_kind, const double* incident_vec3, double amp_re, double amp_im, int32_t slots,
                       ma_c64* X_out, int32_t* status_or_null) {
  MA_REQUIRE(plan && X_out, MA_ERR_INVALID, "bad argument");
  int rc = check_args(n_freq, frequencies_hz, speed_of_sound, incident_vec3);
  if (rc) return rc;
  ma_bem_sweep_t* S = nullptr;
  if ((rc = ma_bem_sweep_create(plan, slots, n_freq, &S))) return rc;
  rc = ma_bem_sweep_run(S, n_freq, frequencies_hz, speed_of_sound, harmonic_factor, tau, beta_scale, incident_kind, incident_vec3, amp_re, amp_im, X_out, status_or_null);
  ma_bem_sweep_destroy(S);
  return rc;
}

// The sharding rule of the multi-device sweep, exposed so that callers, tests and bench.py agree on it: frequency f belongs
// to device slot f mod ndev (room_simulator_bem.rs:329's loop dealt round-robin).
int ma_sweep_owner(int32_t frequency_index, int32_t ndev) { return ndev > 0 ? frequency_index % ndev : 0; }

// The multi-device loop behind a reusable handle (round 5): per device one BEM plan (geometry upload, near-pair list) and one sweep
// handle (LU plan, streams, the systems in flight, the spares, the parked solutions), made once and kept across runs -- the reference's
// driver sweeps once per source position (room_simulator_bem.rs:243-256 builds the mesh once, :328-360 loops over the frequencies), and
// ma_bem_solve_sweep_multi paid 9 hipMallocs of 1.6 GB per device per call. A run gives device d the frequencies d, d + ndev, ... on a
// host thread of its own; no collective on the data path.
struct ma_bem_sweep_multi {
  std::vector<int32_t> devices; std::vector<ma_bem_plan_t*> plans; std::vector<ma_bem_sweep*> sweeps;
  int32_t cap = 0;                                           // frequencies per run, over all devices
  std::vector<double> last_seconds; std::vector<int32_t> last_count;
};

int ma_bem_sweep_multi_destroy(ma_bem_sweep_multi_t* M) {
  if (!M) return MA_OK;
  int prev = -1;
  const bool had = hipGetDevice(&prev) == hipSuccess;
  for (size_t d = 0; d < M->sweeps.size(); ++d) if (M->sweeps[d]) { M->sweeps[d]->release(); delete M->sweeps[d]; }
  for (ma_bem_plan_t* p : M->plans) if (p) ma_bem_plan_destroy(p);
  delete M;
  if (had) (void)hipSetDevice(prev);
  return MA_OK;
}

int ma_bem_sweep_multi_create(const ma_mesh_t* mesh, const int32_t* devices, int32_t ndev, int32_t slots, int32_t max_frequencies, ma_bem_sweep_multi_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL");
  *out = nullptr;
  MA_REQUIRE(mesh && devices && ndev >= 1 && ndev <= 64 && max_frequencies > 0, MA_ERR_INVALID, "bad argument");
  int count = 0;
  int rc = ma_device_count(&count);
  if (rc) return rc;
  MA_REQUIRE(count > 0, MA_ERR_NO_DEVICE, "no gfx950 device visible");
  for (int d = 0; d < ndev; ++d) {
    MA_REQUIRE(devices[d] >= 0 && devices[d] < count, MA_ERR_INVALID, "device %d (entry %d) outside 0..%d", devices[d], d, count - 1);
#ifdef MA_DIAGNOSTICS
    // diagnostic build only (MA_TEST_ALLOW_DUPLICATE_DEVICES=1): several host threads on one GPU, so that a one-GPU box exercises the sharding
    const bool dup_ok = getenv("MA_TEST_ALLOW_DUPLICATE_DEVICES") != nullptr;
#else
    const bool dup_ok = false;
#endif
    for (int o = 0; o < d; ++o) MA_REQUIRE(devices[o] != devices[d] || dup_ok, MA_ERR_INVALID, "device %d listed twice", devices[d]);
  }
  ma_bem_sweep_multi* M = new (std::nothrow) ma_bem_sweep_multi();
  MA_REQUIRE(M, MA_ERR_NOMEM, "host allocation failed");
  M->devices.assign(devices, devices + ndev); M->plans.assign((size_t)ndev, nullptr); M->sweeps.assign((size_t)ndev, nullptr);
  M->cap = max_frequencies; M->last_seconds.assign((size_t)ndev, 0.0); M->last_count.assign((size_t)ndev, 0);
  std::vector<int> rcs((size_t)ndev, MA_OK);
  std::vector<std::string> texts((size_t)ndev);
  auto work = [&](int d) {
    const int c = (max_frequencies - d + ndev - 1) / ndev;   // the most frequencies device slot d sees in a run
    int r = c > 0 ? ma_bem_plan_create(mesh, devices[d], &M->plans[(size_t)d]) : MA_OK;
    if (!r && c > 0) r = sweep_create(M->plans[(size_t)d], slots, c, default_pivoting(), &M->sweeps[(size_t)d]);
    if (r) texts[(size_t)d] = ma_last_error_string();
    rcs[(size_t)d] = r;
  };
  int prev = -1;
  const bool had = hipGetDevice(&prev) == hipSuccess;
  std::vector<std::thread> th;
  for (int d = 1; d < ndev; ++d) th.emplace_back(work, d);
  work(0);
  for (auto& t : th) t.join();
  if (had) (void)hipSetDevice(prev);
  for (int d = 0; d < ndev; ++d)
    if (rcs[(size_t)d]) { const int code = rcs[(size_t)d]; const std::string why = texts[(size_t)d]; ma_bem_sweep_multi_destroy(M); set_error("sweep handle on device %d: %s", devices[d], why.c_str()); return code; }
  *out = M;
  return MA_OK;
}

// arguments as ma_bem_solve_sweep_multi; X_out (n_freq x num_dofs) and status_or_null are indexed by the frequency
int ma_bem_sweep_multi_run(ma_bem_sweep_multi_t* M, int32_t n_freq, const double* frequencies_hz, double speed_of_sound, double harmonic_factor, double tau, double beta_scale,
                           int incident_kind, const double* incident_vec3, double amp_re, double amp_im, ma_c64* X_out, int32_t* status_or_null) {
  MA_REQUIRE(M && X_out, MA_ERR_INVALID, "bad argument");
  int rc = check_args(n_freq, frequencies_hz, speed_of_sound, incident_vec3);
  if (rc) return rc;
  MA_REQUIRE(n_freq <= M->cap, MA_ERR_INVALID, "%d frequencies for a handle created for %d", n_freq, M->cap);
  const int ndev = (int)M->devices.size();
  const SweepArgs a{speed_of_sound, harmonic_factor, tau, beta_scale, incident_kind, incident_vec3, amp_re, amp_im};
  std::vector<int> rcs((size_t)ndev, MA_OK);
  std::vector<std::string> texts((size_t)ndev);
  auto work = [&](int d) {
    // one host thread per device; errors are thread-local and carried back as text
    int c = 0; for (int f = d; f < n_freq; f += ndev) ++c;
    const auto t1 = std::chrono::steady_clock::now();
    int r = (c > 0 && M->sweeps[(size_t)d]) ? sweep_run(M->sweeps[(size_t)d], n_freq, frequencies_hz, d, ndev, a, X_out, status_or_null) : MA_OK;
    if (r && r != MA_ERR_SINGULAR) texts[(size_t)d] = ma_last_error_string();
    rcs[(size_t)d] = r;
    M->last_seconds[(size_t)d] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    M->last_count[(size_t)d] = c;
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
    set_error("sweep on device %d: %s", M->devices[(size_t)d], texts[(size_t)d].c_str());
    return rcs[(size_t)d];
  }
  return worst;
}

// per-device account of the last run: device_seconds[d] = wall time of device d's share, device_frequencies[d] = frequencies it solved (either may be NULL)
int ma_bem_sweep_multi_last_timing(ma_bem_sweep_multi_t* M, double* device_seconds, int32_t* device_frequencies) {
  MA_REQUIRE(M, MA_ERR_INVALID, "NULL handle");
  for (size_t d = 0; d < M->devices.size(); ++d) { if (device_seconds) device_seconds[d] = M->last_seconds[d]; if (device_frequencies) device_frequencies[d] = M->last_count[d]; }
  return MA_OK;
}

// create + run + destroy, with a per-device account for the caller: device_seconds[d] = wall time of device d's sweep run, device_setup_seconds[d]
// = the handle's creation (all devices in parallel: the same figure for each), device_frequencies[d] = frequencies it solved; any may be NULL
int ma_bem_solve_sweep_multi_timed(const ma_mesh_t* mesh, const int32_t* devices, int32_t ndev, int32_t n_freq, const double* frequencies_hz, double speed_of_sound,
                                   double harmonic_factor, double tau, double beta_scale, int incident_kind, const double* incident_vec3, double amp_re, double amp_im,
                                   int32_t slots, ma_c64* X_out, int32_t* status_or_null, double* device_seconds, double* device_setup_seconds, int32_t* device_frequencies) {
  MA_REQUIRE(mesh && devices && ndev >= 1 && ndev <= 64 && X_out, MA_ERR_INVALID, "bad argument");
  int rc = check_args(n_freq, frequencies_hz, speed_of_sound, incident_vec3);
  if (rc) return rc;
  const auto t0 = std::chrono::steady_clock::now();
  ma_bem_sweep_multi_t* M = nullptr;
  if ((rc = ma_bem_sweep_multi_create(mesh, devices, ndev, slots, n_freq, &M))) return rc;
  const double setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  rc = ma_bem_sweep_multi_run(M, n_freq, frequencies_hz, speed_of_sound, harmonic_factor, tau, beta_scale, incident_kind, incident_vec3, amp_re, amp_im, X_out, status_or_null);
  (void)ma_bem_sweep_multi_last_timing(M, device_seconds, device_frequencies);
  if (device_setup_seconds) for (int d = 0; d < ndev; ++d) device_setup_seconds[d] = setup;
  ma_bem_sweep_multi_destroy(M);
  return rc;
}

int ma_bem_solve_sweep_multi(const ma_mesh_t* mesh, const int32_t* devices, int32_t ndev, int32_t n_freq, const double* frequencies_hz, double speed_of_sound,
                             double harmonic_factor, double tau, double beta_scale, int incident_kind, const double* incident_vec3, double amp_re, double amp_im,
                             int32_t slots, ma_c64* X_out, int32_t* status_or_null) {
  return ma_bem_solve_sweep_multi_timed(mesh, devices, ndev, n_freq, frequencies_hz, speed_of_sound, harmonic_factor, tau, beta_scale, incident_kind, incident_vec3,
                                        amp_re, amp_im, slots, X_out, status_or_null, nullptr, nullptr, nullptr);
}

}  // extern "C"
