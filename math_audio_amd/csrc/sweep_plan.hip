// sweep_plan.hip — the frequency loop of the BEM drivers as one device-resident call
// (math-bem/bin/room_simulator_bem.rs:329-360 and BemSolver::solve, bem_solver.rs:355-480: per frequency
// PhysicsParams::new, beta = burton_miller_beta_scaled, build_tbem_system_with_beta, compute_rhs_with_beta, lu_solve).
// Built on the public entry points only: the systems go through the staged plan API as a pipeline -- `slots` of them in
// HBM at a time, slot s a quarter of a factorisation behind slot s-1 (see ma_lu_plan_stage_*) -- without a host
// synchronisation inside; the solutions are parked on the device and travel back once at the end.
#include "ma_common.hpp"
#include <vector>
#include <algorithm>
#include <cmath>

using namespace ma;

extern "C" {

int ma_bem_solve_sweep(ma_bem_plan_t* plan, int32_t n_freq, const double* frequencies_hz, double speed_of_sound, double harmonic_factor, double tau,
                       double beta_scale, int incident_kind, const double* incident_vec3, double amp_re, double amp_im, int32_t slots,
                       ma_c64* X_out, int32_t* status_or_null) {
  MA_REQUIRE(plan && n_freq > 0 && frequencies_hz && incident_vec3 && X_out, MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(speed_of_sound > 0.0, MA_ERR_INVALID, "speed of sound must be positive");
  int32_t n = 0;
  int rc = ma_bem_plan_num_dofs(plan, &n);
  if (rc) return rc;
  if (slots < 1) slots = 3;
  if (slots > 4) slots = 4;
  if (slots > n_freq) slots = n_freq;
  int device = 0;
  MA_HIP(hipGetDevice(&device));
  ma_lu_plan_t* lu = nullptr;
  if ((rc = ma_lu_plan_create(n, device, &lu))) return rc;
  std::vector<void*> dA((size_t)slots, nullptr), dx((size_t)slots, nullptr);
  auto cleanup = [&]() { for (void* p : dA) if (p) (void)hipFree(p); for (void* p : dx) if (p) (void)hipFree(p); ma_lu_plan_destroy(lu); };
  for (int s = 0; s < slots; ++s)
    if (hipMalloc(&dA[(size_t)s], sizeof(ma_c64) * (size_t)n * (size_t)n) != hipSuccess || hipMalloc(&dx[(size_t)s], sizeof(ma_c64) * (size_t)n) != hipSuccess) {
      set_error("sweep: %d systems of %d x %d do not fit the device", slots, n, n);
      cleanup();
      return MA_ERR_NOMEM;
    }
  int worst = MA_OK;
  auto assemble = [&](int f, int s) -> int {
    const double freq = frequencies_hz[f];
    ma_physics_t ph;
    ph.wave_number = 2.0 * 3.14159265358979323846 * freq / speed_of_sound;     // PhysicsParams::new, types.rs:39-58
    ph.harmonic_factor = harmonic_factor; ph.tau = tau; ph.gamma = 1.0;
    const double bim = tau > 0.0 ? harmonic_factor * beta_scale / ph.wave_number : 0.0;   // burton_miller_beta_scaled, types.rs:144-150
    int r = ma_bem_plan_assemble_dev(plan, &ph, 0.0, bim, dA[(size_t)s], dx[(size_t)s], nullptr);
    if (!r) r = ma_bem_plan_incident_rhs_dev(plan, &ph, 0.0, bim, incident_kind, incident_vec3, amp_re, amp_im, 1, dx[(size_t)s], nullptr);
    return r;
  };
  int32_t G = 0;
  ma_c64* dX = nullptr; int32_t* dinfo = nullptr;
  const bool staged = ma_lu_plan_num_blocks(lu, &G) == MA_OK && G > 0 && ma_lu_plan_stage_reset(lu, nullptr) == MA_OK;
  if (staged) {
    if (hipMalloc(&dX, sizeof(ma_c64) * (size_t)n_freq * (size_t)n) != hipSuccess || hipMalloc(&dinfo, sizeof(int32_t) * (size_t)n_freq) != hipSuccess) {
      if (dX) (void)hipFree(dX);
      set_error("sweep: the solutions of %d frequencies do not fit the device", n_freq);
      cleanup();
      return MA_ERR_NOMEM;
    }
    std::vector<int> off((size_t)slots);
    for (int s = 0; s < slots; ++s) off[(size_t)s] = s * std::max(1, (G + slots) / (slots + 1));   // G/4 apart for 3 slots (measured best)
    for (int r = 0; !rc; ++r) {
      int32_t sl[4], bl[4]; int cnt = 0; bool live = false;
      for (int s = 0; s < slots && !rc; ++s) {
        const int lr = r - off[(size_t)s];
        if (lr < 0) { live = true; continue; }
        const int f = s + slots * (lr / G), g = lr % G;
        if (f >= n_freq) continue;
        live = true;
        if (g == 0) {
          rc = assemble(f, s);
          if (!rc) rc = ma_lu_plan_stage_begin(lu, s, dA[(size_t)s], dx[(size_t)s], 1, nullptr);
        }
        sl[cnt] = s; bl[cnt] = g; ++cnt;
      }
      if (!live || rc) break;
      if (cnt) rc = ma_lu_plan_stage_round(lu, cnt, sl, bl, nullptr);
      for (int i = 0; i < cnt && !rc; ++i) {
        if (bl[i] != G - 1) continue;
        const int s = sl[i], f = s + slots * ((r - off[(size_t)s]) / G);
        rc = ma_lu_plan_stage_finish(lu, s, nullptr);
        if (!rc && hipMemcpyAsync(dX + (size_t)f * (size_t)n, dx[(size_t)s], sizeof(ma_c64) * (size_t)n, hipMemcpyDeviceToDevice, nullptr) != hipSuccess) { set_error("sweep: parking a solution failed"); rc = MA_ERR_HIP; }
        if (!rc) rc = ma_lu_plan_stage_info_dev(lu, s, dinfo + f, nullptr);
      }
    }
    if (!rc) {
      int st = ma_lu_plan_status(lu, nullptr);                 // synchronises; a time-out of the panel kernels surfaces here
      if (st != MA_OK && st != MA_ERR_SINGULAR) rc = st;
    }
    if (!rc) {
      std::vector<int32_t> hinfo((size_t)n_freq);
      if (hipMemcpy(X_out, dX, sizeof(ma_c64) * (size_t)n_freq * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(hinfo.data(), dinfo, sizeof(int32_t) * (size_t)n_freq, hipMemcpyDeviceToHost) != hipSuccess) { set_error("sweep: copy back failed"); rc = MA_ERR_HIP; }
      for (int f = 0; f < n_freq && !rc; ++f) {
        const int st = hinfo[(size_t)f] ? MA_ERR_SINGULAR : MA_OK;
        if (status_or_null) status_or_null[f] = st;
        if (st != MA_OK) worst = st;
      }
    }
    if (rc) (void)hipDeviceSynchronize();                       // nothing may still be running on buffers that are about to go
    (void)hipFree(dX); (void)hipFree(dinfo);
    cleanup();
    return rc ? rc : worst;
  }
  // look-ahead lanes switched off (MA_LU_LOOKAHEAD=0 / MA_LU_PANEL_OVERLAP=0): lock-step batches
  for (int f0 = 0; f0 < n_freq && !rc; f0 += slots) {
    const int cnt = std::min(slots, n_freq - f0);
    for (int s = 0; s < cnt && !rc; ++s) rc = assemble(f0 + s, s);
    if (rc) break;
    rc = ma_lu_plan_factor_solve_batch_dev(lu, cnt, dA.data(), dx.data(), 1, nullptr);
    if (rc) break;
    int st = ma_lu_plan_status(lu, nullptr);
    if (st != MA_OK && st != MA_ERR_SINGULAR) { rc = st; break; }
    for (int s = 0; s < cnt; ++s) {
      if (status_or_null) status_or_null[f0 + s] = st;       // a singular member marks its whole batch; the caller may re-run those singly
      if (hipMemcpy(X_out + (size_t)(f0 + s) * (size_t)n, dx[(size_t)s], sizeof(ma_c64) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) { set_error("sweep: copy back failed"); rc = MA_ERR_HIP; }
    }
    if (st != MA_OK) worst = st;
  }
  cleanup();
  return rc ? rc : worst;
}

}  // extern "C"
