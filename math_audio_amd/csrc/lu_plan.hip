// lu_plan.hip — host driver of the blocked right-looking LU and the C-ABI of the dense solve
// (replaces lu_solve, math-solvers/src/direct/lu.rs:142-153).
//
// One schedule (rounds 1-4 kept several; the ones that lost are in profiles/r0*_lu_*.md, not here): 64-column panels, each factored as
// two 32-column half-panels; kb panels (default 6: K = 384) per trailing update; a look-ahead lane per system that factors the next
// block's panels beside the current block's big update. A half-panel is factored
//   - speculatively first (lu_spec.hip: partial pivoting inside the panel's top rows, verified against every row below; widened
//     attempts with the rows the check turned up) -- accepted, that IS zgetrf's panel --, and where all attempts give up
//   - by the plan's own panel kernel: lu_panel_reg_kernel (partial pivoting, co-resident workgroups that exchange once per column;
//     MA_LU_PIVOT_PARTIAL, the mode of every entry that hands pivots or factors across this boundary) or lu_calu_panel_kernel
//     (tournament pivoting, no workgroup waits for another; MA_LU_PIVOT_TOURNAMENT, the frequency sweep's mode).
#include "lu_kernels.hpp"
#include <vector>
#include <cstring>
#include <algorithm>
#include <new>

using namespace ma;

#define LU_LISTS_LEN (1 + 4 * LU_NB_MAX)
// A partial-pivoting plan of MA_LU_SPLIT_MIN_N..MA_LU_SPLIT_MAX_N rows on a 256-CU chip keeps its big trailing updates off 64 CUs (the
// sizes measured: 6 000 - 14 000 rows, profiles/r03_lu_panel_experiments.md): its spinning panel kernels find those free of update
// workgroups. A tournament plan (whose panels do not spin) keeps them off 32. The mask is worth 0, 32 or 64 CUs only: 32 bits of it
// take ONE CU out of every shader engine of every XCD, and the dispatcher runs every engine at the pace of the smallest
// (profiles/r05_lu_panel_experiments.md (b)).
#ifndef MA_LU_SPLIT_MIN_N
#define MA_LU_SPLIT_MIN_N 4096
#endif
#ifndef MA_LU_SPLIT_MAX_N
#define MA_LU_SPLIT_MAX_N 16384
#endif
#define MA_LU_CU_SPLIT_PARTIAL 64
#define MA_LU_CU_SPLIT_TOURNAMENT 32
#define LU_PANEL 64                         // columns of a panel (two half-panels of LU_REG_NB)
#define LU_KB_MAX 8                         // panels per trailing update
#define LU_KB_DEFAULT 6                     // K = 384: the update kernel's prologue and C read-modify-write are a fifth of a K = 256 tile's time (r03 (f))
#define LU_LANE_TSTRIDE 512                 // the lane's interchanges touch at most (kb - 1) panels' columns: 7 x 64

struct ma_lu_plan {
  int device = 0;
  int n = 0;
  int ncu = 256;
  void* ws_block = nullptr;       // one allocation: sync words | info | cand | candrow | diagrow of the spinning panel kernel, per system
  LuPanelWs pws{};                // system 0's panel workspace; pws_m[m] for the other systems of a batch (own gather buffers,
  LuPanelWs pws_m[LU_BATCH_MAX]{};  // so that two systems' panel kernels may be in flight together when the chip holds both)
  // per system of a batch: pivots, the folded interchange lists, and 2*NB rows x (n + nrhs_max) staging for the interchanges
  int* d_lists[LU_BATCH_MAX] = {};
  int* d_ipiv[LU_BATCH_MAX] = {};
  c64* d_tmp[LU_BATCH_MAX] = {};
  c64* d_invd[LU_BATCH_MAX] = {};  // inverted 32 x 32 diagonal blocks of L11, one slot per panel of a block; lists and these are written by the
                                   // look-ahead lane (block g+1 -> slots of parity (g+1)&1) and read by the main lane (block g)
  int* d_half_lists[LU_BATCH_MAX] = {}; c64* d_half_invd[LU_BATCH_MAX] = {}; c64* d_half_l10[LU_BATCH_MAX] = {};   // the two half-panels of the panel being factored
  int kb = LU_KB_DEFAULT;         // panels per trailing update (MA_LU_KB=1..8)
  double gemm_flops = 0.0;        // algorithmic flops of the update launches of the last call
  double gemm_cbytes = 0.0;       // and their algorithmic C read + write bytes
  int last_batch = 1;
  // staged (pipelined) use: the system currently in each slot, and the event that says its A/b are ready
  c64* cur_A[LU_BATCH_MAX] = {}; c64* cur_B[LU_BATCH_MAX] = {}; int cur_nrhs = 0;
  hipEvent_t ev_prep[LU_BATCH_MAX] = {};
  int stage_first_mark = -1;
  hipEvent_t ev_start = nullptr, ev_panel[LU_BATCH_MAX] = {}, ev_narrow[LU_BATCH_MAX] = {}, ev_mid[LU_BATCH_MAX] = {}, ev_big[LU_BATCH_MAX] = {};
  int ensure_batch(int nmat);
  int nrhs_max = 4;
  bool timing = false;
  bool timing_detail = true;
  std::vector<hipEvent_t> ev;     // event pool for per-phase timing
  size_t ev_used = 0;
  struct Iv { int a, b, phase; };
  std::vector<Iv> iv;             // timed intervals of the last call
  int ev_last = -1;
  int n_gemm_launch = 0;
  int n_big_launch = 0; double big_flops = 0.0;   // of those: the big trailing updates on the caller's stream (phase 3)
  bool ev_valid = false;
  hipStream_t panel_streams[LU_BATCH_MAX] = {};   // the look-ahead lanes, one per system of a batch / slot of the staged schedule
  bool zgemm_dma = true;          // MA_ZGEMM_DMA=0: the register-staged update kernel (the LDS-DMA one is bit-identical and faster; tests hold the identity)
  int cu_split = 0;               // CUs the big updates stay off (MA_LU_CU_SPLIT=<0 | 32 | 64>); they then run on big_stream, masked to the others
  hipStream_t big_stream = nullptr;
  int pivoting = MA_LU_PIVOT_PARTIAL;
  LuCaluWs calu[LU_BATCH_MAX]{};
  bool speculate = true;          // the speculative panel ahead of every half-panel (MA_LU_SPECULATE=0 / ma_lu_plan_set_speculation switch it off)
  bool optimistic = false;        // MA_LU_SPECULATE_OPTIMISTIC: nothing is launched behind the speculative panels; a system that met one all attempts
                                  // give up carries -1 in its status word (MA_ERR_RETRY) and the CALLER solves it again in the verified mode
  LuSpecWs spec[LU_BATCH_MAX]{};
  unsigned long long* d_spec_stats = nullptr;
};

int ma_lu_plan::ensure_batch(int nmat) {
  for (int m = 0; m < nmat; ++m) {
    if (d_tmp[m]) continue;
    MA_HIP(hipMalloc(&d_tmp[m], sizeof(c64) * 2 * LU_NB_MAX * ((size_t)n + nrhs_max)));
    MA_HIP(hipMalloc(&d_ipiv[m], sizeof(int) * (size_t)n));
    MA_HIP(hipMemset(d_ipiv[m], 0, sizeof(int) * (size_t)n));
    MA_HIP(hipMalloc(&d_lists[m], sizeof(int) * 2 * LU_KB_MAX * LU_LISTS_LEN));
    MA_HIP(hipMalloc(&d_invd[m], sizeof(c64) * 2 * LU_KB_MAX * LU_NB_MAX * 32));
    MA_HIP(hipMalloc(&d_half_lists[m], sizeof(int) * 2 * LU_LISTS_LEN));
    MA_HIP(hipMemset(d_half_lists[m], 0, sizeof(int) * 2 * LU_LISTS_LEN));
    MA_HIP(hipMalloc(&d_half_invd[m], sizeof(c64) * 32 * 32));
    MA_HIP(hipMalloc(&d_half_l10[m], sizeof(c64) * 32 * 32));
    MA_HIP(hipMemset(d_half_l10[m], 0, sizeof(c64) * 32 * 32));
    {
      LuSpecWs& w = spec[m];
      MA_HIP(hipMalloc(&w.u11, sizeof(c64) * LU_REG_NB * LU_REG_NB));
      MA_HIP(hipMalloc(&w.rinv, sizeof(c64) * LU_REG_NB));
      MA_HIP(hipMalloc(&w.pivmag, sizeof(double) * LU_REG_NB));
      MA_HIP(hipMalloc(&w.ctl, 64));
      MA_HIP(hipMemset(w.ctl, 0, 64));
      MA_HIP(hipMalloc(&w.vlist, sizeof(int) * LU_REG_NB));
      MA_HIP(hipMalloc(&w.ext, sizeof(int) * 2 * LU_REG_NB));
      MA_HIP(hipMalloc(&w.backup, sizeof(c64) * (size_t)n * LU_REG_NB));
      w.rows = n; w.stats = d_spec_stats;
    }
    if (pivoting == MA_LU_PIVOT_TOURNAMENT) {
      const int nodes = lu_calu_tree_nodes((n + 255) / 256);
      MA_HIP(hipMalloc(&calu[m].cand, sizeof(int) * (size_t)nodes * LU_REG_NB));
      MA_HIP(hipMalloc(&calu[m].counters, sizeof(unsigned) * (size_t)nodes));
      MA_HIP(hipMemset(calu[m].counters, 0, sizeof(unsigned) * (size_t)nodes));
      calu[m].max_nodes = nodes;
    }
    // the memsets above run on the null stream; the plan's lanes and the callers' streams may be non-blocking streams that do
    // not order themselves against it: without this wait one can land AFTER a panel kernel has written its pivots (seen as
    // "pivot outside its range" under two host threads)
    MA_HIP(hipStreamSynchronize(nullptr));
  }
  return MA_OK;
}

extern "C" {

int ma_lu_plan_create(int32_t n, int device, ma_lu_plan_t** out) {
  return ma_lu_plan_create_pivoting(n, device, MA_LU_PIVOT_PARTIAL, out);          // the drop-in entries of lu.rs: LAPACK's pivoting
}

int ma_lu_plan_pivoting(ma_lu_plan_t* P, int32_t* pivoting) {
  MA_REQUIRE(P && pivoting, MA_ERR_INVALID, "NULL argument");
  *pivoting = P->pivoting;
  return MA_OK;
}

int ma_lu_plan_create_pivoting(int32_t n, int device, int32_t pivoting, ma_lu_plan_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL");
  *out = nullptr;
  MA_REQUIRE(n > 0, MA_ERR_DIM, "n must be positive (got %d)", n);
  MA_REQUIRE(pivoting == MA_LU_PIVOT_PARTIAL || pivoting == MA_LU_PIVOT_TOURNAMENT, MA_ERR_INVALID, "pivoting mode %d", pivoting);
  int rc = use_device(device);
  if (rc) return rc;
  hipDeviceProp_t prop;
  MA_HIP(hipGetDeviceProperties(&prop, device));
  const int ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  MA_REQUIRE((long long)n <= 256LL * ncu && n <= 65535, MA_ERR_UNSUPPORTED, "n = %d exceeds the panel kernels' capacity (%d rows)", n, std::min(65535, 256 * ncu));
  ma_lu_plan* P = new (std::nothrow) ma_lu_plan();
  MA_REQUIRE(P, MA_ERR_NOMEM, "host allocation failed");
  P->device = device; P->n = n; P->ncu = ncu; P->pivoting = pivoting;
  if (const char* es = getenv("MA_LU_SPECULATE")) P->speculate = atoi(es) != 0;
  if (hipMalloc(&P->d_spec_stats, 64) != hipSuccess || hipMemset(P->d_spec_stats, 0, 64) != hipSuccess) { set_error("hipMalloc of the LU plan's counters failed"); delete P; return MA_ERR_NOMEM; }
  const int mb = ncu;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_sync = take(16), o_info = take(64);
  size_t o_cand[LU_BATCH_MAX], o_crow[LU_BATCH_MAX], o_drow[LU_BATCH_MAX];
  for (int m = 0; m < LU_BATCH_MAX; ++m) {
    o_cand[m] = take(lu_panel_granule_bytes(mb));
    o_crow[m] = take(sizeof(unsigned long long) * 2 * (size_t)mb * 2 * LU_NB_MAX);
    o_drow[m] = take(sizeof(unsigned long long) * (2 * 2 * LU_NB_MAX + 16));
  }
  hipError_t e = hipMalloc(&P->ws_block, off);
  if (e != hipSuccess) {
    set_error("hipMalloc of the LU workspace failed: %s", hipGetErrorString(e));
    (void)hipFree(P->d_spec_stats);
    delete P;
    return MA_ERR_NOMEM;
  }
  if ((rc = P->ensure_batch(1))) { ma_lu_plan_destroy(P); return rc; }
  char* base = (char*)P->ws_block;
  P->pws.counter = (unsigned*)(base + o_sync);
  P->pws.info = (int*)(base + o_info);
  P->pws.timeout = (unsigned*)(P->pws.info + LU_BATCH_MAX);   // persists over the factorisation, like info
  P->pws.max_blocks = mb;
  P->pws.test_abort_col = -1;
  for (int m = 0; m < LU_BATCH_MAX; ++m) {
    P->pws_m[m] = P->pws;
    P->pws_m[m].info = P->pws.info + m;                       // per-system first-zero-pivot word
    P->pws_m[m].cand = (unsigned long long*)(base + o_cand[m]);
    P->pws_m[m].candrow = (unsigned long long*)(base + o_crow[m]);
    P->pws_m[m].diagrow = (unsigned long long*)(base + o_drow[m]);
  }
#ifdef MA_DIAGNOSTICS
  // diagnostic build only: the last workgroup of the spinning panel kernel gives up at this global column (tests of the abandoned-wait path)
  if (const char* e9 = getenv("MA_LU_TEST_ABORT_COL")) for (int m = 0; m < LU_BATCH_MAX; ++m) P->pws_m[m].test_abort_col = atoi(e9);
#endif
  P->pws = P->pws_m[0];
  rc = lu_trsm_configure();
  if (const char* e0 = getenv("MA_ZGEMM_DMA")) P->zgemm_dma = atoi(e0) != 0;
  if (const char* e6 = getenv("MA_LU_KB")) { int v = atoi(e6); if (v >= 1 && v <= LU_KB_MAX) P->kb = v; }
  // the spinning kernel's tallest grid (256 rows per workgroup) must be co-resident on the chip on its own
  if (!rc && pivoting == MA_LU_PIVOT_PARTIAL) rc = lu_panel_reg_admissible((n + 255) / 256, ncu);
  {
    const bool tour = pivoting == MA_LU_PIVOT_TOURNAMENT;
    int split = (n >= MA_LU_SPLIT_MIN_N && n <= MA_LU_SPLIT_MAX_N && ncu == 256) ? (tour ? MA_LU_CU_SPLIT_TOURNAMENT : MA_LU_CU_SPLIT_PARTIAL) : 0;
    if (const char* es = getenv("MA_LU_CU_SPLIT")) split = atoi(es);
    if (split < 8 || split % 8 != 0 || split > ncu - 64 || ncu % 32 != 0) split = 0;
    P->cu_split = split;
  }
  if (!rc) {
    hipError_t e4 = hipSuccess;
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipStreamCreateWithFlags(&P->panel_streams[i], hipStreamNonBlocking);
    if (e4 == hipSuccess) e4 = hipEventCreateWithFlags(&P->ev_start, hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_panel[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_narrow[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_mid[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_big[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_prep[i], hipEventDisableTiming);
    if (e4 == hipSuccess && P->cu_split) {
      // CU masks: bit i of the mask is XCD i mod 8, then shader engine (i / 8) mod 4, then the CU inside it (tools/cumask_probe.hip): bits
      // [0, P) are P / 8 CUs of every XCD. Masked streams are blocking streams (the only kind hipExtStreamCreateWithCUMask makes): a caller
      // that drives the staged schedule from the NULL stream serialises against them -- use a non-blocking stream (ma_bem_sweep_run does)
      const int words = ncu / 32;
      std::vector<uint32_t> mB(words, 0u);
      for (int i = P->cu_split; i < ncu; ++i) mB[i / 32] |= 1u << (i % 32);
      e4 = hipExtStreamCreateWithCUMask(&P->big_stream, words, mB.data());
      if (e4 != hipSuccess && !getenv("MA_LU_CU_SPLIT")) {     // the default split on a runtime that makes no masked streams: the whole chip for everything
        (void)hipGetLastError(); e4 = hipSuccess; P->big_stream = nullptr; P->cu_split = 0;
      }
      // the mask's bit layout is what tools/cumask_probe.hip found on an SPX MI355X; a census on the new stream says whether THIS
      // device agrees: ncu - split CUs in use, the same number in every XCD. If not (another partition mode, another part), the plan
      // runs its updates on the whole chip -- slower, never wrong (ma_lu_plan_cu_split then reports 0).
      if (e4 == hipSuccess && P->big_stream) {
        bool mask_ok = false;
        rc = lu_cumask_selfcheck(P->big_stream, ncu - P->cu_split, &mask_ok);
        if (!rc && !mask_ok) { (void)hipStreamDestroy(P->big_stream); P->big_stream = nullptr; P->cu_split = 0; }
      }
    }
    if (e4 != hipSuccess) { set_error("stream/event creation failed: %s", hipGetErrorString(e4)); rc = MA_ERR_HIP; }
  }
  if (rc) { ma_lu_plan_destroy(P); return rc; }
  *out = P;
  return MA_OK;
}

int ma_lu_plan_destroy(ma_lu_plan_t* P) {
  if (!P) return MA_OK;
  (void)hipSetDevice(P->device);
  // nothing of this plan may still be running when its streams and workspaces go (the panel sequencer keeps events that
  // were recorded on these streams)
  for (int i = 0; i < LU_BATCH_MAX; ++i) if (P->panel_streams[i]) (void)hipStreamSynchronize(P->panel_streams[i]);
  if (P->big_stream) (void)hipStreamSynchronize(P->big_stream);
  for (int i = 0; i < LU_BATCH_MAX; ++i) lu_panel_forget_stream(P->device, P->panel_streams[i]);
  for (hipEvent_t e : P->ev) (void)hipEventDestroy(e);
  if (P->ev_start) (void)hipEventDestroy(P->ev_start);
  for (int i = 0; i < LU_BATCH_MAX; ++i) {
    if (P->ev_panel[i]) (void)hipEventDestroy(P->ev_panel[i]); if (P->ev_narrow[i]) (void)hipEventDestroy(P->ev_narrow[i]);
    if (P->ev_prep[i]) (void)hipEventDestroy(P->ev_prep[i]); if (P->ev_mid[i]) (void)hipEventDestroy(P->ev_mid[i]); if (P->ev_big[i]) (void)hipEventDestroy(P->ev_big[i]);
    if (P->panel_streams[i]) (void)hipStreamDestroy(P->panel_streams[i]);
  }
  if (P->big_stream) (void)hipStreamDestroy(P->big_stream);
  for (int i = 0; i < LU_BATCH_MAX; ++i) {
    if (P->d_tmp[i]) (void)hipFree(P->d_tmp[i]); if (P->d_ipiv[i]) (void)hipFree(P->d_ipiv[i]); if (P->d_lists[i]) (void)hipFree(P->d_lists[i]); if (P->d_invd[i]) (void)hipFree(P->d_invd[i]);
    if (P->calu[i].cand) (void)hipFree(P->calu[i].cand); if (P->calu[i].counters) (void)hipFree(P->calu[i].counters);
    { LuSpecWs& w = P->spec[i]; if (w.u11) (void)hipFree(w.u11); if (w.rinv) (void)hipFree(w.rinv); if (w.pivmag) (void)hipFree(w.pivmag); if (w.ctl) (void)hipFree(w.ctl);
      if (w.vlist) (void)hipFree(w.vlist); if (w.ext) (void)hipFree(w.ext); if (w.backup) (void)hipFree(w.backup); }
    if (P->d_half_lists[i]) (void)hipFree(P->d_half_lists[i]); if (P->d_half_invd[i]) (void)hipFree(P->d_half_invd[i]); if (P->d_half_l10[i]) (void)hipFree(P->d_half_l10[i]);
  }
  if (P->ws_block) (void)hipFree(P->ws_block);
  if (P->d_spec_stats) (void)hipFree(P->d_spec_stats);
  delete P;
  return MA_OK;
}

// 0: no speculation; 1: verified, the plan's own panel kernel in line behind every speculative panel (what a plan starts with); 2: optimistic
int ma_lu_plan_set_speculation(ma_lu_plan_t* P, int32_t mode) {
  MA_REQUIRE(P && mode >= MA_LU_SPECULATE_OFF && mode <= MA_LU_SPECULATE_OPTIMISTIC, MA_ERR_INVALID, "speculation mode %d", mode);
  P->speculate = mode != MA_LU_SPECULATE_OFF;
  P->optimistic = mode == MA_LU_SPECULATE_OPTIMISTIC;
  return MA_OK;
}
int ma_lu_plan_speculation(ma_lu_plan_t* P, int32_t* mode) {
  MA_REQUIRE(P && mode, MA_ERR_INVALID, "NULL argument");
  *mode = !P->speculate ? MA_LU_SPECULATE_OFF : (P->optimistic ? MA_LU_SPECULATE_OPTIMISTIC : MA_LU_SPECULATE_VERIFIED);
  return MA_OK;
}

// half-panels the speculative panel factored at the first attempt / at a widened attempt / handed to the plan's own panel kernel or
// marked for another solve, since the plan was made; synchronises the device
int ma_lu_plan_speculation_stats(ma_lu_plan_t* P, int64_t* accepted, int64_t* accepted_widened, int64_t* rejected) {
  MA_REQUIRE(P && accepted && accepted_widened && rejected, MA_ERR_INVALID, "NULL argument");
  *accepted = 0; *accepted_widened = 0; *rejected = 0;
  MA_HIP(hipSetDevice(P->device));
  MA_HIP(hipDeviceSynchronize());
  unsigned long long h[3] = {0, 0, 0};
  MA_HIP(hipMemcpy(h, P->d_spec_stats, sizeof(h), hipMemcpyDeviceToHost));
  *accepted = (int64_t)(h[0] - h[1]); *accepted_widened = (int64_t)h[2]; *rejected = (int64_t)(h[1] - h[2]);
  return MA_OK;
}

// create timing events ahead of a timed run (the pool otherwise grows inside it: a pipelined sweep of K systems records
// about 1700 K events)
int ma_lu_plan_reserve_events(ma_lu_plan_t* P, int64_t count) {
  MA_REQUIRE(P && count >= 0, MA_ERR_INVALID, "bad argument");
  MA_HIP(hipSetDevice(P->device));
  while ((int64_t)P->ev.size() < count) { hipEvent_t e; MA_HIP(hipEventCreate(&e)); P->ev.push_back(e); }
  return MA_OK;
}

int ma_lu_plan_set_timing(ma_lu_plan_t* P, int enable) {
  MA_REQUIRE(P, MA_ERR_INVALID, "NULL plan");
  P->timing = enable != 0;
  P->timing_detail = enable != 2;    // 2: only the trailing-update launches are bracketed (fewer events on the latency-bound chains)
  P->ev_valid = false;
  return MA_OK;
}

}  // extern "C"

// timing bookkeeping: events are taken from a pool; an interval is (begin event, end event, phase)
static int mark(ma_lu_plan* P, hipStream_t st, int* idx_out, bool detail = false) {
  *idx_out = -1;
  if (!P->timing || (detail && !P->timing_detail)) return MA_OK;
  if (P->ev_used >= P->ev.size()) { hipEvent_t e; MA_HIP(hipEventCreate(&e)); P->ev.push_back(e); }
  MA_HIP(hipEventRecord(P->ev[P->ev_used], st));
  *idx_out = (int)P->ev_used++;
  return MA_OK;
}
static void interval(ma_lu_plan* P, int a, int b, int phase) {
  if (a >= 0 && b >= 0) P->iv.push_back({a, b, phase});
}
#define MA_MARK(var, stream) int var; if ((rc = mark(P, (stream), &var))) return rc
#define MA_MARKD(var, stream) int var; if ((rc = mark(P, (stream), &var, true))) return rc   /* phases other than the trailing updates */

namespace {

// The schedule of one factorisation and its pieces, shared by the lock-step batch (factor_solve_batch) and the staged pipeline (the
// ma_lu_plan_stage_* entries):
//
//   look-ahead lane (own stream per system), block g+1 = panels p_0..p_{kb-1}, columns [a_0, e):
//     for j: panel(p_j) as two half-panels;  interchanges of p_j -> columns [a_{j+1}, e);  U = L_jj^-1 A[p_j rows, a_{j+1}:e);
//            A[a_{j+1}:n, a_{j+1}:e) -= L[a_{j+1}:n, p_j] U          (a small right-looking LU of the block column)
//   main work of block g, once its panels are done:
//     interchanges of every p_j -> columns [0, a_j) U [e, n) and the right-hand sides
//     for j: U_j = L_jj^-1 A[p_j rows, e:n)  (+ forward substitution of b's rows, riding in the same launch);
//            b[a_{j+1}:n) -= L b_j;   A[a_{j+1}:e, e:n) -= L[a_{j+1}:e, p_j] U_j
//     A[e:n, e:n) -= A[e:n, a_0:e) A[a_0:e, e:n): first the columns of block g+1 (then the look-ahead lane starts on them,
//     concurrently with ...) then the rest: the BIG update, on the caller's stream (or the plan's masked stream).
struct Sched {
  ma_lu_plan* P; int n, tstride; int rc = MA_OK;
  std::vector<int> k0s, nbs; int Q = 0, kb = 1, G = 0;
  explicit Sched(ma_lu_plan* P_) : P(P_), n(P_->n), tstride(P_->n + P_->nrhs_max) {
    for (int k0 = 0; k0 < n; k0 += LU_PANEL) { k0s.push_back(k0); nbs.push_back(std::min(n - k0, LU_PANEL)); }
    Q = (int)k0s.size();
    kb = std::max(1, std::min(P->kb, LU_KB_MAX));
    while (kb > 1 && (kb - 1) * LU_PANEL > LU_LANE_TSTRIDE) --kb;   // the lane's interchange staging holds (kb - 1) panels' columns
    G = (Q + kb - 1) / kb;
  }
  int blk_first(int g) const { return g * kb; }
  int blk_last(int g) const { return std::min(Q, (g + 1) * kb); }            // one past
  int blk_end(int g) const { int q = blk_last(g) - 1; return k0s[q] + nbs[q]; }   // first column right of block g
  int* lists_of(int m, int g, int q) const { return P->d_lists[m] + (size_t)((g & 1) * LU_KB_MAX + (q - blk_first(g))) * LU_LISTS_LEN; }
  c64* invd_of(int m, int g, int q) const { return P->d_invd[m] + (size_t)((g & 1) * LU_KB_MAX + (q - blk_first(g))) * LU_NB_MAX * 32; }
  int gemm(int M_, int N_, int K_, const c64* a, const c64* b_, c64* c, hipStream_t s_, bool big_ = false) {
    if (M_ <= 0 || N_ <= 0 || K_ <= 0) return MA_OK;
    P->n_gemm_launch++; P->gemm_flops += 8.0 * M_ * (double)N_ * K_; P->gemm_cbytes += 32.0 * M_ * (double)N_;   // every launch is timed: phase 3 (big) or 5 (the lanes')
    if (big_) { P->n_big_launch++; P->big_flops += 8.0 * M_ * (double)N_ * K_; }
    return lu_launch_zgemm_sub(M_, N_, K_, a, (size_t)n, b_, (size_t)n, c, (size_t)n, s_, big_, P->zgemm_dma);
  }
  // one half-panel of system m at (k0, nb <= 32 columns): the speculative attempts, then -- in line, gated by their verdict, unless the
  // plan is optimistic -- the plan's own panel kernel
  int half_panel(int m, c64* A, int k0, int nb, int* lists, bool clear_tags, hipStream_t st, c64* lrows, int lcol0) {
    const LuPanelWs& ws = P->pws_m[m];
    const bool spec = P->speculate;
    if (spec && (rc = lu_launch_panel_spec(A, n, k0, nb, P->spec[m], P->d_ipiv[m], lists, st, lrows, lcol0, P->optimistic, ws.info))) return rc;
    if (spec && P->optimistic) return MA_OK;
    const int* gate = spec ? P->spec[m].ctl : nullptr;
    if (P->pivoting == MA_LU_PIVOT_TOURNAMENT) return lu_launch_panel_calu(A, n, k0, nb, P->calu[m], ws.info, P->d_ipiv[m], lists, st, lrows, lcol0, gate);
    return lu_launch_panel_reg(A, n, k0, nb, (n - k0 + 255) / 256, P->ncu, ws, P->d_ipiv[m], lists, clear_tags, st, lrows, lcol0, gate);
  }
  // a 64-column panel as two half-panels: left half; its interchanges + U12 + rank-32 update on the right half's columns
  // (lu_lane_step_kernel + one K = 32 update); right half. The pivots land in ipiv as one 64-column panel's: everything after
  // this (lu_lane_step2_kernel, the main lane) is the 64-column schedule
  int panel(int m, c64* A, int q, hipStream_t st) {
    const int k0 = k0s[q], nb = nbs[q];
    const int h1 = std::min(nb, LU_REG_NB), h2 = nb - h1;
    // stale exchange tags of the spinning kernel must not match: cleared at the start of a factorisation and after a very narrow panel
    const bool clear_tags = q == 0 || nbs[q - 1] - LU_REG_NB < 4;
    if ((rc = half_panel(m, A, k0, h1, P->d_half_lists[m], clear_tags, st, nullptr, 0)) || h2 <= 0) return rc;
    const int a1 = k0 + h1;
    if ((rc = lu_launch_lane_step(A, n, k0, h1, P->d_half_lists[m], a1, h2, P->d_half_invd[m], P->pws.timeout, st))) return rc;
    if ((rc = lu_launch_zgemm_sub(n - a1, h2, h1, A + (size_t)a1 * n + k0, (size_t)n, A + (size_t)k0 * n + a1, (size_t)n, A + (size_t)a1 * n + a1, (size_t)n, st, false, P->zgemm_dma))) return rc;
    // (the right half's interchanges on the LEFT half's columns are the first job of lu_lane_step2_kernel, which follows)
    return half_panel(m, A, a1, h2, P->d_half_lists[m] + LU_LISTS_LEN, false, st, P->d_half_l10[m], k0);
  }
  // the look-ahead lane: factor the block column of block g of system m on stream sp
  int lane(int m, c64* A, int g, hipStream_t sp) {
    const int e = blk_end(g);
    for (int q = blk_first(g); q < blk_last(g); ++q) {
      const int k0 = k0s[q], nb = nbs[q], a1 = k0 + nb;
      MA_MARKD(t0, sp);
      if ((rc = panel(m, A, q, sp))) return rc;
      MA_MARKD(t1, sp);
      interval(P, t0, t1, 0);
      // both halves' interchanges on the rest of the block column + U12 + the inverted diagonal blocks + the folded 64-pivot list: one launch
      if ((rc = lu_launch_lane_step2(A, n, k0, nb, P->d_half_lists[m], P->d_half_lists[m] + LU_LISTS_LEN, a1, e - a1, P->d_ipiv[m], lists_of(m, g, q), invd_of(m, g, q), P->pws.timeout,
                                     P->d_half_l10[m], sp))) return rc;
      if (a1 < e) {
        MA_MARK(t2, sp);
        if ((rc = gemm(n - a1, e - a1, nb, A + (size_t)a1 * n + k0, A + (size_t)k0 * n + a1, A + (size_t)a1 * n + a1, sp))) return rc;
        MA_MARK(t3, sp);
        interval(P, t2, t3, 5);
      }
    }
    return MA_OK;
  }
  // the per-panel work of block g right of the block and on the right-hand sides, then the update of the next block's columns
  int main_work(int m, c64* A, c64* B, int nrhs, int g, hipStream_t sm, bool lanes_gemm_phase) {
    const int a0 = k0s[blk_first(g)], e = blk_end(g), nright = n - e;
    const int enext = (g + 1 < G) ? blk_end(g + 1) : e;
    MA_MARKD(t0, sm);
    for (int q = blk_first(g); q < blk_last(g); ++q)
      if ((rc = lu_launch_row_moves(A, n, nbs[q], lists_of(m, g, q), P->d_tmp[m], tstride, 0, k0s[q], e, n, B, nrhs, sm))) return rc;
    MA_MARKD(t1, sm);
    interval(P, t0, t1, 1);
    for (int q = blk_first(g); q < blk_last(g); ++q) {
      const int k0 = k0s[q], nb = nbs[q], a1 = k0 + nb;
      MA_MARKD(u0, sm);
      if ((rc = lu_launch_trsm_mfma(A + (size_t)k0 * n + k0, n, nb, invd_of(m, g, q), A + (size_t)k0 * n + e, (size_t)n, nright, nrhs ? B + k0 : nullptr, (size_t)n, nrhs, sm))) return rc;
      MA_MARKD(u1, sm);
      interval(P, u0, u1, 2);
      for (int r = 0; r < nrhs && a1 < n; ++r)
        if ((rc = lu_launch_zgemv_sub(n - a1, nb, A + (size_t)a1 * n + k0, (size_t)n, B + (size_t)r * n + k0, B + (size_t)r * n + a1, sm))) return rc;
      MA_MARKD(u2, sm);
      interval(P, u1, u2, 4);
      if (a1 < e && nright > 0) {
        MA_MARK(v0, sm);
        if ((rc = gemm(e - a1, nright, nb, A + (size_t)a1 * n + k0, A + (size_t)k0 * n + e, A + (size_t)a1 * n + e, sm))) return rc;
        MA_MARK(v1, sm);
        interval(P, v0, v1, lanes_gemm_phase ? 5 : 3);
      }
    }
    MA_MARK(t3, sm);
    if (nright > 0 && g + 1 < G && (rc = gemm(nright, enext - e, e - a0, A + (size_t)e * n + a0, A + (size_t)a0 * n + e, A + (size_t)e * n + e, sm))) return rc;
    MA_MARK(t4, sm);
    interval(P, t3, t4, lanes_gemm_phase ? 5 : 3);
    return MA_OK;
  }
  // the rest of block g's trailing update: everything right of the next block's columns
  int big_update(c64* A, int g, hipStream_t bs) {
    const int a0 = k0s[blk_first(g)], e = blk_end(g), nright = n - e;
    const int enext = (g + 1 < G) ? blk_end(g + 1) : e;
    if (nright <= 0) return MA_OK;
    MA_MARK(t5, bs);
    if ((rc = gemm(nright, n - enext, e - a0, A + (size_t)e * n + a0, A + (size_t)a0 * n + enext, A + (size_t)e * n + enext, bs, true))) return rc;
    MA_MARK(t6, bs);
    interval(P, t5, t6, 3);
    return MA_OK;
  }
  // backward substitution U x = y, block rows from the bottom
  int backsub(c64* A, c64* B, int nrhs, hipStream_t sb) {
    MA_MARKD(t7, sb);
    for (int q = Q - 1; q >= 0 && nrhs > 0; --q) {
      const int k0 = k0s[q], nb = nbs[q];
      if ((rc = lu_launch_trsv(true, A + (size_t)k0 * n + k0, n, nb, B + k0, (size_t)n, nrhs, sb))) return rc;
      for (int r = 0; r < nrhs && k0 > 0; ++r)
        if ((rc = lu_launch_zgemv_sub(k0, nb, A + k0, (size_t)n, B + (size_t)r * n + k0, B + (size_t)r * n, sb))) return rc;
    }
    MA_MARKD(t8, sb);
    interval(P, t7, t8, 4);
    return MA_OK;
  }
};

void reset_accounts(ma_lu_plan* P) {
  P->ev_used = 0; P->iv.clear(); P->n_gemm_launch = 0; P->gemm_flops = 0.0; P->gemm_cbytes = 0.0; P->n_big_launch = 0; P->big_flops = 0.0; P->ev_valid = false;
}

// Factor the matrices in place and solve for nrhs right-hand sides each (d_B[nrhs][n]); everything asynchronous. A batch of
// independent systems of the same size moves in lock step, block by block: each system has its own look-ahead lane, which also
// carries its per-panel work, so that one system's latency-bound chain runs under the others' updates; the caller's stream carries
// the big updates, back to back over the systems. (A long sequence of systems is better served by the staged entries below.)
int factor_solve_batch(ma_lu_plan* P, int nmat, c64* const* As, c64* const* Bs, int32_t nrhs, hipStream_t st) {
  Sched S(P);
  int& rc = S.rc;
  reset_accounts(P);
  P->last_batch = nmat;
  MA_HIP(hipMemsetAsync(P->pws.info, 0, 64, st));
  MA_MARK(e_begin, st);
  const int G = S.G;
  const bool many = nmat > 1;
  // the big updates run on the plan's masked stream when it splits the chip; `st` then only brackets the call
  hipStream_t bs = P->cu_split ? P->big_stream : st;
  MA_HIP(hipEventRecord(P->ev_start, st));
  for (int m = 0; m < nmat; ++m) MA_HIP(hipStreamWaitEvent(P->panel_streams[m], P->ev_start, 0));
  if (bs != st) MA_HIP(hipStreamWaitEvent(bs, P->ev_start, 0));
  for (int m = 0; m < nmat; ++m) {
    if ((rc = S.lane(m, As[m], 0, P->panel_streams[m]))) return rc;
    MA_HIP(hipEventRecord(P->ev_panel[m], P->panel_streams[m]));
  }
  for (int g = 0; g < G; ++g) {
    const int nright = P->n - S.blk_end(g);
    for (int m = 0; m < nmat; ++m) {
      c64* B = Bs ? Bs[m] : nullptr;
      hipStream_t sp = P->panel_streams[m], sm = many ? sp : bs;
      MA_HIP(hipStreamWaitEvent(sm, P->ev_panel[m], 0));
      if (many && g > 0) MA_HIP(hipStreamWaitEvent(sm, P->ev_big[m], 0));      // block g-1's big update of this system
      if ((rc = S.main_work(m, As[m], B, nrhs, g, sm, many))) return rc;
      if (many) MA_HIP(hipEventRecord(P->ev_mid[m], sm));
      if (nright > 0 && g + 1 < G) {
        // the next block's columns are up to date: factor them beside the rest of the update
        if (sm != sp) { MA_HIP(hipEventRecord(P->ev_narrow[m], sm)); MA_HIP(hipStreamWaitEvent(sp, P->ev_narrow[m], 0)); }
        if ((rc = S.lane(m, As[m], g + 1, sp))) return rc;
        MA_HIP(hipEventRecord(P->ev_panel[m], sp));
      }
    }
    for (int m = 0; m < nmat && nright > 0; ++m) {
      if (many) MA_HIP(hipStreamWaitEvent(bs, P->ev_mid[m], 0));
      if ((rc = S.big_update(As[m], g, bs))) return rc;
      if (many) MA_HIP(hipEventRecord(P->ev_big[m], bs));
    }
  }
  if (nrhs > 0) {
    // a chain of 2 Q short launches per system: the systems of a batch run theirs side by side on their own streams
    for (int m = 0; m < nmat; ++m) {
      hipStream_t sb = many ? P->panel_streams[m] : bs;
      if (many && G > 0) MA_HIP(hipStreamWaitEvent(sb, P->ev_big[m], 0));
      if ((rc = S.backsub(As[m], Bs[m], nrhs, sb))) return rc;
      if (many) MA_HIP(hipEventRecord(P->ev_panel[m], sb));
    }
  } else if (many) {
    for (int m = 0; m < nmat; ++m) MA_HIP(hipEventRecord(P->ev_panel[m], P->panel_streams[m]));
  }
  if (many) for (int m = 0; m < nmat; ++m) MA_HIP(hipStreamWaitEvent(st, P->ev_panel[m], 0));
  if (bs != st) { MA_HIP(hipEventRecord(P->ev_start, bs)); MA_HIP(hipStreamWaitEvent(st, P->ev_start, 0)); }
  MA_MARK(e_end, st);
  interval(P, e_begin, e_end, 6);
  P->ev_last = e_end;
  if (P->timing) P->ev_valid = true;
  return MA_OK;
}

}  // namespace

// ------------------------------------------------------------------ staged use: a pipeline over many systems
// factor_solve_batch moves its systems in lock step: all of them are in the update-bound early blocks together and in the
// latency-bound last blocks together. A driver that has a long sequence of systems (a frequency sweep) can instead keep the
// slots at DIFFERENT block indices -- slot s starts a third of a factorisation after slot s-1 -- so that in every round one
// slot brings a big update, one a medium one and one a small one: the caller's stream always has update work and every
// slot's latency-bound chain has the time of three updates to finish. The driver calls, per slot, stage_begin (A and b of
// the next system are ready on `stream`), then one stage_round per block index 0..G-1 together with the other slots, then
// stage_finish (backward substitution; `stream` waits for it). Same kernels, same arithmetic as factor_solve_batch.
extern "C" {

int ma_lu_plan_num_blocks(ma_lu_plan_t* P, int32_t* blocks) {
  MA_REQUIRE(P && blocks, MA_ERR_INVALID, "NULL argument");
  Sched S(P);
  *blocks = S.G;
  return MA_OK;
}
// before a pipelined run: clear the status words and the timing accumulators
int ma_lu_plan_stage_reset(ma_lu_plan_t* P, void* stream) {
  MA_REQUIRE(P, MA_ERR_INVALID, "NULL plan");
  // a plan that splits the chip runs its big updates on a CU-masked stream, which is a BLOCKING stream: a driver on the NULL stream would
  // serialise against it at every launch and lose the lanes' overlap without any error
  MA_REQUIRE(!(P->cu_split && stream == nullptr), MA_ERR_INVALID, "this plan splits the chip: drive its staged schedule from ma_lu_plan_main_stream (or any non-blocking stream), not from the NULL stream");
  MA_HIP(hipSetDevice(P->device));
  hipStream_t st = (hipStream_t)stream;
  int rc;
  reset_accounts(P);
  P->last_batch = 0;
  MA_HIP(hipMemsetAsync(P->pws.info, 0, 64, st));
  MA_MARK(e0, st);
  P->stage_first_mark = e0;
  return MA_OK;
}
// the stream slot `slot`'s chain runs on (idle between stage_finish and the next stage_begin: a driver may assemble the
// slot's next system there, beside the other slots' work, and pass the same stream to stage_begin)
int ma_lu_plan_slot_stream(ma_lu_plan_t* P, int32_t slot, void** stream) {
  MA_REQUIRE(P && stream && slot >= 0 && slot < LU_BATCH_MAX, MA_ERR_INVALID, "bad argument");
  *stream = (void*)P->panel_streams[slot];
  return MA_OK;
}
// rounds between the starts of two slots of the staged schedule: the slots evenly spread, so that the sum of the updates of one round
// never falls far below the chain of the slot that is closest to its end (50.8 ms against 51.4 at G / (slots + 1): r03 (f))
int ma_lu_plan_stage_spacing(ma_lu_plan_t* P, int32_t slots, int32_t* spacing) {
  MA_REQUIRE(P && spacing && slots >= 1, MA_ERR_INVALID, "bad argument");
  Sched S(P);
  *spacing = std::max(1, (S.G + slots / 2) / slots);
  return MA_OK;
}
// how the plan splits the chip: CUs its big updates stay off (0: no split; the mask's bits [0, panel_cus): panel_cus / 8 CUs of every XCD) and the chip's CUs
int ma_lu_plan_cu_split(ma_lu_plan_t* P, int32_t* panel_cus, int32_t* total_cus) {
  MA_REQUIRE(P && panel_cus && total_cus, MA_ERR_INVALID, "bad argument");
  *panel_cus = P->cu_split; *total_cus = P->ncu;
  return MA_OK;
}
// the stream the plan runs its big trailing updates on when the chip is split: masked to the update CUs. A driver that issues its own
// work between stage calls (assembly) puts it there instead of on a stream of its own (one hardware queue less); NULL when not split
int ma_lu_plan_main_stream(ma_lu_plan_t* P, void** stream) {
  MA_REQUIRE(P && stream, MA_ERR_INVALID, "bad argument");
  *stream = (void*)(P->cu_split ? P->big_stream : nullptr);
  return MA_OK;
}
int ma_lu_plan_stage_begin(ma_lu_plan_t* P, int32_t slot, void* dA, void* dB, int32_t nrhs, void* stream) {
  MA_REQUIRE(P && dA, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(slot >= 0 && slot < LU_BATCH_MAX, MA_ERR_INVALID, "slot %d outside 0..%d", slot, LU_BATCH_MAX - 1);
  MA_REQUIRE(nrhs >= 0 && nrhs <= P->nrhs_max && (nrhs == 0 || dB), MA_ERR_DIM, "nrhs must be 0..%d", P->nrhs_max);
  MA_REQUIRE(!(P->cu_split && stream == nullptr), MA_ERR_INVALID, "this plan splits the chip: drive its staged schedule from ma_lu_plan_main_stream (or any non-blocking stream), not from the NULL stream");
  MA_HIP(hipSetDevice(P->device));
  int rc = P->ensure_batch(slot + 1);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  P->cur_A[slot] = (c64*)dA; P->cur_B[slot] = (c64*)dB; P->cur_nrhs = nrhs;
  if (slot + 1 > P->last_batch) P->last_batch = slot + 1;
  MA_HIP(hipMemsetAsync(P->pws.info + slot, 0, sizeof(int), st));          // this slot's status word
  MA_HIP(hipEventRecord(P->ev_prep[slot], st));
  MA_HIP(hipStreamWaitEvent(P->panel_streams[slot], P->ev_prep[slot], 0));
  Sched S(P);
  return S.lane(slot, P->cur_A[slot], 0, P->panel_streams[slot]);
}
// one round: slot slots[i] does block blocks[i] (consecutive rounds of a slot use consecutive blocks 0..G-1)
int ma_lu_plan_stage_round(ma_lu_plan_t* P, int32_t count, const int32_t* slots, const int32_t* blocks, void* stream) {
  MA_REQUIRE(P && slots && blocks && count >= 0 && count <= LU_BATCH_MAX, MA_ERR_INVALID, "bad argument");
  MA_HIP(hipSetDevice(P->device));
  Sched S(P);
  int& rc = S.rc;
  for (int i = 0; i < count; ++i)
    MA_REQUIRE(slots[i] >= 0 && slots[i] < LU_BATCH_MAX && P->cur_A[slots[i]] && blocks[i] >= 0 && blocks[i] < S.G, MA_ERR_INVALID, "slot %d / block %d", slots[i], blocks[i]);
  hipStream_t bs = P->cu_split ? P->big_stream : (hipStream_t)stream;
  for (int i = 0; i < count; ++i) {
    const int m = slots[i], g = blocks[i];
    hipStream_t sm = P->panel_streams[m];
    if (g > 0) MA_HIP(hipStreamWaitEvent(sm, P->ev_big[m], 0));                  // block g-1's big update of this slot
    if ((rc = S.main_work(m, P->cur_A[m], P->cur_B[m], P->cur_nrhs, g, sm, true))) return rc;
    MA_HIP(hipEventRecord(P->ev_mid[m], sm));
    if (P->n - S.blk_end(g) > 0 && g + 1 < S.G && (rc = S.lane(m, P->cur_A[m], g + 1, sm))) return rc;
  }
  // the big updates of the round, smallest first: the slot closest to the end of its factorisation has the least slack in
  // its chain (its next block waits for this update), the one at the start has the most
  int order[LU_BATCH_MAX];
  for (int i = 0; i < count; ++i) order[i] = i;
  std::sort(order, order + count, [&](int a, int b) { return blocks[a] > blocks[b]; });
  for (int i = 0; i < count; ++i) {
    const int m = slots[order[i]], g = blocks[order[i]];
    if (P->n - S.blk_end(g) <= 0) continue;
    MA_HIP(hipStreamWaitEvent(bs, P->ev_mid[m], 0));
    if ((rc = S.big_update(P->cur_A[m], g, bs))) return rc;
    MA_HIP(hipEventRecord(P->ev_big[m], bs));
  }
  return MA_OK;
}
// after stage_finish: copy the slot's status word (0; 1 + the column of the first zero pivot; -1: an optimistic plan met a panel it gave up) to a device int on `stream`
int ma_lu_plan_stage_info_dev(ma_lu_plan_t* P, int32_t slot, int32_t* d_out, void* stream) {
  MA_REQUIRE(P && d_out && slot >= 0 && slot < LU_BATCH_MAX, MA_ERR_INVALID, "bad argument");
  MA_HIP(hipSetDevice(P->device));
  MA_HIP(hipMemcpyAsync(d_out, P->pws.info + slot, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MA_OK;
}
int ma_lu_plan_stage_finish(ma_lu_plan_t* P, int32_t slot, void* stream) {
  MA_REQUIRE(P && slot >= 0 && slot < LU_BATCH_MAX && P->cur_A[slot], MA_ERR_INVALID, "bad argument");
  MA_HIP(hipSetDevice(P->device));
  hipStream_t st = (hipStream_t)stream, sb = P->panel_streams[slot];
  Sched S(P);
  int& rc = S.rc;
  MA_HIP(hipStreamWaitEvent(sb, P->ev_big[slot], 0));          // (a one-block plan has no big update: the event is then an old, completed one)
  if ((rc = S.backsub(P->cur_A[slot], P->cur_B[slot], P->cur_nrhs, sb))) return rc;
  MA_HIP(hipEventRecord(P->ev_panel[slot], sb));
  MA_HIP(hipStreamWaitEvent(st, P->ev_panel[slot], 0));       // `st` waits for what was recorded on the lane; the run's last mark
  MA_MARK(e_end, st);
  interval(P, P->stage_first_mark, e_end, 6);
  P->ev_last = e_end;
  if (P->timing) P->ev_valid = true;
  return MA_OK;
}
}  // extern "C"

// Solve with factors that an earlier ma_lu_plan_factor_solve_dev call on THIS plan left in d_A (the plan still holds
// the pivots): b <- P b panel by panel, forward substitution with the unit-lower factor, backward with the upper one.
// LuFactorization::solve (lu.rs:38-78).
static int solve_only(ma_lu_plan* P, c64* A, c64* B, int32_t nrhs, hipStream_t st) {
  Sched S(P);
  const int n = P->n, Q = S.Q;
  int rc;
  // the stored factors are in their final row order (later interchanges were applied to the earlier L columns), so every
  // interchange goes onto b first (zgetrs: laswp, then the triangular solves)
  for (int q = 0; q < Q; ++q)
    if ((rc = lu_launch_swaps(A, n, S.k0s[q], S.nbs[q], P->d_ipiv[0], P->d_lists[0], P->d_tmp[0], S.tstride, 0, 0, 0, 0, B, nrhs, nullptr, nullptr, st))) return rc;
  for (int q = 0; q < Q; ++q) {
    const int k0 = S.k0s[q], nb = S.nbs[q], a1 = k0 + nb;
    if ((rc = lu_launch_swaps(A, n, k0, nb, P->d_ipiv[0], P->d_lists[0], P->d_tmp[0], S.tstride, 0, 0, 0, 0, nullptr, 0, P->d_invd[0], nullptr, st))) return rc;   // inverted diagonal blocks only
    if ((rc = lu_launch_trsm_mfma(A + (size_t)k0 * n + k0, n, nb, P->d_invd[0], A, (size_t)n, 0, B + k0, (size_t)n, nrhs, st))) return rc;
    for (int r = 0; r < nrhs && a1 < n; ++r)
      if ((rc = lu_launch_zgemv_sub(n - a1, nb, A + (size_t)a1 * n + k0, (size_t)n, B + (size_t)r * n + k0, B + (size_t)r * n + a1, st))) return rc;
  }
  return S.backsub(A, B, nrhs, st);
}

extern "C" {

int ma_lu_plan_solve_dev(ma_lu_plan_t* P, void* dA_factored, void* dB, int32_t nrhs, void* stream) {
  MA_REQUIRE(P && dA_factored && dB, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(nrhs >= 1 && nrhs <= P->nrhs_max, MA_ERR_DIM, "nrhs must be 1..%d", P->nrhs_max);
  MA_HIP(hipSetDevice(P->device));
  const bool t = P->timing; P->timing = false;                 // (the substitutions' marks belong to a factorisation's accounts)
  const int rc = solve_only(P, (c64*)dA_factored, (c64*)dB, nrhs, (hipStream_t)stream);
  P->timing = t;
  return rc;
}

int ma_lu_plan_factor_solve_dev(ma_lu_plan_t* P, void* dA, void* dB, int32_t nrhs, void* stream) {
  MA_REQUIRE(P && dA, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(nrhs >= 0 && nrhs <= P->nrhs_max, MA_ERR_DIM, "nrhs must be 0..%d", P->nrhs_max);
  MA_REQUIRE(nrhs == 0 || dB, MA_ERR_INVALID, "d_B is NULL");
  MA_HIP(hipSetDevice(P->device));
  c64* A = (c64*)dA; c64* B = (c64*)dB;
  return factor_solve_batch(P, 1, &A, &B, nrhs, (hipStream_t)stream);
}

// nmat (1..LU_GROUP_MAX) independent n x n systems, e.g. the frequencies of a sweep kept in flight together.
int ma_lu_plan_factor_solve_batch_dev(ma_lu_plan_t* P, int32_t nmat, void* const* dAs, void* const* dBs, int32_t nrhs, void* stream) {
  MA_REQUIRE(P && dAs, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(nmat >= 1 && nmat <= LU_GROUP_MAX, MA_ERR_INVALID, "batch must be 1..%d systems", LU_GROUP_MAX);
  MA_REQUIRE(nrhs >= 0 && nrhs <= P->nrhs_max, MA_ERR_DIM, "nrhs must be 0..%d", P->nrhs_max);
  MA_REQUIRE(nrhs == 0 || dBs, MA_ERR_INVALID, "d_Bs is NULL");
  c64* As[LU_BATCH_MAX]; c64* Bs[LU_BATCH_MAX];
  for (int m = 0; m < nmat; ++m) {
    MA_REQUIRE(dAs[m] && (nrhs == 0 || dBs[m]), MA_ERR_INVALID, "system %d has a NULL pointer", m);
    As[m] = (c64*)dAs[m]; Bs[m] = nrhs ? (c64*)dBs[m] : nullptr;
    for (int o = 0; o < m; ++o) MA_REQUIRE(As[o] != As[m], MA_ERR_INVALID, "systems %d and %d alias", o, m);
  }
  MA_HIP(hipSetDevice(P->device));
  int rc = P->ensure_batch(nmat);
  if (rc) return rc;
  return factor_solve_batch(P, nmat, As, Bs, nrhs, (hipStream_t)stream);
}

int ma_lu_plan_status(ma_lu_plan_t* P, void* stream) {
  MA_REQUIRE(P, MA_ERR_INVALID, "NULL plan");
  MA_HIP(hipSetDevice(P->device));
  MA_HIP(hipStreamSynchronize((hipStream_t)stream));
  for (int i = 0; i < LU_BATCH_MAX; ++i) if (P->panel_streams[i]) MA_HIP(hipStreamSynchronize(P->panel_streams[i]));
  if (P->big_stream) MA_HIP(hipStreamSynchronize(P->big_stream));
  int info[16];
  MA_HIP(hipMemcpy(info, P->pws.info, sizeof(info), hipMemcpyDeviceToHost));
  MA_REQUIRE(info[LU_BATCH_MAX] != 2, MA_ERR_HIP, "a panel left a pivot outside its range: the factorisation was abandoned (no rows were moved with it)");
  MA_REQUIRE(info[LU_BATCH_MAX] == 0, MA_ERR_HIP, "panel factorisation abandoned: an exchange between the co-resident workgroups did not complete within its limit");
  for (int m = 0; m < P->last_batch; ++m) {
    MA_REQUIRE(info[m] >= 0, MA_ERR_RETRY, "system %d: a speculative panel was given up and the plan runs without its own panel kernel behind it (optimistic mode): solve it again with ma_lu_plan_set_speculation(plan, MA_LU_SPECULATE_VERIFIED)", m);
    MA_REQUIRE(info[m] == 0, MA_ERR_SINGULAR, "system %d is singular: zero pivot at column %d", m, info[m] - 1);
  }
  return MA_OK;
}

int ma_lu_plan_last_timing(ma_lu_plan_t* P, double* out8) {
  MA_REQUIRE(P && out8, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(P->ev_valid && P->ev_last >= 0, MA_ERR_INVALID, "no timed factorisation has run on this plan");
  MA_HIP(hipSetDevice(P->device));
  MA_HIP(hipEventSynchronize(P->ev[P->ev_last]));
  for (int i = 0; i < LU_BATCH_MAX; ++i) if (P->panel_streams[i]) MA_HIP(hipStreamSynchronize(P->panel_streams[i]));
  if (P->big_stream) MA_HIP(hipStreamSynchronize(P->big_stream));
  for (int i = 0; i < 8; ++i) out8[i] = 0.0;
  for (const auto& v : P->iv) {
    float ms = 0.f;
    MA_HIP(hipEventElapsedTime(&ms, P->ev[v.a], P->ev[v.b]));
    if (v.phase >= 0 && v.phase < 5) out8[v.phase] += ms;
    else if (v.phase == 5) out8[7] += ms;
    else if (v.phase == 6) out8[6] = ms;
  }
  out8[5] = P->n_gemm_launch;
  return MA_OK;
}

// Diagnostic: the timed intervals of one phase of the last call as (start, end) in ms after the call's first mark, in the order
// they were enqueued (phase 3 = the big updates on the caller's stream: their gaps are that stream's waits). After last_timing.
int ma_lu_plan_dump_intervals(ma_lu_plan_t* P, int32_t phase, double* out_pairs, int32_t capacity, int32_t* count) {
  MA_REQUIRE(P && out_pairs && count && capacity >= 0, MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(P->ev_valid && P->stage_first_mark >= 0, MA_ERR_INVALID, "no timed staged run on this plan");
  MA_HIP(hipSetDevice(P->device));
  int c = 0;
  for (const auto& v : P->iv) {
    if (v.phase != phase) continue;
    if (c < capacity) {
      float a = 0.f, b = 0.f;
      MA_HIP(hipEventElapsedTime(&a, P->ev[P->stage_first_mark], P->ev[v.a]));
      MA_HIP(hipEventElapsedTime(&b, P->ev[P->stage_first_mark], P->ev[v.b]));
      out_pairs[2 * c] = a; out_pairs[2 * c + 1] = b;
    }
    ++c;
  }
  *count = c;
  return MA_OK;
}

// the big trailing updates on the caller's stream alone (their time is out8[3] of ma_lu_plan_last_timing)
int ma_lu_plan_last_big_update_stats(ma_lu_plan_t* P, double* launches, double* flops) {
  MA_REQUIRE(P && launches && flops, MA_ERR_INVALID, "NULL argument");
  *launches = P->n_big_launch; *flops = P->big_flops;
  return MA_OK;
}
// Update (zgemm) launches of the last call, main lane and look-ahead lanes, all systems of the batch: count, algorithmic
// flops (8 M N K each) and algorithmic C bytes (32 M N each). Their time is out8[3] + out8[7] of ma_lu_plan_last_timing.
int ma_lu_plan_last_update_stats(ma_lu_plan_t* P, double* launches, double* flops, double* c_bytes) {
  MA_REQUIRE(P && launches && flops && c_bytes, MA_ERR_INVALID, "NULL argument");
  *launches = P->n_gemm_launch; *flops = P->gemm_flops; *c_bytes = P->gemm_cbytes;
  return MA_OK;
}

int ma_zgesv(int32_t n, ma_c64* A, ma_c64* b, int32_t* ipiv) { return ma_zgesv_pivoting(n, A, b, ipiv, MA_LU_PIVOT_PARTIAL); }

int ma_zgesv_pivoting(int32_t n, ma_c64* A, ma_c64* b, int32_t* ipiv, int32_t pivoting) {
  MA_REQUIRE(n >= 0, MA_ERR_DIM, "n is negative");
  if (n == 0) return MA_OK;
  MA_REQUIRE(A && b, MA_ERR_INVALID, "A or b is NULL");
  int dev = 0;
  if (const char* s = getenv("MA_DEVICE")) dev = atoi(s);
  ma_lu_plan_t* P = nullptr;
  int rc = ma_lu_plan_create_pivoting(n, dev, pivoting, &P);
  if (rc) return rc;
  void *dA = nullptr, *db = nullptr;
  const size_t nn = (size_t)n;
  hipError_t e = hipMalloc(&dA, nn * nn * sizeof(c64));
  if (e == hipSuccess) e = hipMalloc(&db, nn * sizeof(c64));
  if (e != hipSuccess) {
    set_error("hipMalloc for a %d x %d system failed: %s", n, n, hipGetErrorString(e));
    if (dA) (void)hipFree(dA);
    ma_lu_plan_destroy(P);
    return MA_ERR_NOMEM;
  }
  e = hipMemcpy(dA, A, nn * nn * sizeof(c64), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(db, b, nn * sizeof(c64), hipMemcpyHostToDevice);
  if (e != hipSuccess) { set_error("upload failed: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; }
  if (!rc) rc = ma_lu_plan_factor_solve_dev(P, dA, db, 1, nullptr);
  if (!rc) rc = ma_lu_plan_status(P, nullptr);
  if (!rc || rc == MA_ERR_SINGULAR) {
    // the factors are returned in either case (LAPACK leaves them in A); x only when non-singular
    hipError_t e2 = hipMemcpy(A, dA, nn * nn * sizeof(c64), hipMemcpyDeviceToHost);
    if (e2 == hipSuccess && !rc) e2 = hipMemcpy(b, db, nn * sizeof(c64), hipMemcpyDeviceToHost);
    if (e2 == hipSuccess && ipiv) e2 = hipMemcpy(ipiv, P->d_ipiv[0], nn * sizeof(int), hipMemcpyDeviceToHost);
    if (e2 != hipSuccess) { set_error("copy back failed: %s", hipGetErrorString(e2)); rc = MA_ERR_HIP; }
  }
  (void)hipFree(dA); (void)hipFree(db);
  ma_lu_plan_destroy(P);
  return rc;
}

// lu_solve(&a, &b) -> x (lu.rs:142-153) with the reference's own signature: A and b are not modified and the factors are
// not brought back (half the PCIe traffic of ma_zgesv).
int ma_lu_solve(int32_t n, const ma_c64* A, const ma_c64* b, ma_c64* x) {
  MA_REQUIRE(n >= 0, MA_ERR_DIM, "n is negative");
  if (n == 0) return MA_OK;
  MA_REQUIRE(A && b && x, MA_ERR_INVALID, "NULL argument");
  ma_lu_factorization_t* F = nullptr;
  int rc = ma_lu_factorize(n, A, &F);
  if (!rc) rc = ma_lu_factorization_solve(F, b, x);
  ma_lu_factorization_destroy(F);
  return rc;
}

// lu_factorize (lu.rs:83-137): the factors stay in HBM behind the handle
struct ma_lu_factorization { ma_lu_plan* plan = nullptr; c64* dA = nullptr; c64* db = nullptr; int n = 0; };

int ma_lu_factorize(int32_t n, const ma_c64* A, ma_lu_factorization_t** out) {
  MA_REQUIRE(out, MA_ERR_INVALID, "out is NULL");
  *out = nullptr;
  MA_REQUIRE(n > 0, MA_ERR_DIM, "n must be positive");
  MA_REQUIRE(A, MA_ERR_INVALID, "A is NULL");
  int dev = 0;
  if (const char* s = getenv("MA_DEVICE")) dev = atoi(s);
  ma_lu_factorization* F = new (std::nothrow) ma_lu_factorization();
  MA_REQUIRE(F, MA_ERR_NOMEM, "host allocation failed");
  F->n = n;
  int rc = ma_lu_plan_create_pivoting(n, dev, MA_LU_PIVOT_PARTIAL, &F->plan);     // lu.rs:83-137 exposes `pivots`: LAPACK's
  const size_t nn = (size_t)n;
  if (!rc) {
    hipError_t e = hipMalloc(&F->dA, nn * nn * sizeof(c64));
    if (e == hipSuccess) e = hipMalloc(&F->db, nn * sizeof(c64));
    if (e == hipSuccess) e = hipMemcpy(F->dA, A, nn * nn * sizeof(c64), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error("factorisation buffers for n = %d: %s", n, hipGetErrorString(e)); rc = MA_ERR_NOMEM; }
  }
  if (!rc) rc = ma_lu_plan_factor_solve_dev(F->plan, F->dA, nullptr, 0, nullptr);
  if (!rc) rc = ma_lu_plan_status(F->plan, nullptr);
  if (rc) { ma_lu_factorization_destroy(F); return rc; }
  *out = F;
  return MA_OK;
}

// LuFactorization::solve (lu.rs:38-78)
int ma_lu_factorization_solve(ma_lu_factorization_t* F, const ma_c64* b, ma_c64* x) {
  MA_REQUIRE(F && b && x, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(F->plan->device));
  const size_t nn = (size_t)F->n;
  MA_HIP(hipMemcpy(F->db, b, nn * sizeof(c64), hipMemcpyHostToDevice));
  int rc = ma_lu_plan_solve_dev(F->plan, F->dA, F->db, 1, nullptr);
  if (rc) return rc;
  MA_HIP(hipMemcpy(x, F->db, nn * sizeof(c64), hipMemcpyDeviceToHost));
  return MA_OK;
}

int ma_lu_factorization_destroy(ma_lu_factorization_t* F) {
  if (!F) return MA_OK;
  if (F->dA) (void)hipFree(F->dA);
  if (F->db) (void)hipFree(F->db);
  if (F->plan) ma_lu_plan_destroy(F->plan);
  delete F;
  return MA_OK;
}

// C <- C - A B on host buffers (row-major, tight leading dimensions) with the trailing update's kernel: the f64 matrix cores, three
// real products per complex product
int ma_zgemm_sub(int32_t M, int32_t N, int32_t K, const ma_c64* A, const ma_c64* B, ma_c64* C) {
  MA_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, MA_ERR_INVALID, "bad argument");
  int dev = 0;
  if (const char* s = getenv("MA_DEVICE")) dev = atoi(s);
  int rc = use_device(dev);
  if (rc) return rc;
  c64 *dA = nullptr, *dB = nullptr, *dC = nullptr;
  MA_HIP(hipMalloc(&dA, sizeof(c64) * (size_t)M * K));
  MA_HIP(hipMalloc(&dB, sizeof(c64) * (size_t)K * N));
  MA_HIP(hipMalloc(&dC, sizeof(c64) * (size_t)M * N));
  MA_HIP(hipMemcpy(dA, A, sizeof(c64) * (size_t)M * K, hipMemcpyHostToDevice));
  MA_HIP(hipMemcpy(dB, B, sizeof(c64) * (size_t)K * N, hipMemcpyHostToDevice));
  MA_HIP(hipMemcpy(dC, C, sizeof(c64) * (size_t)M * N, hipMemcpyHostToDevice));
  { bool dma = true; if (const char* e0 = getenv("MA_ZGEMM_DMA")) dma = atoi(e0) != 0;
    rc = lu_launch_zgemm_sub(M, N, K, dA, (size_t)K, dB, (size_t)N, dC, (size_t)N, nullptr, false, dma); }
  if (!rc) { hipError_t e = hipMemcpy(C, dC, sizeof(c64) * (size_t)M * N, hipMemcpyDeviceToHost); if (e != hipSuccess) { set_error("copy back: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; } }
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  return rc;
}

// The admission rule of the kernels whose workgroups wait for one another as a pure function (lu_kernels.hip, "Residency"): such
// workgroups of `lds_bytes` of LDS and `regs` vector registers per lane that a CU can ALWAYS take, whatever offsets terminating kernels left them at.
int ma_lu_panel_slots_per_cu(int64_t lds_bytes, int32_t regs, int32_t* slots) {
  MA_REQUIRE(slots && lds_bytes > 0 && regs >= 0, MA_ERR_INVALID, "bad argument");
  *slots = lu_panel_slots_per_cu((size_t)lds_bytes, regs);
  return MA_OK;
}

// MFMA f64 issue-rate probe (bench.py uses it to state the measured matrix-core peak next to the
// datasheet figure): returns TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64 on every CU.
int ma_probe_mfma_f64(int device, double* tflops) {
  MA_REQUIRE(tflops, MA_ERR_INVALID, "NULL argument");
  int rc = use_device(device);
  if (rc) return rc;
  hipDeviceProp_t prop;
  MA_HIP(hipGetDeviceProperties(&prop, device));
  const int blocks = prop.multiProcessorCount * 2, iters = 20000;   // 6.5 ms per launch: the ramp of a 1.3 ms launch read 69 TFLOP/s where tools/mfma_peak_probe.py reads 77
  double* d = nullptr;
  MA_HIP(hipMalloc(&d, sizeof(double) * 256 * (size_t)blocks));
  hipEvent_t a, b;
  MA_HIP(hipEventCreate(&a)); MA_HIP(hipEventCreate(&b));
  rc = lu_launch_mfma_probe(d, blocks, 2000, nullptr);   // warm-up (clocks)
  MA_HIP(hipEventRecord(a, nullptr));
  for (int q = 0; q < 3 && !rc; ++q) rc = lu_launch_mfma_probe(d, blocks, iters, nullptr);
  MA_HIP(hipEventRecord(b, nullptr));
  MA_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  MA_HIP(hipEventElapsedTime(&ms, a, b));
  const double flops = 3.0 * (double)blocks * 4.0 * iters * 12.0 * (2.0 * 16 * 16 * 4);
  *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipFree(d);
  return rc;
}

#ifdef MA_DIAGNOSTICS
// Diagnostic build only (tools/*.py load it through MA_LIB_PATH): background load on a stream of the caller's -- `repeat` launches of the
// trailing-update kernel on device buffers (tight leading dimensions), or of the matrix-core probe (d_out: 256 * blocks doubles)
int ma_diag_zgemm_dev(int32_t M, int32_t N, int32_t K, const void* dA, const void* dB, void* dC, int32_t repeat, void* stream) {
  MA_REQUIRE(dA && dB && dC && M > 0 && N > 0 && K > 0, MA_ERR_INVALID, "bad argument");
  int rc = MA_OK;
  for (int i = 0; i < repeat && !rc; ++i)
    rc = lu_launch_zgemm_sub(M, N, K, (const c64*)dA, (size_t)K, (const c64*)dB, (size_t)N, (c64*)dC, (size_t)N, (hipStream_t)stream, true, true);
  return rc;
}
int ma_diag_mfma_burn(void* d_out, int32_t blocks, int32_t iters, int32_t repeat, void* stream) {
  MA_REQUIRE(d_out && blocks > 0 && iters > 0, MA_ERR_INVALID, "bad argument");
  int rc = MA_OK;
  for (int i = 0; i < repeat && !rc; ++i) rc = lu_launch_mfma_probe((double*)d_out, blocks, iters, (hipStream_t)stream);
  return rc;
}
#endif

}  // extern "C"
