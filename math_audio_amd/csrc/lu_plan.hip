// lu_plan.hip — host driver of the blocked right-looking LU and the C-ABI of the dense solve
// (replaces lu_solve, math-solvers/src/direct/lu.rs:142-153).
#include "lu_kernels.hpp"
#include <vector>
#include <cstring>
#include <algorithm>
#include <new>

using namespace ma;

#define LU_LISTS_LEN (1 + 4 * LU_NB_MAX)
// Round-3 default for systems of MA_LU_PAIR_MIN_N..MA_LU_PAIR_MAX_N rows (the sizes measured: 6 000 - 14 000 rows, every one faster
// on the staged, the batched and the one-system path, profiles/r03_lu_panel_experiments.md): 64-column panels factored as two
// register half-panels (MA_LU_REG_PANEL=2) and the big trailing updates kept off 64 of the 256 CUs (MA_LU_CU_SPLIT=64).
// Outside that range, on a chip that is not 256 CUs, and for a plan tuned with the LDS family's switches (MA_LU_NB, MA_LU_RPB,
// MA_LU_BATCH_PANEL), the round-2 schedule.
#ifndef MA_LU_PAIR_MIN_N
#define MA_LU_PAIR_MIN_N 4096
#endif
#ifndef MA_LU_PAIR_MAX_N
#define MA_LU_PAIR_MAX_N 16384
#endif
#ifndef MA_LU_CU_SPLIT_DEFAULT
#define MA_LU_CU_SPLIT_DEFAULT 64
#endif
#ifndef MA_LU_CU_SPLIT_TOURNAMENT
#define MA_LU_CU_SPLIT_TOURNAMENT 32        // CUs the big updates stay off in a tournament-pivoting plan: one per shader engine of every XCD (see below)
#endif
#define LU_KB_MAX 8                         // panels per trailing update
#define LU_LANE_TSTRIDE 512                   // the lane's interchanges touch at most (kb-1) panels' columns: 7 x 64 or 3 x 128

struct ma_lu_plan {
  int device = 0;
  int n = 0;
  int ncu = 256;
  void* ws_block = nullptr;       // one allocation: sync words | info | cand | candrow | diagrow | lists | ipiv
  LuPanelWs pws{};                // system 0's panel workspace; pws_m[m] for the other systems of a batch (own gather buffers,
  LuPanelWs pws_m[LU_BATCH_MAX]{};  // so that two systems' panel kernels may be in flight together when the chip holds both)
  // per system of a batch: pivots, the folded interchange lists, and 2*NB rows x (n + nrhs_max) staging for the interchanges
  int* d_lists[LU_BATCH_MAX] = {};
  int* d_ipiv[LU_BATCH_MAX] = {};
  c64* d_tmp[LU_BATCH_MAX] = {};
  c64* d_invd[LU_BATCH_MAX] = {};  // inverted 32 x 32 diagonal blocks of L11, one slot per panel of a block; lists and these are written by the
                                   // look-ahead lane (block g+1 -> slots of parity (g+1)&1) and read by the main lane (block g)
  // the look-ahead lane's own interchange staging (it works on block g+1 while the main lane works on block g)
  c64* d_tmp_l[LU_BATCH_MAX] = {};
  int kb = 4;                     // panels per trailing update (MA_LU_KB=1..8); without the switch: as many as make K = kb * nb = 256
  bool kb_env = false;
  double gemm_flops = 0.0;        // algorithmic flops of the update launches of the last call
  double gemm_cbytes = 0.0;       // and their algorithmic C read + write bytes
  int last_batch = 1;
  // staged (pipelined) use: the system currently in each slot, and the event that says its A/b are ready
  c64* cur_A[LU_BATCH_MAX] = {}; c64* cur_B[LU_BATCH_MAX] = {}; int cur_nrhs = 0;
  hipEvent_t ev_prep[LU_BATCH_MAX] = {};
  int stage_first_mark = -1;
  // deferred finish (ma_lu_plan_stage_finish_defer / _issue / _wait): the system a slot has factored, whose backward substitution is
  // issued later -- behind the first block columns of the slot's NEXT system, where the lane has slack -- and waited for later still
  c64* fin_A[LU_BATCH_MAX] = {}; c64* fin_B[LU_BATCH_MAX] = {}; int fin_nrhs[LU_BATCH_MAX] = {}; int fin_state[LU_BATCH_MAX] = {};   // 0 none, 1 deferred, 2 issued
  hipEvent_t ev_fin[LU_BATCH_MAX] = {};
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
  hipStream_t panel_stream = nullptr;   // stream of the look-ahead lane (system 0)
  hipStream_t panel_streams[LU_BATCH_MAX] = {};   // [0] aliases panel_stream; one per system of a batch
  bool panel_overlap = true;      // MA_LU_PANEL_OVERLAP=0: all systems' panels on one stream (strictly serial)
  hipStream_t mid_streams[LU_BATCH_MAX] = {};     // per system: the small per-panel work of the current block (MA_LU_MIDLANE=0: on the caller's stream)
  int midlane = 1;                // 1: on the system's look-ahead stream, 2: on a third stream per system, 0: on the caller's stream
  bool lookahead = true;          // factor panel q+1 on a second stream under panel q's trailing update (MA_LU_LOOKAHEAD=0 disables)
  int want_nb = 64;               // panel width (MA_LU_NB): 64 columns keep two systems' panels co-resident from the first column of a 10k system
  int lane_alias = 0;             // MA_LU_LANE_ALIAS=<L> (experiment): slots m and m + L share ONE lane stream (m mod L), so that six systems run on the three
                                  // lanes' hardware queues -- an update-bound system (early blocks) and a chain-bound one (late blocks) per lane
  bool block_step = false;        // round 4: the main lane's per-panel launches of a block (12 gathers / scatters, 6 trsm, 6 zgemv, 5 in-block updates) as
                                  // lu_block_row_moves_kernel + lu_block_trsm_kernel + one zgemv (MA_LU_BLOCK_STEP; default with the register pair panels)
  ZgemmMode zmode;                // the update kernel family, resolved from the MA_ZGEMM_* switches when the plan is created
  bool use_3m = true;             // 3-product complex zgemm in the trailing update (MA_ZGEMM_3M=0 selects the 4-product form)
  bool rpb_env = false;           // MA_LU_RPB given
  int rpb_cap = 44;               // rows per panel workgroup when MA_LU_RPB is given
  // lock-step batches: ONE panel kernel factors the same panel of every system of the batch (lu_panel_batch_kernel); panels of
  // batch_nb columns, batch_lds bytes of LDS per workgroup over all systems (off by default: MA_LU_BATCH_PANEL=1)
  bool batch_panel = false; int batch_nb = 32; int batch_lds = 56 * 1024;   // measured: the staged pipeline of per-system panel kernels is faster; MA_LU_BATCH_PANEL=1 selects this form
  hipEvent_t ev_bp = nullptr, ev_lane[LU_BATCH_MAX] = {};
  int last_bp_nsys = 0;           // > 0: the last factorisation used batched panels over that many systems (its panel partition differs)
  // staged use with GROUPS: slots [k group, (k+1) group) move in lock step and share one panel kernel per panel (a wavefront per
  // system, lu_panel_wave_kernel); different groups sit at different block indices, so one group's latency-bound chain runs under
  // the other groups' trailing updates. 0 / 1: every slot on its own (the round-1 pipeline).
  int stage_group = 0;
  bool stage_lane_pending[LU_BATCH_MAX] = {};
  // round 3 (the default for MA_LU_PAIR_MIN_N .. MA_LU_PAIR_MAX_N rows, see the top of this file): panels by lu_panel_reg_kernel
  // (rows in registers, 32 columns, 256 rows per workgroup; MA_LU_REG_PANEL=0: the LDS-resident lu_panel_kernel) and the chip
  // split in two sets of CUs (MA_LU_CU_SPLIT=<P>, 0 = off): the big trailing updates run on a stream masked to ncu - P CUs, so that
  // the panel kernels' workgroups find the other P (P / 8 per XCD) free of update workgroups -- a panel kernel's exchange runs at
  // its idle round trip there instead of 2-3 x that beside update workgroups on the same CU (profiles/r03_cumask_probe.txt).
  // With MA_LU_PAN_MASK=1 the panel kernels run on streams masked to those P CUs themselves (measured: the extra hardware queues
  // cost far more than the guarantee buys).
  bool reg_panel0 = false, reg_pair0 = false;             // what the plan was created with (slot groups switch a plan to the LDS family and back)
  bool reg_panel = false;
  bool reg_pair = false;                                  // MA_LU_REG_PANEL=2: the 64-column structure of round 2 (K = 64 in-block updates, 4 panels per block, the main lane's
                                                          // per-panel work on 64 columns) with each 64-column panel factored as TWO register half-panels and the step between them
  int* d_half_lists[LU_BATCH_MAX] = {}; c64* d_half_invd[LU_BATCH_MAX] = {}; c64* d_half_l10[LU_BATCH_MAX] = {};
  int cu_split = 0;
  int chain_mask = 0;                                     // MA_LU_CHAIN_MASK=1: the per-panel chain launches on streams masked to the update CUs too
  int pan_mask = 0;                                       // MA_LU_PAN_MASK=1: the panel kernels on streams of their own, masked to the P panel CUs (0: only the big updates are masked
                                                          // away from those CUs; every masked stream is one more hardware queue, and more than 4-5 busy queues cost more than they buy)
  hipStream_t pan_streams[LU_BATCH_MAX] = {};             // mask A: panel kernels of slot m
  hipStream_t chain_streams[LU_BATCH_MAX] = {};           // mask B (chain_mask) -- otherwise panel_streams[m] carries the chain
  hipStream_t big_stream = nullptr;                       // mask B: the K = 256 updates of all slots
  hipEvent_t ev_pan[LU_BATCH_MAX] = {}, ev_chain[LU_BATCH_MAX] = {};
  int panel_cus() const { return cu_split > 0 ? cu_split : ncu; }
  int share_pct = 0, share_min_rows = 0;                  // staged schedule: blocks with at least share_min_rows rows left give share_pct % of their big update's columns to the slot's lane (see Stage::lane_share)
  int tail_rows = 0;                                      // staged schedule: blocks with at most this many rows left take their WHOLE trailing update on the slot's lane (see Stage::tail)
  int admit_cus = 0;                                      // MA_LU_ADMIT_CUS: the CU count the admission window counts register panels against (0: what the launch may use)
  // round 5: MA_LU_PIVOT_TOURNAMENT -- the half-panels by lu_launch_panel_calu (lu_calu.hip: one tournament per 32 columns instead of one
  // chip-wide exchange per column; no workgroup waits for another). The pivots differ from LAPACK's, the solution does not (to
  // rounding): the mode of the sweep, where lu_solve's contract (x only, lu.rs:142-153) is the boundary. MA_LU_PIVOT_PARTIAL elsewhere.
  int pivoting = MA_LU_PIVOT_PARTIAL;
  LuCaluWs calu[LU_BATCH_MAX]{};
  // the speculative panel (lu_spec.hip) ahead of every half-panel of the pair structure, in either pivoting mode: accepted, it IS the
  // partial-pivoting panel (verified); rejected, the mode's own panel kernel runs behind it. MA_LU_SPECULATE=0 switches it off.
  bool speculate = false;
  bool optimistic = false;                                // MA_LU_SPECULATE_OPTIMISTIC (ma_lu_plan_set_speculation): no fallback behind the speculative panel; a rejected one
                                                          // leaves -1 in the system's status word (MA_ERR_RETRY) and the CALLER solves that system again in the verified mode
  LuSpecWs spec[LU_BATCH_MAX]{};
  unsigned long long* d_spec_stats = nullptr;
};

static void panel_schedule(const ma_lu_plan* P, std::vector<int>& k0s, std::vector<int>& nbs, std::vector<int>& rpbs, std::vector<int>& nblks);

namespace {

// Panel geometry: widest panel (128/64/32/16) whose rows fit the co-resident workgroups' LDS.
void panel_shape(int R, int ncu, int want_nb, int rpb_cap, int* nb_out, int* rpb_out, int* nblk_out) {
  const int widths[4] = {128, 64, 32, 16};
  for (int w = 0; w < 4; ++w) {
    int nb = widths[w];
    int rpb_max = (int)((150000 - 2 * nb * 16) / ((nb + 1) * 16));
    if (rpb_max > 256) rpb_max = 256;
    long long cap = (long long)rpb_max * ncu;
    if (cap >= R || w == 3) {
      int rpb = rpb_max < rpb_cap ? rpb_max : rpb_cap; // few rows per workgroup: small LDS footprint, short local update
      int nblk = (R + rpb - 1) / rpb;
      if (nblk > ncu) { rpb = (R + ncu - 1) / ncu; nblk = (R + rpb - 1) / rpb; }
      if (nb > want_nb) nb = want_nb;
      *nb_out = nb; *rpb_out = rpb; *nblk_out = nblk;
      return;
    }
  }
}

}  // namespace

int ma_lu_plan::ensure_batch(int nmat) {
  for (int m = 0; m < nmat; ++m) {
    if (d_tmp[m]) continue;
    MA_HIP(hipMalloc(&d_tmp[m], sizeof(c64) * 2 * LU_NB_MAX * ((size_t)n + nrhs_max)));
    MA_HIP(hipMalloc(&d_ipiv[m], sizeof(int) * (size_t)n));
    MA_HIP(hipMemset(d_ipiv[m], 0, sizeof(int) * (size_t)n));
    MA_HIP(hipMalloc(&d_lists[m], sizeof(int) * 2 * LU_KB_MAX * LU_LISTS_LEN));
    MA_HIP(hipMalloc(&d_invd[m], sizeof(c64) * 2 * LU_KB_MAX * LU_NB_MAX * 32));
    MA_HIP(hipMalloc(&d_tmp_l[m], sizeof(c64) * 2 * LU_NB_MAX * LU_LANE_TSTRIDE));
    MA_HIP(hipMalloc(&d_half_lists[m], sizeof(int) * 2 * LU_LISTS_LEN));
    MA_HIP(hipMemset(d_half_lists[m], 0, sizeof(int) * 2 * LU_LISTS_LEN));
    MA_HIP(hipMalloc(&d_half_invd[m], sizeof(c64) * 32 * 32));
    MA_HIP(hipMalloc(&d_half_l10[m], sizeof(c64) * 32 * 32));
    MA_HIP(hipMemset(d_half_l10[m], 0, sizeof(c64) * 32 * 32));
    if (speculate) {
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
    // the memset above runs on the null stream; the plan's lanes and the callers' streams may be non-blocking streams that do
    // not order themselves against it: without this wait it can land AFTER a panel kernel has written its pivots (seen as
    // "pivot outside its range" under two host threads)
    MA_HIP(hipStreamSynchronize(nullptr));
  }
  return MA_OK;
}

extern "C" {

int ma_lu_plan_create(int32_t n, int device, ma_lu_plan_t** out) {
  // MA_LU_PIVOTING=tournament|partial: the mode of plans made through this entry (the drop-in entries of lu.rs: partial)
  int mode = MA_LU_PIVOT_PARTIAL;
  if (const char* e = getenv("MA_LU_PIVOTING")) mode = (e[0] == 't' || e[0] == 'T' || e[0] == '1') ? MA_LU_PIVOT_TOURNAMENT : MA_LU_PIVOT_PARTIAL;
  return ma_lu_plan_create_pivoting(n, device, mode, out);
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
  MA_REQUIRE((long long)n <= 256LL * ncu, MA_ERR_UNSUPPORTED, "n = %d exceeds the co-resident panel capacity (%d rows)", n, 256 * ncu);
  ma_lu_plan* P = new (std::nothrow) ma_lu_plan();
  MA_REQUIRE(P, MA_ERR_NOMEM, "host allocation failed");
  P->device = device; P->n = n; P->ncu = ncu; P->pivoting = pivoting;
  // (decided before the first ensure_batch: that is where a slot's speculative-panel workspace is made)
  {
    const bool lds_tuned0 = getenv("MA_LU_RPB") || getenv("MA_LU_NB") || (getenv("MA_LU_BATCH_PANEL") && atoi(getenv("MA_LU_BATCH_PANEL")) != 0);
    bool pair = pivoting == MA_LU_PIVOT_TOURNAMENT || (n >= MA_LU_PAIR_MIN_N && n <= MA_LU_PAIR_MAX_N && ncu == 256 && !lds_tuned0);
    if (pivoting != MA_LU_PIVOT_TOURNAMENT) if (const char* er = getenv("MA_LU_REG_PANEL")) pair = atoi(er) == 2;
    P->speculate = pair;
    if (const char* es = getenv("MA_LU_SPECULATE")) P->speculate = pair && atoi(es) != 0;
    if (P->speculate) {
      if (hipMalloc(&P->d_spec_stats, 64) != hipSuccess || hipMemset(P->d_spec_stats, 0, 64) != hipSuccess) { set_error("hipMalloc of the LU plan's counters failed"); delete P; return MA_ERR_NOMEM; }
    }
  }
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
    if (P->ws_block) (void)hipFree(P->ws_block);
    delete P;
    return MA_ERR_NOMEM;
  }
  if ((rc = P->ensure_batch(1))) { (void)hipFree(P->ws_block); delete P; return rc; }
  char* base = (char*)P->ws_block;
  P->pws.counter = (unsigned*)(base + o_sync);
  P->pws.info = (int*)(base + o_info);
  P->pws.timeout = (unsigned*)(P->pws.info + LU_BATCH_MAX);   // persists over the factorisation, like info
  P->pws.max_blocks = mb;
  for (int m = 0; m < LU_BATCH_MAX; ++m) {
    P->pws_m[m] = P->pws;
    P->pws_m[m].info = P->pws.info + m;                       // per-system first-zero-pivot word
    P->pws_m[m].cand = (unsigned long long*)(base + o_cand[m]);
    P->pws_m[m].candrow = (unsigned long long*)(base + o_crow[m]);
    P->pws_m[m].diagrow = (unsigned long long*)(base + o_drow[m]);
  }
  P->pws = P->pws_m[0];
  rc = lu_panel_configure();
  if (!rc) rc = lu_trsm_configure();
  if (const char* e1 = getenv("MA_LU_NB")) { int v = atoi(e1); if (v >= 16 && v <= LU_NB_MAX && v % 16 == 0) P->want_nb = v; }
  if (const char* e0 = getenv("MA_ZGEMM_3M")) P->use_3m = atoi(e0) != 0;
  P->zmode = zgemm_mode_from_env();
  if (const char* e2 = getenv("MA_LU_LOOKAHEAD")) P->lookahead = atoi(e2) != 0;
  if (const char* e6 = getenv("MA_LU_KB")) { int v = atoi(e6); if (v >= 1 && v <= LU_KB_MAX) { P->kb = v; P->kb_env = true; } }
  if (const char* e5 = getenv("MA_LU_PANEL_OVERLAP")) P->panel_overlap = atoi(e5) != 0;
  if (const char* e7 = getenv("MA_LU_MIDLANE")) P->midlane = atoi(e7);
  if (const char* eb = getenv("MA_LU_BATCH_PANEL")) P->batch_panel = atoi(eb) != 0;
  if (const char* eb = getenv("MA_LU_BATCH_NB")) { int v = atoi(eb); if (v >= 16 && v <= LU_NB_MAX && v % 16 == 0) P->batch_nb = v; }
  if (const char* eb = getenv("MA_LU_BATCH_LDS")) { int v = atoi(eb); if (v >= 16 && v <= 150) P->batch_lds = v * 1024; }
  if (const char* e3 = getenv("MA_LU_RPB")) { int v = atoi(e3); if (v >= 8 && v <= 256) { P->rpb_cap = v; P->rpb_env = true; } }
  // the register panel kernel when the tallest panel's workgroups (256 rows each) are co-resident on the CUs its stream may use
  {
    const bool lds_tuned = P->rpb_env || getenv("MA_LU_NB") || P->batch_panel;
    const bool tour = pivoting == MA_LU_PIVOT_TOURNAMENT;   // tournament panels: the pair structure at every size (nothing has to be co-resident)
    int want_reg = (tour || (n >= MA_LU_PAIR_MIN_N && n <= MA_LU_PAIR_MAX_N && ncu == 256 && !lds_tuned)) ? 2 : 0;
    if (const char* er = getenv("MA_LU_REG_PANEL")) { if (!tour) want_reg = atoi(er); }
    int split = want_reg == 2 ? (tour ? MA_LU_CU_SPLIT_TOURNAMENT : MA_LU_CU_SPLIT_DEFAULT) : 0;
    if (tour && !(n >= MA_LU_PAIR_MIN_N && n <= MA_LU_PAIR_MAX_N && ncu == 256)) split = 0;
    if (const char* es = getenv("MA_LU_CU_SPLIT")) split = atoi(es);
    if (const char* ec = getenv("MA_LU_CHAIN_MASK")) P->chain_mask = atoi(ec) != 0;
    if (const char* ep = getenv("MA_LU_PAN_MASK")) P->pan_mask = atoi(ep) != 0;
    if (const char* ea = getenv("MA_LU_ADMIT_CUS")) { const int v = atoi(ea); if (v >= 20 && v <= ncu) P->admit_cus = v; }
    if (const char* et = getenv("MA_LU_TAIL_ROWS")) { const int v = atoi(et); if (v >= 0) P->tail_rows = v; }
    if (const char* et = getenv("MA_LU_LANE_SHARE")) { const int v = atoi(et); if (v >= 0 && v <= 90) P->share_pct = v; }
    if (const char* et = getenv("MA_LU_LANE_SHARE_MIN_ROWS")) { const int v = atoi(et); if (v >= 0) P->share_min_rows = v; }
    if (split < 8 || split % 8 != 0 || split > ncu - 64 || ncu % 32 != 0 || !P->lookahead || !P->panel_overlap) split = 0;
    P->cu_split = split;
    if (P->batch_panel && !tour) want_reg = 0;             // the shared (wavefront-per-system) panel kernel is the LDS family: a plan stays in one family
    if (tour) { P->batch_panel = false; P->pan_mask = 0; P->reg_panel = true; P->reg_pair = true; }
    else if (!rc && want_reg && n <= 65535) {
      const int nblk0 = (n + 255) / 256;
      if (nblk0 <= P->pws.max_blocks && lu_panel_reg_admissible(nblk0, (split && P->pan_mask) ? split : ncu) == MA_OK) { P->reg_panel = true; P->reg_pair = want_reg == 2; }
      else if (split && P->pan_mask && nblk0 <= P->pws.max_blocks && lu_panel_reg_admissible(nblk0, ncu) == MA_OK) { P->pan_mask = 0; P->reg_panel = true; P->reg_pair = want_reg == 2; }   // too tall for the panel CUs: panels anywhere
    }
    if (!P->reg_panel) P->pan_mask = 0;                   // the LDS-resident panel kernel's grid does not fit a small CU set: only the big updates are masked
    if (!P->reg_panel && !getenv("MA_LU_CU_SPLIT")) P->cu_split = 0;   // the default split comes with the register panels only
    P->reg_panel0 = P->reg_panel; P->reg_pair0 = P->reg_pair;
    if (!(P->reg_panel && P->reg_pair)) P->speculate = false;   // (the pair structure was refused after all: the workspace stays unused)
    // measured (profiles/r04_lu_schedule_experiments.md): three launches per block instead of 31 shorten every slot's chain (the stream's
    // waits 4.3 -> 3.8 ms per frequency) but the fused kernel's 157 four-wavefront workgroups cost the big updates what the chain gains
    // (49.5 against 49.3 ms per frequency; 51.8 against 48.6 with eight panels per block): built, tested, off by default
    P->block_step = false;
    if (const char* eb = getenv("MA_LU_BLOCK_STEP")) P->block_step = atoi(eb) != 0 && P->reg_panel && P->reg_pair;
    if (const char* el = getenv("MA_LU_LANE_ALIAS")) { const int v = atoi(el); if (v >= 1 && v < LU_BATCH_MAX) P->lane_alias = v; }
  }
  for (int m = 0; m < LU_BATCH_MAX; ++m) { P->pws_m[m].test_abort_col = -1; P->pws_m[m].diag_sleep = 0; }
  if (const char* ed = getenv("MA_DIAG_PANEL_SLEEP")) for (int m = 0; m < LU_BATCH_MAX; ++m) P->pws_m[m].diag_sleep = std::max(0, std::min(64, atoi(ed)));
  if (const char* e9 = getenv("MA_LU_TEST_ABORT_COL")) for (int m = 0; m < LU_BATCH_MAX; ++m) P->pws_m[m].test_abort_col = atoi(e9);
  P->pws = P->pws_m[0];
  if (!rc) {
    // every panel shape of this plan's schedule must be co-resident on its own (lu_kernels.hip, "Residency"): a tuning switch
    // that asks for more LDS per workgroup than the chip can hold at the grid's size is refused here, not at the first launch
    std::vector<int> k0s, nbs, rpbs, nblks;
    panel_schedule(P, k0s, nbs, rpbs, nblks);
    for (size_t q = 0; q < k0s.size() && !rc && !P->reg_panel; ++q)
      if (q == 0 || rpbs[q] != rpbs[q - 1] || nbs[q] != nbs[q - 1] || nblks[q] > nblks[q - 1]) rc = lu_panel_admissible(nbs[q], rpbs[q], nblks[q], P->ncu);
    if (rc && (P->rpb_env || getenv("MA_LU_NB"))) rc = MA_ERR_INVALID;      // the text of the refusal is already in the error string
  }
  if (!rc) {
    int lo = 0, hi = 0;
    hipError_t e4 = hipDeviceGetStreamPriorityRange(&lo, &hi);
    // The look-ahead lanes run at the caller's (normal) priority: measured equal to the highest priority with up to three
    // systems in flight, and with four the high-priority lanes starved the main lane's small launches, on which every
    // lane waits (MA_LU_LANE_PRIO=1 selects the highest priority, -1 the lowest).
    { int v = 0; if (const char* e8 = getenv("MA_LU_LANE_PRIO")) v = atoi(e8); if (v == 0) hi = 0; else if (v < 0) hi = lo; }
    if (e4 == hipSuccess) e4 = hipStreamCreateWithPriority(&P->panel_stream, hipStreamNonBlocking, hi);
    P->panel_streams[0] = P->panel_stream;
    for (int i = 1; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipStreamCreateWithPriority(&P->panel_streams[i], hipStreamNonBlocking, hi);
    if (e4 == hipSuccess) e4 = hipEventCreateWithFlags(&P->ev_start, hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_panel[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_narrow[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipStreamCreateWithPriority(&P->mid_streams[i], hipStreamNonBlocking, hi);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_mid[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_big[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_prep[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_lane[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_fin[i], hipEventDisableTiming);
    if (e4 == hipSuccess) e4 = hipEventCreateWithFlags(&P->ev_bp, hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_pan[i], hipEventDisableTiming);
    for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess; ++i) e4 = hipEventCreateWithFlags(&P->ev_chain[i], hipEventDisableTiming);
    if (e4 == hipSuccess && P->cu_split) {
      // CU masks: bit i of the mask is CU (i / 8 mod 4 SEs ...) of XCD i mod 8 (tools/cumask_probe.hip): bits [0, P) are P / 8 CUs of
      // every XCD. Masked streams are blocking streams (the only kind hipExtStreamCreateWithCUMask makes): a caller that drives the
      // staged schedule from the NULL stream serialises against them -- use a non-blocking stream (bench.py, ma_bem_solve_sweep do)
      const int words = ncu / 32;
      std::vector<uint32_t> mA(words, 0u), mB(words, 0u);
      // MA_LU_SPLIT_SHAPE=xcd: the panel CUs as WHOLE XCDs (split / 32 of them) instead of split / 8 CUs of every XCD
      const char* shp = getenv("MA_LU_SPLIT_SHAPE");
      const bool by_xcd = shp && shp[0] == 'x' && P->cu_split % 32 == 0;
      for (int i = 0; i < ncu; ++i) ((by_xcd ? (i % 8) < P->cu_split / 32 : i < P->cu_split) ? mA : mB)[i / 32] |= 1u << (i % 32);
      for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess && P->pan_mask; ++i) e4 = hipExtStreamCreateWithCUMask(&P->pan_streams[i], words, mA.data());
      for (int i = 0; i < LU_BATCH_MAX && e4 == hipSuccess && P->chain_mask; ++i) e4 = hipExtStreamCreateWithCUMask(&P->chain_streams[i], words, mB.data());
      if (e4 == hipSuccess) {
        e4 = hipExtStreamCreateWithCUMask(&P->big_stream, words, mB.data());
        if (e4 != hipSuccess && !getenv("MA_LU_CU_SPLIT")) {   // the default split on a runtime that makes no masked streams: the whole chip for everything
          (void)hipGetLastError(); e4 = hipSuccess; P->big_stream = nullptr; P->cu_split = 0;
        }
        // the mask's bit layout (bit i = XCD i mod 8, ...) is what tools/cumask_probe.hip found on an SPX MI355X; a census on the new
        // stream says whether THIS device agrees: ncu - split CUs in use, the same number in every XCD. If not (another partition
        // mode, another part), the plan runs its updates on the whole chip -- the round-2 placement, slower, never wrong.
        if (e4 == hipSuccess && P->big_stream && !by_xcd) {
          bool mask_ok = false;
          rc = lu_cumask_selfcheck(P->big_stream, ncu - P->cu_split, &mask_ok);
          if (!rc && !mask_ok) {
            (void)hipStreamDestroy(P->big_stream); P->big_stream = nullptr; P->cu_split = 0;
            for (int i = 0; i < LU_BATCH_MAX; ++i) { if (P->pan_streams[i]) { (void)hipStreamDestroy(P->pan_streams[i]); P->pan_streams[i] = nullptr; }
                                                     if (P->chain_streams[i]) { (void)hipStreamDestroy(P->chain_streams[i]); P->chain_streams[i] = nullptr; } }
            P->pan_mask = 0; P->chain_mask = 0;
          }
        }
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
  for (int i = 0; i < LU_BATCH_MAX; ++i) { if (P->panel_streams[i]) (void)hipStreamSynchronize(P->panel_streams[i]); if (P->mid_streams[i]) (void)hipStreamSynchronize(P->mid_streams[i]); }
  for (int i = 0; i < LU_BATCH_MAX; ++i) { if (P->pan_streams[i]) (void)hipStreamSynchronize(P->pan_streams[i]); if (P->chain_streams[i]) (void)hipStreamSynchronize(P->chain_streams[i]); }
  if (P->big_stream) (void)hipStreamSynchronize(P->big_stream);
  for (int i = 0; i < LU_BATCH_MAX; ++i) { lu_panel_forget_stream(P->device, P->panel_streams[i]); lu_panel_forget_stream(P->device, P->mid_streams[i]); lu_panel_forget_stream(P->device, P->pan_streams[i]); }
  lu_panel_forget_stream(P->device, P->panel_stream);
  for (hipEvent_t e : P->ev) (void)hipEventDestroy(e);
  if (P->ev_start) (void)hipEventDestroy(P->ev_start);
  if (P->ev_bp) (void)hipEventDestroy(P->ev_bp);
  for (int i = 0; i < LU_BATCH_MAX; ++i) if (P->ev_lane[i]) (void)hipEventDestroy(P->ev_lane[i]);
  for (int i = 0; i < LU_BATCH_MAX; ++i) if (P->ev_fin[i]) (void)hipEventDestroy(P->ev_fin[i]);
  for (int i = 0; i < LU_BATCH_MAX; ++i) { if (P->ev_panel[i]) (void)hipEventDestroy(P->ev_panel[i]); if (P->ev_narrow[i]) (void)hipEventDestroy(P->ev_narrow[i]);
    if (P->ev_prep[i]) (void)hipEventDestroy(P->ev_prep[i]); if (P->ev_mid[i]) (void)hipEventDestroy(P->ev_mid[i]); if (P->ev_big[i]) (void)hipEventDestroy(P->ev_big[i]); if (P->mid_streams[i]) (void)hipStreamDestroy(P->mid_streams[i]); }
  for (int i = 0; i < LU_BATCH_MAX; ++i) if (P->panel_streams[i]) (void)hipStreamDestroy(P->panel_streams[i]);
  for (int i = 0; i < LU_BATCH_MAX; ++i) { if (P->pan_streams[i]) (void)hipStreamDestroy(P->pan_streams[i]); if (P->chain_streams[i]) (void)hipStreamDestroy(P->chain_streams[i]);
    if (P->ev_pan[i]) (void)hipEventDestroy(P->ev_pan[i]); if (P->ev_chain[i]) (void)hipEventDestroy(P->ev_chain[i]); }
  if (P->big_stream) (void)hipStreamDestroy(P->big_stream);
  for (int i = 0; i < LU_BATCH_MAX; ++i) { if (P->d_tmp[i]) (void)hipFree(P->d_tmp[i]); if (P->d_ipiv[i]) (void)hipFree(P->d_ipiv[i]); if (P->d_lists[i]) (void)hipFree(P->d_lists[i]); if (P->d_invd[i]) (void)hipFree(P->d_invd[i]);
    if (P->calu[i].cand) (void)hipFree(P->calu[i].cand); if (P->calu[i].counters) (void)hipFree(P->calu[i].counters);
    { LuSpecWs& w = P->spec[i]; if (w.u11) (void)hipFree(w.u11); if (w.rinv) (void)hipFree(w.rinv); if (w.pivmag) (void)hipFree(w.pivmag); if (w.ctl) (void)hipFree(w.ctl);
      if (w.vlist) (void)hipFree(w.vlist); if (w.ext) (void)hipFree(w.ext); if (w.backup) (void)hipFree(w.backup); }
    if (P->d_tmp_l[i]) (void)hipFree(P->d_tmp_l[i]); if (P->d_half_lists[i]) (void)hipFree(P->d_half_lists[i]); if (P->d_half_invd[i]) (void)hipFree(P->d_half_invd[i]); if (P->d_half_l10[i]) (void)hipFree(P->d_half_l10[i]); }
  if (P->ws_block) (void)hipFree(P->ws_block);
  if (P->d_spec_stats) (void)hipFree(P->d_spec_stats);
  delete P;
  return MA_OK;
}

// 0: no speculation; 1: verified with the fallback in line (what a plan of the pair structure starts with); 2: optimistic
int ma_lu_plan_set_speculation(ma_lu_plan_t* P, int32_t mode) {
  MA_REQUIRE(P && mode >= MA_LU_SPECULATE_OFF && mode <= MA_LU_SPECULATE_OPTIMISTIC, MA_ERR_INVALID, "speculation mode %d", mode);
  MA_REQUIRE(mode == MA_LU_SPECULATE_OFF || P->d_spec_stats, MA_ERR_UNSUPPORTED, "this plan does not factor in half-panel pairs: no speculative panel");
  P->speculate = mode != MA_LU_SPECULATE_OFF && P->reg_panel && P->reg_pair;
  P->optimistic = P->speculate && mode == MA_LU_SPECULATE_OPTIMISTIC;
  return MA_OK;
}
int ma_lu_plan_speculation(ma_lu_plan_t* P, int32_t* mode) {
  MA_REQUIRE(P && mode, MA_ERR_INVALID, "NULL argument");
  *mode = !P->speculate ? MA_LU_SPECULATE_OFF : (P->optimistic ? MA_LU_SPECULATE_OPTIMISTIC : MA_LU_SPECULATE_VERIFIED);
  return MA_OK;
}

// half-panels the speculative panel factored at the first attempt / at the widened attempt / handed to the plan's own panel kernel or
// marked for another solve, since the plan was made; synchronises the device
int ma_lu_plan_speculation_stats(ma_lu_plan_t* P, int64_t* accepted, int64_t* accepted_widened, int64_t* rejected) {
  MA_REQUIRE(P && accepted && accepted_widened && rejected, MA_ERR_INVALID, "NULL argument");
  *accepted = 0; *accepted_widened = 0; *rejected = 0;
  if (!P->d_spec_stats) return MA_OK;
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

// panels of the factorisation: first column, width, rows per panel workgroup, workgroups
static void panel_schedule(const ma_lu_plan* P, std::vector<int>& k0s, std::vector<int>& nbs, std::vector<int>& rpbs, std::vector<int>& nblks) {
  const int n = P->n;
  if (P->reg_panel && P->reg_pair) {                      // 64-column panels, each factored as two register half-panels (launch_panel_pair)
    for (int k0 = 0; k0 < n; k0 += 2 * LU_REG_NB) { k0s.push_back(k0); nbs.push_back(std::min(n - k0, 2 * LU_REG_NB)); rpbs.push_back(256); nblks.push_back((n - k0 + 255) / 256); }
    return;
  }
  if (P->reg_panel) {                                     // rows in registers: 256 rows per workgroup, LU_REG_NB columns per panel
    for (int k0 = 0; k0 < n; k0 += LU_REG_NB) { k0s.push_back(k0); nbs.push_back(std::min(n - k0, LU_REG_NB)); rpbs.push_back(256); nblks.push_back((n - k0 + 255) / 256); }
    return;
  }
  for (int k0 = 0; k0 < n;) {
    int nb, rpb, nblk;
    // rows per panel workgroup: at most 47.5 KB of LDS, so that two systems' panel workgroups AND two trailing-update
    // workgroups fit on a CU together (the launcher admits panel kernels up to 96 KB per CU): 43 rows at nb = 64.
    // Few rows per workgroup also keep the per-column local work -- which a co-tenant update slows down -- short.
    const int wnb = std::min(n - k0, P->want_nb);
    const int cap = P->rpb_env ? P->rpb_cap : std::max(8, (int)((48640 - 3 * wnb * 16 - 256) / ((wnb + 1) * 16)));
    panel_shape(n - k0, P->ncu, std::min(n - k0, P->want_nb), cap, &nb, &rpb, &nblk);
    k0s.push_back(k0); nbs.push_back(nb); rpbs.push_back(rpb); nblks.push_back(nblk);
    k0 += nb;
  }
}

// the same for a batch whose panels are factored by ONE kernel for all `nsys` systems: narrower panels, and rows per workgroup such
// that the workgroup's LDS over all systems stays within batch_lds (two trailing-update workgroups still fit beside it)
static void panel_schedule_batched(const ma_lu_plan* P, int nsys, std::vector<int>& k0s, std::vector<int>& nbs, std::vector<int>& rpbs, std::vector<int>& nblks) {
  const int n = P->n;
  for (int k0 = 0; k0 < n;) {
    const int nb = std::min(n - k0, P->batch_nb);
    int rpb = P->rpb_env ? P->rpb_cap : std::min(64, std::max(8, (int)((P->batch_lds / nsys - 3 * nb * 16 - 256) / ((nb + 1) * 16))));
    int nblk = (n - k0 + rpb - 1) / rpb;
    if (nblk > P->ncu) { rpb = (n - k0 + P->ncu - 1) / P->ncu; nblk = (n - k0 + rpb - 1) / rpb; }
    k0s.push_back(k0); nbs.push_back(nb); rpbs.push_back(rpb); nblks.push_back(nblk);
    k0 += nb;
  }
}

// one panel of system (A, ws, ipiv) on stream st: the kernel the plan's schedule was made for
static int launch_panel(ma_lu_plan* P, int m, c64* A, int k0, int nb, int rpb, int nblk, const LuPanelWs& ws, int* ipiv, int* lists, bool clear_tags, hipStream_t st, bool masked) {
  if (P->reg_panel && P->reg_pair) {
    // a 64-column panel as two register half-panels: left half; its interchanges + U12 + rank-32 update on the right half's columns
    // (lu_lane_step_kernel + one K = 32 update); right half. The pivots land in ipiv as one 64-column panel's: everything after
    // this (lu_perm_kernel on 64 pivots, interchanges, U12, K = 64 updates, the main lane) is the 64-column schedule
    const int n = P->n, cus = P->admit_cus > 0 ? P->admit_cus : (masked ? P->panel_cus() : P->ncu);
    const int h1 = std::min(nb, LU_REG_NB), h2 = nb - h1;
    const bool tour = P->pivoting == MA_LU_PIVOT_TOURNAMENT;
    const bool spec = P->speculate && P->spec[m].backup;
    const int* gate = spec ? P->spec[m].ctl : nullptr;     // the mode's own panel kernel runs only where the speculative one was rejected
    const bool opt = spec && P->optimistic;
    int rc = spec ? lu_launch_panel_spec(A, n, k0, h1, P->spec[m], ipiv, P->d_half_lists[m], st, nullptr, 0, opt, ws.info) : MA_OK;
    if (rc) return rc;
    if (opt) {                                               // nothing behind the speculative panels
      if (h2 <= 0) return MA_OK;
      const int a1o = k0 + h1;
      if ((rc = lu_launch_lane_step(A, n, k0, h1, P->d_half_lists[m], a1o, h2, P->d_half_invd[m], P->pws.timeout, st))) return rc;
      if ((rc = lu_launch_zgemm_sub(n - a1o, h2, h1, A + (size_t)a1o * n + k0, (size_t)n, A + (size_t)k0 * n + a1o, (size_t)n, A + (size_t)a1o * n + a1o, (size_t)n, st, P->use_3m, false, &P->zmode))) return rc;
      return lu_launch_panel_spec(A, n, a1o, h2, P->spec[m], ipiv, P->d_half_lists[m] + LU_LISTS_LEN, st, P->d_half_l10[m], k0, true, ws.info);
    }
    rc = tour ? lu_launch_panel_calu(A, n, k0, h1, P->calu[m], ws.info, ipiv, P->d_half_lists[m], st, nullptr, 0, gate)
              : lu_launch_panel_reg(A, n, k0, h1, nblk, cus, ws, ipiv, P->d_half_lists[m], clear_tags, st, nullptr, 0, gate);
    if (rc || h2 <= 0) return rc;
    const int a1 = k0 + h1;
    if ((rc = lu_launch_lane_step(A, n, k0, h1, P->d_half_lists[m], a1, h2, P->d_half_invd[m], P->pws.timeout, st))) return rc;
    static const int skip_k32 = [] { const char* e = getenv("MA_DIAG_SKIP_LANE_GEMM"); return e ? atoi(e) : 0; }();   // diagnostic only (wrong results)
    if (skip_k32 < 32 && (rc = lu_launch_zgemm_sub(n - a1, h2, h1, A + (size_t)a1 * n + k0, (size_t)n, A + (size_t)k0 * n + a1, (size_t)n, A + (size_t)a1 * n + a1, (size_t)n, st, P->use_3m, false, &P->zmode))) return rc;
    // (the right half's interchanges on the LEFT half's columns -- part of the interchange itself inside a 64-column panel kernel --
    // are the first job of lu_lane_step2_kernel, which every caller launches next)
    if (spec && (rc = lu_launch_panel_spec(A, n, a1, h2, P->spec[m], ipiv, P->d_half_lists[m] + LU_LISTS_LEN, st, P->d_half_l10[m], k0))) return rc;
    if (tour) return lu_launch_panel_calu(A, n, a1, h2, P->calu[m], ws.info, ipiv, P->d_half_lists[m] + LU_LISTS_LEN, st, P->d_half_l10[m], k0, gate);
    return lu_launch_panel_reg(A, n, a1, h2, (n - a1 + 255) / 256, cus, ws, ipiv, P->d_half_lists[m] + LU_LISTS_LEN, false, st, P->d_half_l10[m], k0, gate);
  }
  if (P->reg_panel) return lu_launch_panel_reg(A, P->n, k0, nb, nblk, masked ? P->panel_cus() : P->ncu, ws, ipiv, lists, clear_tags, st);
  return lu_launch_panel(A, P->n, k0, nb, rpb, nblk, P->ncu, ws, ipiv, clear_tags, st);
}

// panels per trailing update: K = kb * nb = 256 (tall systems factor in narrower panels -- 32 columns from 36 353 rows
// on -- and then take 8 of them per update: 65.8 -> 72.1 TFLOP/s on a 50 172-row system); the look-ahead lane's interchange
// staging holds (kb - 1) panels' columns
static int effective_kb(const ma_lu_plan* P, const std::vector<int>& nbs) {
  const int nb0 = nbs.empty() ? P->want_nb : std::max(1, nbs[0]);
  int nbmax = 1;
  for (int v : nbs) nbmax = std::max(nbmax, v);             // panels widen again once the remaining rows fit (32 -> 64 columns)
  // register pair panels: 6 panels (K = 384) per update -- the update kernel's prologue and C read-modify-write are a fifth of a
  // K = 256 tile's time, and the short chain of the pair panels leaves the lanes room for two more panels per block (51.4 ms per
  // frequency against 52.3 at 4 and 51.5 at 8: profiles/r03_lu_panel_experiments.md)
  int kb = P->kb_env ? P->kb : (P->reg_panel && P->reg_pair) ? 6 : std::max(P->kb, 256 / nb0);
  kb = std::max(1, std::min(kb, LU_KB_MAX));
  while (kb > 1 && (kb - 1) * nbmax > LU_LANE_TSTRIDE) --kb;
  return kb;
}

// The main lane's work on block [q0, q1) right of the block and on the right-hand sides, as three launches (round 4): every panel's
// interchanges on the columns left of it and right of the block, U12 of the whole block row with its in-block updates and the
// right-hand sides' forward substitution, and the right-hand sides' rows below the block. Only for panels of <= 64 columns.
static bool block_step_ok(const ma_lu_plan* P, const std::vector<int>& nbs, int q0, int q1) {
  if (!P->block_step || q1 - q0 > 8 || q1 <= q0) return false;
  for (int q = q0; q < q1; ++q) if (nbs[q] > 64) return false;
  return true;
}
static int block_main(ma_lu_plan* P, int m, c64* A, c64* B, int nrhs, int g, const std::vector<int>& k0s, const std::vector<int>& nbs, int q0, int q1, int e, hipStream_t sm) {
  const int n = P->n, np = q1 - q0, a0 = k0s[q0], nright = n - e;
  const int* lists = P->d_lists[m] + (size_t)((g & 1) * LU_KB_MAX) * LU_LISTS_LEN;
  const c64* invd = P->d_invd[m] + (size_t)((g & 1) * LU_KB_MAX) * LU_NB_MAX * 32;
  int rc = lu_launch_block_row_moves(A, n, lists, LU_LISTS_LEN, np, &k0s[q0], &nbs[q0], 0, k0s[q1 - 1], e, n, B, nrhs, P->pws.timeout, sm);
  if (rc) return rc;
  if ((rc = lu_launch_block_trsm(A, n, np, &k0s[q0], &nbs[q0], invd, LU_NB_MAX * 32, A + (size_t)a0 * n + e, (size_t)n, nright, nrhs ? B + a0 : nullptr, (size_t)n, nrhs, sm))) return rc;
  for (int r = 0; r < nrhs && nright > 0; ++r)
    if ((rc = lu_launch_zgemv_sub(nright, e - a0, A + (size_t)e * n + a0, (size_t)n, B + (size_t)r * n + a0, B + (size_t)r * n + e, sm))) return rc;
  return MA_OK;
}

// Factor the matrices in place and solve for nrhs right-hand sides each (d_B[nrhs][n]); everything asynchronous.
//
// Right-looking blocked LU on two levels. Pivoting works on panels of <= 128 columns (lu_panel_kernel: the width whose
// rows fit the LDS of the co-resident workgroups); the trailing matrix is updated once per BLOCK of `kb` panels (default
// 2 => K = 256), because the update kernel's cost per launch is the read-modify-write of C: at K = 128 it runs at 61
// TFLOP/s, at K = 256 at 71 (tools/zgemm_bench.hip).
//
//   look-ahead lane (own stream per system), block g+1 = panels p_0..p_{kb-1}, columns [a_0, e):
//     for j: panel(p_j);  if j < kb-1: interchanges of p_j -> columns [a_{j+1}, e);  U = L_jj^-1 A[p_j rows, a_{j+1}:e);
//            A[a_{j+1}:n, a_{j+1}:e) -= L[a_{j+1}:n, p_j] U          (a small right-looking LU of the block column)
//   main lane (caller's stream), block g, once its panels are done:
//     interchanges of every p_j -> columns [0, a_j) U [e, n) and the right-hand sides
//     for j: U_j = L_jj^-1 A[p_j rows, e:n)  (+ forward substitution of b's rows, riding in the same launch);
//            b[a_{j+1}:n) -= L b_j;   A[a_{j+1}:e, e:n) -= L[a_{j+1}:e, p_j] U_j
//     A[e:n, e:n) -= A[e:n, a_0:e) A[a_0:e, e:n): first the columns of block g+1 (then the look-ahead lane starts on
//     them, concurrently with ...) then the rest.
//
// The panel workgroups are latency-bound (one chip-wide gather per column) and sized to share a CU with update
// workgroups, so the matrix cores stay busy underneath them. A batch of independent systems of the same size
// (frequencies of a sweep) is interleaved block by block on the caller's stream, each system with its own look-ahead
// lane: one system's latency-bound chain hides under the others' throughput-bound work. Two systems' panel kernels
// run at the same time only when the chip holds both (the launcher's sequencer).
static int factor_solve_batch(ma_lu_plan* P, int nmat, c64* const* As, c64* const* Bs, int32_t nrhs, hipStream_t st) {
  hipStream_t sps[LU_BATCH_MAX];
  for (int m = 0; m < LU_BATCH_MAX; ++m) sps[m] = !P->lookahead ? st : (P->panel_overlap ? P->panel_streams[m] : P->panel_stream);
  const bool la = P->lookahead;
  const int n = P->n;
  const int tstride = n + P->nrhs_max;
  int rc;
  P->ev_used = 0; P->iv.clear(); P->n_gemm_launch = 0; P->gemm_flops = 0.0; P->gemm_cbytes = 0.0; P->n_big_launch = 0; P->big_flops = 0.0; P->ev_valid = false; P->last_batch = nmat;
  MA_HIP(hipMemsetAsync(P->pws.info, 0, 64, st));
  MA_MARK(e_begin, st);

  std::vector<int> k0s, nbs, rpbs, nblks;
  // batched panels: the systems of the batch share one panel kernel per panel (lock step is what this schedule is anyway)
  bool bp = nmat >= 2 && P->batch_panel && la && P->panel_overlap && P->midlane == 1;
  if (bp) {
    panel_schedule_batched(P, nmat, k0s, nbs, rpbs, nblks);
    for (size_t q = 0; q < k0s.size() && bp; ++q) {
      const size_t lds = ((lu_panel_lds_bytes(nbs[q], rpbs[q]) + 15) & ~(size_t)15) * (size_t)nmat;
      if (rpbs[q] > 64 || (long long)nblks[q] > (long long)lu_panel_slots_per_cu(lds, lu_panel_regs(1)) * P->ncu) bp = false;    // tall systems: one kernel per system after all
    }
    if (!bp) { k0s.clear(); nbs.clear(); rpbs.clear(); nblks.clear(); }
  }
  if (!bp) panel_schedule(P, k0s, nbs, rpbs, nblks);
  P->last_bp_nsys = bp ? nmat : 0;
  const int Q = (int)k0s.size();
  // the lane's interchange staging holds (kb-1) panels' columns
  const int kb = effective_kb(P, nbs);
  const int G = (Q + kb - 1) / kb;
  auto blk_first = [&](int g) { return g * kb; };
  auto blk_last = [&](int g) { return std::min(Q, (g + 1) * kb); };           // one past
  auto blk_end = [&](int g) { int q = blk_last(g) - 1; return k0s[q] + nbs[q]; };   // first column right of block g
  auto gemm = [&](int M_, int N_, int K_, const c64* a, const c64* b_, c64* c, hipStream_t s_, bool big_ = false) -> int {
    if (M_ <= 0 || N_ <= 0 || K_ <= 0) return MA_OK;
    P->n_gemm_launch++; P->gemm_flops += 8.0 * M_ * (double)N_ * K_; P->gemm_cbytes += 32.0 * M_ * (double)N_;   // every launch is timed: phase 3 (main lane) or 5 (look-ahead lanes)
    if (big_) { P->n_big_launch++; P->big_flops += 8.0 * M_ * (double)N_ * K_; }
    return lu_launch_zgemm_sub(M_, N_, K_, a, (size_t)n, b_, (size_t)n, c, (size_t)n, s_, P->use_3m, big_, &P->zmode);
  };

  // the look-ahead lane: factor the block column of block g of system m
  auto lane = [&](int m, int g) -> int {
    c64* A = As[m];
    hipStream_t sp = sps[m];
    const int e = blk_end(g);
    for (int q = blk_first(g); q < blk_last(g); ++q) {
      const int k0 = k0s[q], nb = nbs[q], a1 = k0 + nb;
      MA_MARK(t0, sp);
      // the panel's gather lists and inverted diagonal blocks: once, here; the main lane reuses them
      const int slot = (g & 1) * LU_KB_MAX + (q - blk_first(g));
      int* lists = P->d_lists[m] + (size_t)slot * LU_LISTS_LEN;
      c64* invd = P->d_invd[m] + (size_t)slot * LU_NB_MAX * 32;
      if ((rc = launch_panel(P, m, A, k0, nb, rpbs[q], nblks[q], P->pws_m[m], P->d_ipiv[m], lists, q == 0 || nbs[q - 1] < 4 || (P->reg_pair && nbs[q - 1] - LU_REG_NB < 4), sp, false))) return rc;
      MA_MARK(t1, sp);
      interval(P, t0, t1, 0);
      const bool fused_step = P->reg_panel;
      if (P->reg_panel && P->reg_pair) {                    // both halves' interchanges + U12 + inverses + the folded 64-pivot list: one launch
        if ((rc = lu_launch_lane_step2(A, n, k0, nb, P->d_half_lists[m], P->d_half_lists[m] + LU_LISTS_LEN, a1, e - a1, P->d_ipiv[m], lists, invd, P->pws.timeout, P->d_half_l10[m], sp))) return rc;
      } else if (fused_step) { if ((rc = lu_launch_lane_step(A, n, k0, nb, lists, a1, e - a1, invd, P->pws.timeout, sp))) return rc; }   // lists came from the panel kernel
      else if ((rc = lu_launch_perm(A, n, k0, nb, P->d_ipiv[m], lists, invd, P->pws.timeout, sp))) return rc;
      if (a1 < e) {
        if (!fused_step) {
          if ((rc = lu_launch_row_moves(A, n, nb, lists, P->d_tmp_l[m], LU_LANE_TSTRIDE, a1, e, 0, 0, nullptr, 0, sp))) return rc;
          const c64* T = A + (size_t)k0 * n + k0;
          if ((rc = lu_launch_trsm_mfma(T, n, nb, invd, A + (size_t)k0 * n + a1, (size_t)n, e - a1, nullptr, 0, 0, sp))) return rc;
        }
        MA_MARK(t2, sp);
        if ((rc = gemm(n - a1, e - a1, nb, A + (size_t)a1 * n + k0, A + (size_t)k0 * n + a1, A + (size_t)a1 * n + a1, sp))) return rc;
        MA_MARK(t3, sp);
        interval(P, t2, t3, 5);
      }
    }
    if (la) MA_HIP(hipEventRecord(P->ev_panel[m], sp));
    return MA_OK;
  };

  // batched panels: block g's panels for ALL systems. Stream sps[0] carries the panel kernels; after each one every system
  // does its own small work for that panel (lists + inverted blocks, interchanges, U12, the update of the block's remaining
  // columns) on its own stream, and the next panel kernel waits for all of them.
  bool lane_pending[LU_BATCH_MAX] = {};
  auto lane_bp = [&](int g) -> int {
    hipStream_t s0 = sps[0];
    const int e = blk_end(g);
    for (int q = blk_first(g); q < blk_last(g); ++q) {
      const int k0 = k0s[q], nb = nbs[q], a1 = k0 + nb;
      for (int m = 1; m < nmat; ++m) if (lane_pending[m]) { MA_HIP(hipStreamWaitEvent(s0, P->ev_lane[m], 0)); lane_pending[m] = false; }
      MA_MARK(t0, s0);
      int* ipivs[LU_BATCH_MAX]; for (int m = 0; m < LU_BATCH_MAX; ++m) ipivs[m] = P->d_ipiv[m < nmat ? m : 0];
      if ((rc = lu_launch_panel_batch(nmat, As, n, k0, nb, rpbs[q], nblks[q], P->ncu, P->pws_m, ipivs, q == 0 || nbs[q - 1] < 4, s0))) return rc;
      MA_MARK(t1, s0);
      interval(P, t0, t1, 0);
      MA_HIP(hipEventRecord(P->ev_bp, s0));
      const int slot = (g & 1) * LU_KB_MAX + (q - blk_first(g));
      for (int m = 0; m < nmat; ++m) {
        c64* A = As[m];
        hipStream_t sp = sps[m];
        if (m > 0) MA_HIP(hipStreamWaitEvent(sp, P->ev_bp, 0));
        int* lists = P->d_lists[m] + (size_t)slot * LU_LISTS_LEN;
        c64* invd = P->d_invd[m] + (size_t)slot * LU_NB_MAX * 32;
        if ((rc = lu_launch_perm(A, n, k0, nb, P->d_ipiv[m], lists, invd, P->pws.timeout, sp))) return rc;
        if (a1 < e) {
          if ((rc = lu_launch_row_moves(A, n, nb, lists, P->d_tmp_l[m], LU_LANE_TSTRIDE, a1, e, 0, 0, nullptr, 0, sp))) return rc;
          const c64* T = A + (size_t)k0 * n + k0;
          if ((rc = lu_launch_trsm_mfma(T, n, nb, invd, A + (size_t)k0 * n + a1, (size_t)n, e - a1, nullptr, 0, 0, sp))) return rc;
          MA_MARK(t2, sp);
          if ((rc = gemm(n - a1, e - a1, nb, A + (size_t)a1 * n + k0, A + (size_t)k0 * n + a1, A + (size_t)a1 * n + a1, sp))) return rc;
          MA_MARK(t3, sp);
          interval(P, t2, t3, 5);
        }
        if (m > 0) { MA_HIP(hipEventRecord(P->ev_lane[m], sp)); lane_pending[m] = true; }
      }
    }
    for (int m = 0; m < nmat; ++m) MA_HIP(hipEventRecord(P->ev_panel[m], sps[m]));
    return MA_OK;
  };

  if (la) {
    MA_HIP(hipEventRecord(P->ev_start, st));
    for (int m = 0; m < nmat; ++m) if (m == 0 || sps[m] != sps[0]) MA_HIP(hipStreamWaitEvent(sps[m], P->ev_start, 0));
  }
  if (bp) { if ((rc = lane_bp(0))) return rc; }
  else for (int m = 0; m < nmat; ++m) if ((rc = lane(m, 0))) return rc;
  // With several systems in flight the per-panel work of the current block (interchanges, U12 = L11^-1 A12, the updates
  // inside the block and of the next block's columns: short, latency-bound launches) runs on a stream of its own per
  // system; the caller's stream carries only the big updates, back to back over the systems.
  const bool split = la && nmat > 1 && P->panel_overlap && P->midlane;
  hipStream_t sms[LU_BATCH_MAX];
  for (int m = 0; m < LU_BATCH_MAX; ++m) sms[m] = split ? (P->midlane == 2 ? P->mid_streams[m] : sps[m]) : st;
  if (split) for (int m = 0; m < nmat; ++m) MA_HIP(hipStreamWaitEvent(sms[m], P->ev_start, 0));
  for (int g = 0; g < G; ++g) {
    const int a0 = k0s[blk_first(g)], e = blk_end(g);
    const int nright = n - e;
    const int enext = (g + 1 < G) ? blk_end(g + 1) : e;      // block g+1 occupies columns [e, enext)
    for (int m = 0; m < nmat; ++m) {
      c64* A = As[m]; c64* B = Bs ? Bs[m] : nullptr;
      hipStream_t sm = sms[m];
      if (la) MA_HIP(hipStreamWaitEvent(sm, P->ev_panel[m], 0));
      if (split && g > 0) MA_HIP(hipStreamWaitEvent(sm, P->ev_big[m], 0));     // block g-1's big update of this system
      MA_MARK(t0, sm);
      const bool bstep = !bp && block_step_ok(P, nbs, blk_first(g), blk_last(g));
      if (bstep) {
        if ((rc = block_main(P, m, A, B, nrhs, g, k0s, nbs, blk_first(g), blk_last(g), e, sm))) return rc;
        MA_MARK(tb, sm);
        interval(P, t0, tb, 2);
      } else {
      for (int q = blk_first(g); q < blk_last(g); ++q)
        if ((rc = lu_launch_row_moves(A, n, nbs[q], P->d_lists[m] + (size_t)((g & 1) * LU_KB_MAX + q - blk_first(g)) * LU_LISTS_LEN, P->d_tmp[m], tstride, 0, k0s[q], e, n, B, nrhs, sm))) return rc;
      MA_MARK(t1, sm);
      interval(P, t0, t1, 1);
      for (int q = blk_first(g); q < blk_last(g); ++q) {
        const int k0 = k0s[q], nb = nbs[q], a1 = k0 + nb;
        const c64* T = A + (size_t)k0 * n + k0;
        const c64* invd = P->d_invd[m] + (size_t)((g & 1) * LU_KB_MAX + q - blk_first(g)) * LU_NB_MAX * 32;
        MA_MARK(u0, sm);
        if ((rc = lu_launch_trsm_mfma(T, n, nb, invd, A + (size_t)k0 * n + e, (size_t)n, nright, nrhs ? B + k0 : nullptr, (size_t)n, nrhs, sm))) return rc;
        MA_MARK(u1, sm);
        interval(P, u0, u1, 2);
        for (int r = 0; r < nrhs && a1 < n; ++r)
          if ((rc = lu_launch_zgemv_sub(n - a1, nb, A + (size_t)a1 * n + k0, (size_t)n, B + (size_t)r * n + k0, B + (size_t)r * n + a1, sm))) return rc;
        MA_MARK(u2, sm);
        interval(P, u1, u2, 4);
        if (a1 < e && (rc = gemm(e - a1, nright, nb, A + (size_t)a1 * n + k0, A + (size_t)k0 * n + e, A + (size_t)a1 * n + e, sm))) return rc;
        MA_MARK(u3, sm);
        interval(P, u2, u3, split ? 5 : 3);
      }
      }
      MA_MARK(t3, sm);
      const bool narrow = la && nright > 0 && g + 1 < G;
      // narrow update of the next block's columns first, then factor them concurrently with the rest
      if (narrow && (rc = gemm(nright, enext - e, e - a0, A + (size_t)e * n + a0, A + (size_t)a0 * n + e, A + (size_t)e * n + e, sm))) return rc;
      MA_MARK(t4, sm);
      interval(P, t3, t4, split ? 5 : 3);
      if (split) MA_HIP(hipEventRecord(P->ev_mid[m], sm));
      if (narrow && !bp) {
        if (sm != sps[m]) { MA_HIP(hipEventRecord(P->ev_narrow[m], sm)); MA_HIP(hipStreamWaitEvent(sps[m], P->ev_narrow[m], 0)); }
        if ((rc = lane(m, g + 1))) return rc;
      }
      if (narrow && bp && m > 0) { MA_HIP(hipEventRecord(P->ev_lane[m], sm)); lane_pending[m] = true; }   // sm == sps[m] here (midlane 1)
    }
    // batched panels: the next block's panels once every system's columns of that block are up to date
    if (bp && la && nright > 0 && g + 1 < G && (rc = lane_bp(g + 1))) return rc;
    for (int m = 0; m < nmat && nright > 0; ++m) {
      c64* A = As[m];
      if (split) MA_HIP(hipStreamWaitEvent(st, P->ev_mid[m], 0));
      MA_MARK(t5, st);
      if (la) {
        if ((rc = gemm(nright, n - enext, e - a0, A + (size_t)e * n + a0, A + (size_t)a0 * n + enext, A + (size_t)e * n + enext, st, true))) return rc;
      } else {
        if ((rc = gemm(nright, nright, e - a0, A + (size_t)e * n + a0, A + (size_t)a0 * n + e, A + (size_t)e * n + e, st))) return rc;
        if (g + 1 < G && (rc = lane(m, g + 1))) return rc;
      }
      MA_MARK(t6, st);
      interval(P, t5, t6, 3);
      if (split) MA_HIP(hipEventRecord(P->ev_big[m], st));
    }
  }
  // backward substitution U x = y, block rows from the bottom
  MA_MARK(t7, st);
  if (nrhs > 0) {
    // a chain of 2 Q short launches per system: the systems of a batch run theirs side by side on their own streams
    const bool side = la && nmat > 1 && P->panel_overlap;
    if (side && !split) MA_HIP(hipEventRecord(P->ev_start, st));
    for (int m = 0; m < nmat; ++m) {
      c64* A = As[m]; c64* B = Bs[m];
      hipStream_t sb = split ? sms[m] : (side ? sps[m] : st);
      if (side && !split) MA_HIP(hipStreamWaitEvent(sb, P->ev_start, 0));
      for (int q = Q - 1; q >= 0; --q) {
        const int k0 = k0s[q], nb = nbs[q];
        if ((rc = lu_launch_trsv(true, A + (size_t)k0 * n + k0, n, nb, B + k0, (size_t)n, nrhs, sb))) return rc;
        for (int r = 0; r < nrhs && k0 > 0; ++r)
          if ((rc = lu_launch_zgemv_sub(k0, nb, A + k0, (size_t)n, B + (size_t)r * n + k0, B + (size_t)r * n, sb))) return rc;
      }
      if (side) { MA_HIP(hipEventRecord(P->ev_panel[m], sb)); }
    }
    if (side) for (int m = 0; m < nmat; ++m) MA_HIP(hipStreamWaitEvent(st, P->ev_panel[m], 0));
  } else if (split) {
    for (int m = 0; m < nmat; ++m) { MA_HIP(hipEventRecord(P->ev_mid[m], sms[m])); MA_HIP(hipStreamWaitEvent(st, P->ev_mid[m], 0)); }
  }
  MA_MARK(e_end, st);
  interval(P, t7, e_end, 4);
  interval(P, e_begin, e_end, 6);
  P->ev_last = e_end;
  if (P->timing) P->ev_valid = true;
  return MA_OK;
}

// ------------------------------------------------------------------ staged use: a pipeline over many systems
// factor_solve_batch moves its systems in lock step: all of them are in the update-bound early blocks together and in the
// latency-bound last blocks together. A driver that has a long sequence of systems (a frequency sweep) can instead keep the
// slots at DIFFERENT block indices -- slot s starts a quarter of a factorisation after slot s-1 -- so that in every round one
// slot brings a big update, one a medium one and one a small one: the caller's stream always has update work and every
// slot's latency-bound chain has the time of three updates to finish. The driver calls, per slot, stage_begin (A and b of
// the next system are ready on `stream`), then one stage_round per block index 0..G-1 together with the other slots, then
// stage_finish (backward substitution; `stream` waits for it). Same kernels, same arithmetic as factor_solve_batch.
namespace {
struct Stage {
  ma_lu_plan* P; int n, tstride, nrhs; hipStream_t st; int rc = MA_OK;
  std::vector<int> k0s, nbs, rpbs, nblks; int Q = 0, kb = 1, G = 0;
  explicit Stage(ma_lu_plan* P_, hipStream_t st_) : P(P_), n(P_->n), tstride(P_->n + P_->nrhs_max), nrhs(P_->cur_nrhs), st(st_) {
    if (P->stage_group >= 2) panel_schedule_batched(P, P->stage_group, k0s, nbs, rpbs, nblks);
    else panel_schedule(P, k0s, nbs, rpbs, nblks);
    Q = (int)k0s.size();
    kb = effective_kb(P, nbs);
    G = (Q + kb - 1) / kb;
  }
  int blk_first(int g) const { return g * kb; }
  int blk_last(int g) const { return std::min(Q, (g + 1) * kb); }
  int blk_end(int g) const { int q = blk_last(g) - 1; return k0s[q] + nbs[q]; }
  // Tail blocks: once few rows are left, a slot's big update is small and its latency-bound chain is what a round waits for (and,
  // first in the round's order on the caller's stream, what the other slots' updates queue behind). From there on the slot's
  // lane applies the whole trailing update of a block itself and the slot no longer touches the caller's stream: it finishes
  // at its chain's pace, beside the rounds of the other slots.
  bool tail(int g) const { return P->stage_group < 2 && g >= 0 && g < G && n - blk_end(g) <= P->tail_rows; }
  // The other end: while a slot's updates are big its lane has slack (a round lasts what the caller's stream needs for the three
  // slots' updates, the chain is shorter), and the 64 CUs the updates keep off are mostly idle between panels: the lane takes the
  // LAST columns of the block's big update (a multiple of 128: whole tiles), the caller's stream the rest.
  int lane_share(int g) const {
    if (P->share_pct <= 0 || P->stage_group >= 2 || tail(g) || g + 1 >= G) return 0;
    const int nright = n - blk_end(g), big_cols = n - blk_end(g + 1);
    if (nright < P->share_min_rows || big_cols <= 256) return 0;
    int w = (int)((long long)big_cols * P->share_pct / 100) / 128 * 128;
    return std::max(0, std::min(w, big_cols - 128));
  }
  hipStream_t lane_stream(int m) const { return (P->cu_split && P->chain_mask) ? P->chain_streams[m] : P->panel_streams[P->lane_alias > 0 ? m % P->lane_alias : m]; }
  hipStream_t pan_stream(int m) const { return (P->cu_split && P->pan_mask) ? P->pan_streams[m] : lane_stream(m); }
  hipStream_t big_stream() const { return P->cu_split ? P->big_stream : st; }
  int gemm(int M_, int N_, int K_, const c64* a, const c64* b_, c64* c, hipStream_t s_, bool big_ = false) {
    if (M_ <= 0 || N_ <= 0 || K_ <= 0) return MA_OK;
    // DIAGNOSTIC ONLY (wrong results): what the lanes' small updates cost the step -- MA_DIAG_SKIP_LANE_GEMM=<max K> drops them
    static const int skip_k = [] { const char* e = getenv("MA_DIAG_SKIP_LANE_GEMM"); return e ? atoi(e) : 0; }();
    if (!big_ && skip_k > 0 && K_ <= skip_k) return MA_OK;
    P->n_gemm_launch++; P->gemm_flops += 8.0 * M_ * (double)N_ * K_; P->gemm_cbytes += 32.0 * M_ * (double)N_;
    if (big_) { P->n_big_launch++; P->big_flops += 8.0 * M_ * (double)N_ * K_; }
    return lu_launch_zgemm_sub(M_, N_, K_, a, (size_t)n, b_, (size_t)n, c, (size_t)n, s_, P->use_3m, big_, &P->zmode);
  }
  // the look-ahead lane: factor the block column of block g of slot m
  int lane(int m, int g) {
    c64* A = P->cur_A[m]; hipStream_t sp = lane_stream(m);
    const int e = blk_end(g);
    for (int q = blk_first(g); q < blk_last(g); ++q) {
      const int k0 = k0s[q], nb = nbs[q], a1 = k0 + nb;
      // the panel kernel on the slot's panel stream (its own CUs when the chip is split), the rest of the chain on the lane stream
      hipStream_t pp = pan_stream(m);
      if (pp != sp) { MA_HIP(hipEventRecord(P->ev_chain[m], sp)); MA_HIP(hipStreamWaitEvent(pp, P->ev_chain[m], 0)); }
      MA_MARKD(t0, pp);
      const int slot = (g & 1) * LU_KB_MAX + (q - blk_first(g));
      int* lists = P->d_lists[m] + (size_t)slot * LU_LISTS_LEN;
      c64* invd = P->d_invd[m] + (size_t)slot * LU_NB_MAX * 32;
      if ((rc = launch_panel(P, m, A, k0, nb, rpbs[q], nblks[q], P->pws_m[m], P->d_ipiv[m], lists, q == 0 || nbs[q - 1] < 4 || (P->reg_pair && nbs[q - 1] - LU_REG_NB < 4), pp, pp != sp))) return rc;
      MA_MARKD(t1, pp);
      interval(P, t0, t1, 0);
      if (pp != sp) { MA_HIP(hipEventRecord(P->ev_pan[m], pp)); MA_HIP(hipStreamWaitEvent(sp, P->ev_pan[m], 0)); }
      // register panel kernel: it wrote the interchange list itself, and ONE launch does the interchanges, U12 and the inverted
      // diagonal block; otherwise: fold the pivots + invert, gather, scatter, U12
      const bool fused_step = P->reg_panel;
      if (P->reg_panel && P->reg_pair) {
        if ((rc = lu_launch_lane_step2(A, n, k0, nb, P->d_half_lists[m], P->d_half_lists[m] + LU_LISTS_LEN, a1, e - a1, P->d_ipiv[m], lists, invd, P->pws.timeout, P->d_half_l10[m], sp))) return rc;
      } else if (fused_step) { if ((rc = lu_launch_lane_step(A, n, k0, nb, lists, a1, e - a1, invd, P->pws.timeout, sp))) return rc; }
      else if ((rc = lu_launch_perm(A, n, k0, nb, P->d_ipiv[m], lists, invd, P->pws.timeout, sp))) return rc;
      if (a1 < e) {
        if (!fused_step) {
          if ((rc = lu_launch_row_moves(A, n, nb, lists, P->d_tmp_l[m], LU_LANE_TSTRIDE, a1, e, 0, 0, nullptr, 0, sp))) return rc;
          if ((rc = lu_launch_trsm_mfma(A + (size_t)k0 * n + k0, n, nb, invd, A + (size_t)k0 * n + a1, (size_t)n, e - a1, nullptr, 0, 0, sp))) return rc;
        }
        MA_MARK(t2, sp);
        if ((rc = gemm(n - a1, e - a1, nb, A + (size_t)a1 * n + k0, A + (size_t)k0 * n + a1, A + (size_t)a1 * n + a1, sp))) return rc;
        MA_MARK(t3, sp);
        interval(P, t2, t3, 5);
      }
    }
    return MA_OK;
  }
  // group form of the lane: block g's panels for the `cnt` slots of a group, ONE panel kernel per panel (on the first slot's
  // stream), then every slot's own small work for that panel on its stream; the next panel kernel waits for all of them
  int lane_group(const int* slots, int cnt, int g) {
    hipStream_t s0 = lane_stream(slots[0]);
    const int e = blk_end(g);
    c64* As_[LU_BATCH_MAX]; int* ipivs[LU_BATCH_MAX]; LuPanelWs wss[LU_BATCH_MAX];
    for (int i = 0; i < LU_BATCH_MAX; ++i) { const int m = slots[i < cnt ? i : 0]; As_[i] = P->cur_A[m]; ipivs[i] = P->d_ipiv[m]; wss[i] = P->pws_m[m]; }
    for (int q = blk_first(g); q < blk_last(g); ++q) {
      const int k0 = k0s[q], nb = nbs[q], a1 = k0 + nb;
      for (int i = 1; i < cnt; ++i) { const int m = slots[i]; if (P->stage_lane_pending[m]) { MA_HIP(hipStreamWaitEvent(s0, P->ev_lane[m], 0)); P->stage_lane_pending[m] = false; } }
      MA_MARKD(t0, s0);
      if ((rc = lu_launch_panel_batch(cnt, As_, n, k0, nb, rpbs[q], nblks[q], P->ncu, wss, ipivs, q == 0 || nbs[q - 1] < 4, s0))) return rc;
      MA_MARKD(t1, s0);
      interval(P, t0, t1, 0);
      MA_HIP(hipEventRecord(P->ev_lane[slots[0]], s0));                        // the group leader's event doubles as "panel done"
      const int slot_q = (g & 1) * LU_KB_MAX + (q - blk_first(g));
      for (int i = 0; i < cnt; ++i) {
        const int m = slots[i];
        c64* A = P->cur_A[m]; hipStream_t sp = lane_stream(m);
        if (i > 0) MA_HIP(hipStreamWaitEvent(sp, P->ev_lane[slots[0]], 0));
        int* lists = P->d_lists[m] + (size_t)slot_q * LU_LISTS_LEN;
        c64* invd = P->d_invd[m] + (size_t)slot_q * LU_NB_MAX * 32;
        if ((rc = lu_launch_perm(A, n, k0, nb, P->d_ipiv[m], lists, invd, P->pws.timeout, sp))) return rc;
        if (a1 < e) {
          if ((rc = lu_launch_row_moves(A, n, nb, lists, P->d_tmp_l[m], LU_LANE_TSTRIDE, a1, e, 0, 0, nullptr, 0, sp))) return rc;
          if ((rc = lu_launch_trsm_mfma(A + (size_t)k0 * n + k0, n, nb, invd, A + (size_t)k0 * n + a1, (size_t)n, e - a1, nullptr, 0, 0, sp))) return rc;
          MA_MARK(t2, sp);
          if ((rc = gemm(n - a1, e - a1, nb, A + (size_t)a1 * n + k0, A + (size_t)k0 * n + a1, A + (size_t)a1 * n + a1, sp))) return rc;
          MA_MARK(t3, sp);
          interval(P, t2, t3, 5);
        }
        if (i > 0) { MA_HIP(hipEventRecord(P->ev_lane[m], sp)); P->stage_lane_pending[m] = true; }
      }
    }
    return MA_OK;
  }
  // the per-panel work of block g right of the block, the update of the next block's columns, then the next lane
  int mwork(int m, int g, bool with_lane = true) {
    c64* A = P->cur_A[m]; c64* B = P->cur_B[m]; hipStream_t sm = lane_stream(m);
    const int a0 = k0s[blk_first(g)], e = blk_end(g), nright = n - e;
    const int enext = (g + 1 < G) ? blk_end(g + 1) : e;
    if (g > 0 && !tail(g - 1)) MA_HIP(hipStreamWaitEvent(sm, P->ev_big[m], 0));   // block g-1's big update of this slot
    MA_MARKD(t0, sm);
    if (P->stage_group < 2 && block_step_ok(P, nbs, blk_first(g), blk_last(g))) {
      if ((rc = block_main(P, m, A, B, nrhs, g, k0s, nbs, blk_first(g), blk_last(g), e, sm))) return rc;
      MA_MARKD(tb, sm);
      interval(P, t0, tb, 2);
    } else {
    for (int q = blk_first(g); q < blk_last(g); ++q)
      if ((rc = lu_launch_row_moves(A, n, nbs[q], P->d_lists[m] + (size_t)((g & 1) * LU_KB_MAX + q - blk_first(g)) * LU_LISTS_LEN, P->d_tmp[m], tstride, 0, k0s[q], e, n, B, nrhs, sm))) return rc;
    MA_MARKD(t1, sm);
    interval(P, t0, t1, 1);
    for (int q = blk_first(g); q < blk_last(g); ++q) {
      const int k0 = k0s[q], nb = nbs[q], a1 = k0 + nb;
      const c64* invd = P->d_invd[m] + (size_t)((g & 1) * LU_KB_MAX + q - blk_first(g)) * LU_NB_MAX * 32;
      MA_MARKD(u0, sm);
      if ((rc = lu_launch_trsm_mfma(A + (size_t)k0 * n + k0, n, nb, invd, A + (size_t)k0 * n + e, (size_t)n, nright, nrhs ? B + k0 : nullptr, (size_t)n, nrhs, sm))) return rc;
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
        interval(P, v0, v1, 5);
      }
    }
    }
    MA_MARK(t3, sm);
    const bool narrow = nright > 0 && g + 1 < G;
    if (narrow && (rc = gemm(nright, tail(g) ? nright : enext - e, e - a0, A + (size_t)e * n + a0, A + (size_t)a0 * n + e, A + (size_t)e * n + e, sm))) return rc;
    if (narrow && !tail(g)) {
      const int w = lane_share(g);                           // columns [n - w, n) of the big update on the lane (before the next panels: the lane has the slack here)
      if (w > 0 && (rc = gemm(nright, w, e - a0, A + (size_t)e * n + a0, A + (size_t)a0 * n + (n - w), A + (size_t)e * n + (n - w), sm))) return rc;
    }
    MA_MARK(t4, sm);
    interval(P, t3, t4, 5);
    MA_HIP(hipEventRecord(P->ev_mid[m], sm));
    if (P->stage_group >= 2) {                              // the group's next panels are launched once for all its slots (stage_round)
      if (narrow && m % P->stage_group != 0) { MA_HIP(hipEventRecord(P->ev_lane[m], sm)); P->stage_lane_pending[m] = true; }
      return MA_OK;
    }
    if (with_lane && narrow && (rc = lane(m, g + 1))) return rc;
    return MA_OK;
  }
  // the second half of mwork when a round issues the halves apart (lanes shared by two slots: every slot's work right of its block first --
  // it releases the slot's big update --, then the next block columns, the chain-bound slot's first)
  int next_lane(int m, int g) {
    if (P->stage_group >= 2) return MA_OK;
    const int e = blk_end(g);
    if (n - e > 0 && g + 1 < G && (rc = lane(m, g + 1))) return rc;
    return MA_OK;
  }
  int big(int m, int g) {
    c64* A = P->cur_A[m];
    const int a0 = k0s[blk_first(g)], e = blk_end(g), nright = n - e;
    const int enext = (g + 1 < G) ? blk_end(g + 1) : e;
    if (nright <= 0 || tail(g)) return MA_OK;
    hipStream_t bs = big_stream();
    MA_HIP(hipStreamWaitEvent(bs, P->ev_mid[m], 0));
    MA_MARK(t5, bs);
    if ((rc = gemm(nright, n - enext - lane_share(g), e - a0, A + (size_t)e * n + a0, A + (size_t)a0 * n + enext, A + (size_t)e * n + enext, bs, true))) return rc;
    MA_MARK(t6, bs);
    interval(P, t5, t6, 3);
    MA_HIP(hipEventRecord(P->ev_big[m], bs));
    return MA_OK;
  }
  // backward substitution of the system (A, B) that slot m has factored, on the slot's lane stream
  int backsub_issue(int m, c64* A, c64* B, int nrhs_) {
    hipStream_t sb = lane_stream(m);
    MA_MARKD(t7, sb);
    for (int q = Q - 1; q >= 0 && nrhs_ > 0; --q) {
      const int k0 = k0s[q], nb = nbs[q];
      if ((rc = lu_launch_trsv(true, A + (size_t)k0 * n + k0, n, nb, B + k0, (size_t)n, nrhs_, sb))) return rc;
      for (int r = 0; r < nrhs_ && k0 > 0; ++r)
        if ((rc = lu_launch_zgemv_sub(k0, nb, A + k0, (size_t)n, B + (size_t)r * n + k0, B + (size_t)r * n, sb))) return rc;
    }
    MA_MARKD(t8, sb);
    interval(P, t7, t8, 4);
    return MA_OK;
  }
  int backsub_end(int m, hipEvent_t ev) {                  // `st` waits for what was recorded on the lane; the run's last mark
    MA_HIP(hipStreamWaitEvent(st, ev, 0));
    MA_MARK(e_end, st);
    interval(P, P->stage_first_mark, e_end, 6);
    P->ev_last = e_end;
    if (P->timing) P->ev_valid = true;
    return MA_OK;
  }
  int backsub(int m) {
    if ((rc = backsub_issue(m, P->cur_A[m], P->cur_B[m], nrhs))) return rc;
    MA_HIP(hipEventRecord(P->ev_panel[m], lane_stream(m)));
    return backsub_end(m, P->ev_panel[m]);
  }
};
}  // namespace

extern "C" {
int ma_lu_plan_num_blocks(ma_lu_plan_t* P, int32_t* blocks) {
  MA_REQUIRE(P && blocks, MA_ERR_INVALID, "NULL argument");
  Stage S(P, nullptr);
  *blocks = S.G;
  return MA_OK;
}
// before a pipelined run: clear the status words and the timing accumulators
int ma_lu_plan_stage_reset(ma_lu_plan_t* P, void* stream) {
  MA_REQUIRE(P, MA_ERR_INVALID, "NULL plan");
  MA_REQUIRE(P->lookahead && P->panel_overlap, MA_ERR_UNSUPPORTED, "the staged schedule needs the look-ahead lanes (MA_LU_LOOKAHEAD / MA_LU_PANEL_OVERLAP)");
  // a plan that splits the chip runs its big updates on a CU-masked stream, which is a BLOCKING stream: a driver on the NULL stream would
  // serialise against it at every launch and lose the lanes' overlap without any error
  MA_REQUIRE(!(P->cu_split && stream == nullptr), MA_ERR_INVALID, "this plan splits the chip: drive its staged schedule from ma_lu_plan_main_stream (or any non-blocking stream), not from the NULL stream");
  MA_HIP(hipSetDevice(P->device));
  hipStream_t st = (hipStream_t)stream;
  int rc;
  P->ev_used = 0; P->iv.clear(); P->n_gemm_launch = 0; P->gemm_flops = 0.0; P->gemm_cbytes = 0.0; P->n_big_launch = 0; P->big_flops = 0.0; P->ev_valid = false; P->last_batch = 0;
  P->last_bp_nsys = 0;
  for (int i = 0; i < LU_BATCH_MAX; ++i) P->fin_state[i] = 0;
  MA_HIP(hipMemsetAsync(P->pws.info, 0, 64, st));
  MA_MARK(e0, st);
  P->stage_first_mark = e0;
  return MA_OK;
}
// the stream slot `slot`'s chain runs on (idle between stage_finish and the next stage_begin: a driver may assemble the
// slot's next system there, beside the other slots' work, and pass the same stream to stage_begin)
int ma_lu_plan_slot_stream(ma_lu_plan_t* P, int32_t slot, void** stream) {
  MA_REQUIRE(P && stream && slot >= 0 && slot < LU_BATCH_MAX, MA_ERR_INVALID, "bad argument");
  *stream = (void*)((P->cu_split && P->chain_mask) ? P->chain_streams[slot] : P->panel_streams[P->lane_alias > 0 ? slot % P->lane_alias : slot]);
  return MA_OK;
}
// the stream the plan runs its big trailing updates on when the chip is split (MA_LU_CU_SPLIT): masked to the update CUs. A driver
// that issues its own work between stage calls (assembly) may put it there instead of on a stream of its own; NULL when not split
// rounds between the starts of two slots of the staged schedule. Round-2 kernels (LDS panels, 0.75 ms of chain per panel beside
// the updates): G / (slots + 1), i.e. one quarter of a factorisation empty (59.9 ms against 60.7 at G / 3). Register pair panels
// (0.5 ms of chain): G / slots -- the slots evenly spread, so that the sum of the updates of one round never falls far below the
// chain of the slot that is closest to its end (50.8 ms against 51.4).
int ma_lu_plan_stage_spacing(ma_lu_plan_t* P, int32_t slots, int32_t* spacing) {
  MA_REQUIRE(P && spacing && slots >= 1, MA_ERR_INVALID, "bad argument");
  int32_t G = 0;
  int rc = ma_lu_plan_num_blocks(P, &G);
  if (rc) return rc;
  *spacing = (P->reg_panel && P->reg_pair) ? std::max(1, (G + slots / 2) / slots) : std::max(1, (G + slots) / (slots + 1));
  return MA_OK;
}
// how the plan splits the chip: CUs left to the panel kernels (0: no split; the mask's bits [0, panel_cus): panel_cus / 8 CUs of every XCD) and the chip's CUs
int ma_lu_plan_cu_split(ma_lu_plan_t* P, int32_t* panel_cus, int32_t* total_cus) {
  MA_REQUIRE(P && panel_cus && total_cus, MA_ERR_INVALID, "bad argument");
  *panel_cus = P->cu_split; *total_cus = P->ncu;
  return MA_OK;
}
int ma_lu_plan_main_stream(ma_lu_plan_t* P, void** stream) {
  MA_REQUIRE(P && stream, MA_ERR_INVALID, "bad argument");
  *stream = (void*)(P->cu_split ? P->big_stream : nullptr);
  return MA_OK;
}
int ma_lu_plan_stage_begin(ma_lu_plan_t* P, int32_t slot, void* dA, void* dB, int32_t nrhs, void* stream) {
  MA_REQUIRE(P && dA, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(slot >= 0 && slot < LU_BATCH_MAX, MA_ERR_INVALID, "slot %d outside 0..%d", slot, LU_BATCH_MAX - 1);
  MA_REQUIRE(nrhs >= 0 && nrhs <= P->nrhs_max && (nrhs == 0 || dB), MA_ERR_DIM, "nrhs must be 0..%d", P->nrhs_max);
  MA_REQUIRE(P->lookahead && P->panel_overlap, MA_ERR_UNSUPPORTED, "the staged schedule needs the look-ahead lanes");
  MA_REQUIRE(!(P->cu_split && stream == nullptr), MA_ERR_INVALID, "this plan splits the chip: drive its staged schedule from ma_lu_plan_main_stream (or any non-blocking stream), not from the NULL stream");
  MA_HIP(hipSetDevice(P->device));
  int rc = P->ensure_batch(slot + 1);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  P->cur_A[slot] = (c64*)dA; P->cur_B[slot] = (c64*)dB; P->cur_nrhs = nrhs;
  if (slot + 1 > P->last_batch) P->last_batch = slot + 1;
  MA_HIP(hipMemsetAsync(P->pws.info + slot, 0, sizeof(int), st));          // this slot's first-zero-pivot word
  MA_HIP(hipEventRecord(P->ev_prep[slot], st));
  MA_HIP(hipStreamWaitEvent(P->panel_streams[P->lane_alias > 0 ? slot % P->lane_alias : slot], P->ev_prep[slot], 0));
  if (P->cu_split && P->chain_mask) MA_HIP(hipStreamWaitEvent(P->chain_streams[slot], P->ev_prep[slot], 0));
  if (P->stage_group >= 2) {                               // group mode: the first panels start when the whole group has begun (stage_begin_group)
    if (slot % P->stage_group != 0) { MA_HIP(hipEventRecord(P->ev_lane[slot], P->panel_streams[slot])); P->stage_lane_pending[slot] = true; }
    return MA_OK;
  }
  Stage S(P, st);
  return S.lane(slot, 0);
}
// Staged use with groups (see ma_lu_plan::stage_group): group_size 0 / 1 = every slot on its own; 2..4 = slots [k g, (k+1) g) share
// one panel kernel per panel. Set before stage_reset; the slots of a group call stage_begin (any order), then ONE
// stage_begin_group(first_slot) starts the group's first block column; rounds and finishes as before, the slots of a group
// always with equal block indices.
int ma_lu_plan_stage_set_group(ma_lu_plan_t* P, int32_t group_size) {
  MA_REQUIRE(P && group_size >= 0 && group_size <= LU_GROUP_MAX, MA_ERR_INVALID, "group size must be 0..%d", LU_GROUP_MAX);
  // slot groups share ONE wavefront-per-system panel kernel, which is of the LDS family: a plan created with register panels
  // factors with the LDS family while groups are set (a plan stays in one family per factorisation: the two pivot the same
  // rows but round differently) and returns to its own with group_size < 2
  MA_REQUIRE(group_size < 2 || P->pivoting != MA_LU_PIVOT_TOURNAMENT, MA_ERR_UNSUPPORTED, "slot groups share a partial-pivoting panel kernel: not with a tournament-pivoting plan");
  if (group_size >= 2 && P->reg_panel0) {
    P->reg_panel = false; P->reg_pair = false;
    std::vector<int> k0s, nbs, rpbs, nblks;
    panel_schedule(P, k0s, nbs, rpbs, nblks);
    int rc = MA_OK;
    for (size_t q = 0; q < k0s.size() && !rc; ++q)
      if (q == 0 || rpbs[q] != rpbs[q - 1] || nbs[q] != nbs[q - 1] || nblks[q] > nblks[q - 1]) rc = lu_panel_admissible(nbs[q], rpbs[q], nblks[q], P->ncu);
    if (rc) { P->reg_panel = P->reg_panel0; P->reg_pair = P->reg_pair0; return rc; }
  } else if (group_size < 2) { P->reg_panel = P->reg_panel0; P->reg_pair = P->reg_pair0; }
  if (group_size >= 2) {
    std::vector<int> k0s, nbs, rpbs, nblks;
    panel_schedule_batched(P, group_size, k0s, nbs, rpbs, nblks);
    for (size_t q = 0; q < k0s.size(); ++q) {
      const size_t lds = ((lu_panel_lds_bytes(nbs[q], rpbs[q]) + 15) & ~(size_t)15) * (size_t)group_size;
      MA_REQUIRE(rpbs[q] <= 64 && (long long)nblks[q] <= (long long)lu_panel_slots_per_cu(lds, lu_panel_regs(1)) * P->ncu, MA_ERR_UNSUPPORTED,
                 "systems of %d rows are too tall for the shared panel kernel (%d rows per workgroup)", P->n, rpbs[q]);
    }
  }
  P->stage_group = group_size;
  for (int i = 0; i < LU_BATCH_MAX; ++i) P->stage_lane_pending[i] = false;
  return MA_OK;
}
int ma_lu_plan_stage_begin_group(ma_lu_plan_t* P, int32_t first_slot, void* stream) {
  MA_REQUIRE(P && P->stage_group >= 2 && first_slot >= 0 && first_slot % P->stage_group == 0 && first_slot + P->stage_group <= LU_BATCH_MAX, MA_ERR_INVALID, "bad group");
  MA_HIP(hipSetDevice(P->device));
  int slots[LU_GROUP_MAX];
  for (int i = 0; i < P->stage_group; ++i) { slots[i] = first_slot + i; MA_REQUIRE(P->cur_A[slots[i]], MA_ERR_INVALID, "slot %d has not begun", slots[i]); }
  Stage S(P, (hipStream_t)stream);
  return S.lane_group(slots, P->stage_group, 0);
}
// one round: slot slots[i] does block blocks[i] (consecutive rounds of a slot use consecutive blocks 0..G-1)
int ma_lu_plan_stage_round(ma_lu_plan_t* P, int32_t count, const int32_t* slots, const int32_t* blocks, void* stream) {
  MA_REQUIRE(P && slots && blocks && count >= 0 && count <= LU_BATCH_MAX, MA_ERR_INVALID, "bad argument");
  MA_HIP(hipSetDevice(P->device));
  Stage S(P, (hipStream_t)stream);
  for (int i = 0; i < count; ++i)
    MA_REQUIRE(slots[i] >= 0 && slots[i] < LU_BATCH_MAX && P->cur_A[slots[i]] && blocks[i] >= 0 && blocks[i] < S.G, MA_ERR_INVALID, "slot %d / block %d", slots[i], blocks[i]);
  if (P->lane_alias > 0 && P->stage_group < 2) {
    int ord[LU_BATCH_MAX];
    for (int i = 0; i < count; ++i) ord[i] = i;
    std::sort(ord, ord + count, [&](int a, int b) { return blocks[a] > blocks[b]; });      // the slot closest to its end first
    for (int i = 0; i < count; ++i) { int rc = S.mwork(slots[ord[i]], blocks[ord[i]], false); if (rc) return rc; }
    for (int i = 0; i < count; ++i) { int rc = S.next_lane(slots[ord[i]], blocks[ord[i]]); if (rc) return rc; }
  } else
  for (int i = 0; i < count; ++i) { int rc = S.mwork(slots[i], blocks[i]); if (rc) return rc; }
  if (P->stage_group >= 2) {
    // every group present in this round (all its slots, same block) launches the panels of its next block
    for (int i = 0; i < count; ++i) {
      if (slots[i] % P->stage_group != 0) continue;
      int gs[LU_GROUP_MAX];
      for (int t = 0; t < P->stage_group; ++t) {
        gs[t] = slots[i] + t;
        bool found = false;
        for (int q = 0; q < count; ++q) found = found || (slots[q] == gs[t] && blocks[q] == blocks[i]);
        MA_REQUIRE(found, MA_ERR_INVALID, "slot %d of the group of slot %d is missing from the round (or at another block)", gs[t], slots[i]);
      }
      const int g = blocks[i];
      const int e = S.blk_end(g);
      if (P->n - e > 0 && g + 1 < S.G) { int rc = S.lane_group(gs, P->stage_group, g + 1); if (rc) return rc; }
    }
  }
  // the big updates of the round, smallest first: the slot closest to the end of its factorisation has the least slack in
  // its chain (its next block waits for this update), the one at the start has the most
  int order[LU_BATCH_MAX];
  for (int i = 0; i < count; ++i) order[i] = i;
  static const bool by_size = [] { const char* e = getenv("MA_LU_STAGE_ORDER"); return !e || atoi(e) != 0; }();
  if (by_size) std::sort(order, order + count, [&](int a, int b) { return blocks[a] > blocks[b]; });
  for (int i = 0; i < count; ++i) { int rc = S.big(slots[order[i]], blocks[order[i]]); if (rc) return rc; }
  return MA_OK;
}
// after stage_finish: copy the slot's status word (0, or 1 + the column of the first zero pivot) to a device int on `stream`
int ma_lu_plan_stage_info_dev(ma_lu_plan_t* P, int32_t slot, int32_t* d_out, void* stream) {
  MA_REQUIRE(P && d_out && slot >= 0 && slot < LU_BATCH_MAX, MA_ERR_INVALID, "bad argument");
  MA_HIP(hipSetDevice(P->device));
  MA_HIP(hipMemcpyAsync(d_out, P->pws.info + slot, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MA_OK;
}
int ma_lu_plan_stage_finish(ma_lu_plan_t* P, int32_t slot, void* stream) {
  MA_REQUIRE(P && slot >= 0 && slot < LU_BATCH_MAX && P->cur_A[slot], MA_ERR_INVALID, "bad argument");
  MA_HIP(hipSetDevice(P->device));
  Stage S(P, (hipStream_t)stream);
  return S.backsub(slot);
}
// The finish in three steps, for a driver that hands the slot its next system in OTHER buffers (ma_bem_sweep_run swaps in a spare):
// _defer after the slot's last round: the factorisation of the slot's system is complete on `stream` (its status word may be copied),
// nothing is launched; _issue: the backward substitution goes onto the slot's lane stream behind whatever the lane has been given
// since -- the first block columns of the next system, after which the lane waits for that system's largest update anyway;
// _wait: `stream` waits for it (x is in the system's b). The 2 x 157 short launches of a 10 000-row backward substitution (6 ms on
// the lane) thus leave the path between a slot's last block and its next system's first panels.
int ma_lu_plan_stage_finish_defer(ma_lu_plan_t* P, int32_t slot, void* stream) {
  MA_REQUIRE(P && slot >= 0 && slot < LU_BATCH_MAX && P->cur_A[slot], MA_ERR_INVALID, "bad argument");
  MA_REQUIRE(P->stage_group < 2, MA_ERR_UNSUPPORTED, "deferred finish with slot groups");
  MA_REQUIRE(P->fin_state[slot] == 0, MA_ERR_INVALID, "slot %d already has a deferred finish", slot);
  MA_HIP(hipSetDevice(P->device));
  P->fin_A[slot] = P->cur_A[slot]; P->fin_B[slot] = P->cur_B[slot]; P->fin_nrhs[slot] = P->cur_nrhs; P->fin_state[slot] = 1;
  MA_HIP(hipStreamWaitEvent((hipStream_t)stream, P->ev_mid[slot], 0));      // the last block's work on the lane: every panel of the system is done
  return MA_OK;
}
int ma_lu_plan_stage_finish_issue(ma_lu_plan_t* P, int32_t slot) {
  MA_REQUIRE(P && slot >= 0 && slot < LU_BATCH_MAX && P->fin_state[slot] == 1, MA_ERR_INVALID, "slot %d has no deferred finish to issue", slot);
  MA_HIP(hipSetDevice(P->device));
  Stage S(P, nullptr);
  int rc = S.backsub_issue(slot, P->fin_A[slot], P->fin_B[slot], P->fin_nrhs[slot]);
  if (rc) return rc;
  MA_HIP(hipEventRecord(P->ev_fin[slot], S.lane_stream(slot)));
  P->fin_state[slot] = 2;
  return MA_OK;
}
int ma_lu_plan_stage_finish_wait(ma_lu_plan_t* P, int32_t slot, void* stream) {
  MA_REQUIRE(P && slot >= 0 && slot < LU_BATCH_MAX && P->fin_state[slot] == 2, MA_ERR_INVALID, "slot %d has no issued finish to wait for", slot);
  MA_HIP(hipSetDevice(P->device));
  Stage S(P, (hipStream_t)stream);
  P->fin_state[slot] = 0;
  return S.backsub_end(slot, P->ev_fin[slot]);
}
}  // extern "C"

// Solve with factors that an earlier ma_lu_plan_factor_solve_dev call on THIS plan left in d_A (the plan still holds
// the pivots): b <- P b panel by panel, forward substitution with the unit-lower factor, backward with the upper one.
// LuFactorization::solve (lu.rs:38-78).
static int solve_only(ma_lu_plan* P, c64* A, c64* B, int32_t nrhs, hipStream_t st) {
  const int n = P->n;
  const int tstride = n + P->nrhs_max;
  std::vector<int> k0s, nbs, rpbs, nblks;
  if (P->last_bp_nsys > 0) panel_schedule_batched(P, P->last_bp_nsys, k0s, nbs, rpbs, nblks);    // the panels the factors were built with
  else panel_schedule(P, k0s, nbs, rpbs, nblks);
  const int Q = (int)k0s.size();
  int rc;
  // the stored factors are in their final row order (later interchanges were applied to the earlier L columns), so every
  // interchange goes onto b first (zgetrs: laswp, then the triangular solves)
  for (int q = 0; q < Q; ++q)
    if ((rc = lu_launch_swaps(A, n, k0s[q], nbs[q], P->d_ipiv[0], P->d_lists[0], P->d_tmp[0], tstride, 0, 0, 0, 0, B, nrhs, nullptr, nullptr, st))) return rc;
  for (int q = 0; q < Q; ++q) {
    const int k0 = k0s[q], nb = nbs[q], a1 = k0 + nb;
    if ((rc = lu_launch_swaps(A, n, k0, nb, P->d_ipiv[0], P->d_lists[0], P->d_tmp[0], tstride, 0, 0, 0, 0, nullptr, 0, P->d_invd[0], nullptr, st))) return rc;   // inverted diagonal blocks only
    if ((rc = lu_launch_trsm_mfma(A + (size_t)k0 * n + k0, n, nb, P->d_invd[0], A, (size_t)n, 0, B + k0, (size_t)n, nrhs, st))) return rc;
    for (int r = 0; r < nrhs && a1 < n; ++r)
      if ((rc = lu_launch_zgemv_sub(n - a1, nb, A + (size_t)a1 * n + k0, (size_t)n, B + (size_t)r * n + k0, B + (size_t)r * n + a1, st))) return rc;
  }
  for (int q = Q - 1; q >= 0; --q) {
    const int k0 = k0s[q], nb = nbs[q];
    if ((rc = lu_launch_trsv(true, A + (size_t)k0 * n + k0, n, nb, B + k0, (size_t)n, nrhs, st))) return rc;
    for (int r = 0; r < nrhs && k0 > 0; ++r)
      if ((rc = lu_launch_zgemv_sub(k0, nb, A + k0, (size_t)n, B + (size_t)r * n + k0, B + (size_t)r * n, st))) return rc;
  }
  return MA_OK;
}

int ma_lu_plan_solve_dev(ma_lu_plan_t* P, void* dA_factored, void* dB, int32_t nrhs, void* stream) {
  MA_REQUIRE(P && dA_factored && dB, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(nrhs >= 1 && nrhs <= P->nrhs_max, MA_ERR_DIM, "nrhs must be 1..%d", P->nrhs_max);
  MA_HIP(hipSetDevice(P->device));
  return solve_only(P, (c64*)dA_factored, (c64*)dB, nrhs, (hipStream_t)stream);
}

int ma_lu_plan_factor_solve_dev(ma_lu_plan_t* P, void* dA, void* dB, int32_t nrhs, void* stream) {
  MA_REQUIRE(P && dA, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(nrhs >= 0 && nrhs <= P->nrhs_max, MA_ERR_DIM, "nrhs must be 0..%d", P->nrhs_max);
  MA_REQUIRE(nrhs == 0 || dB, MA_ERR_INVALID, "d_B is NULL");
  MA_HIP(hipSetDevice(P->device));
  c64* A = (c64*)dA; c64* B = (c64*)dB;
  return factor_solve_batch(P, 1, &A, &B, nrhs, (hipStream_t)stream);
}

// nmat (1..MA_LU_BATCH_MAX) independent n x n systems, e.g. the frequencies of a sweep kept in flight together.
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
  for (int i = 0; i < LU_BATCH_MAX; ++i) { if (P->panel_streams[i]) MA_HIP(hipStreamSynchronize(P->panel_streams[i])); if (P->mid_streams[i]) MA_HIP(hipStreamSynchronize(P->mid_streams[i])); }
  for (int i = 0; i < LU_BATCH_MAX; ++i) { if (P->pan_streams[i]) MA_HIP(hipStreamSynchronize(P->pan_streams[i])); if (P->chain_streams[i]) MA_HIP(hipStreamSynchronize(P->chain_streams[i])); }
  if (P->big_stream) MA_HIP(hipStreamSynchronize(P->big_stream));
  int info[16];
  MA_HIP(hipMemcpy(info, P->pws.info, sizeof(info), hipMemcpyDeviceToHost));
  MA_REQUIRE(info[LU_BATCH_MAX] != 2, MA_ERR_HIP, "a panel left a pivot outside its range: the factorisation was abandoned (no rows were moved with it)");
  MA_REQUIRE(info[LU_BATCH_MAX] == 0, MA_ERR_HIP, "panel factorisation abandoned: an exchange between the co-resident workgroups did not complete within its limit");
  for (int m = 0; m < P->last_batch; ++m) {
    MA_REQUIRE(info[m] >= 0, MA_ERR_RETRY, "system %d: a speculative panel was rejected and the plan runs without the fallback (optimistic mode): solve it again with ma_lu_plan_set_speculation(plan, MA_LU_SPECULATE_VERIFIED)", m);
    MA_REQUIRE(info[m] == 0, MA_ERR_SINGULAR, "system %d is singular: zero pivot at column %d", m, info[m] - 1);
  }
  return MA_OK;
}

int ma_lu_plan_last_timing(ma_lu_plan_t* P, double* out8) {
  MA_REQUIRE(P && out8, MA_ERR_INVALID, "NULL argument");
  MA_REQUIRE(P->ev_valid && P->ev_last >= 0, MA_ERR_INVALID, "no timed factorisation has run on this plan");
  MA_HIP(hipSetDevice(P->device));
  MA_HIP(hipEventSynchronize(P->ev[P->ev_last]));
  for (int i = 0; i < LU_BATCH_MAX; ++i) { if (P->panel_streams[i]) MA_HIP(hipStreamSynchronize(P->panel_streams[i])); if (P->mid_streams[i]) MA_HIP(hipStreamSynchronize(P->mid_streams[i])); }
  for (int i = 0; i < LU_BATCH_MAX; ++i) { if (P->pan_streams[i]) MA_HIP(hipStreamSynchronize(P->pan_streams[i])); if (P->chain_streams[i]) MA_HIP(hipStreamSynchronize(P->chain_streams[i])); }
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

// Update (zgemm) launches of the last call, main lane and look-ahead lanes, all systems of the batch: count, algorithmic
// flops (8 M N K each) and algorithmic C bytes (32 M N each). Their time is out8[3] + out8[7] of ma_lu_plan_last_timing.
// the big trailing updates on the caller's stream alone (their time is out8[3] of ma_lu_plan_last_timing)
int ma_lu_plan_last_big_update_stats(ma_lu_plan_t* P, double* launches, double* flops) {
  MA_REQUIRE(P && launches && flops, MA_ERR_INVALID, "NULL argument");
  *launches = P->n_big_launch; *flops = P->big_flops;
  return MA_OK;
}
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

// Test hook: C <- C - A B with the MFMA kernel on host buffers (row-major, tight leading dimensions).
int ma_test_zgemm_sub(int32_t M, int32_t N, int32_t K, const ma_c64* A, const ma_c64* B, ma_c64* C) {
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
  { bool m3 = true; if (const char* e0 = getenv("MA_ZGEMM_3M")) m3 = atoi(e0) != 0; const ZgemmMode zm = zgemm_mode_from_env();   /* test hook: the switches as they are NOW */
    rc = lu_launch_zgemm_sub(M, N, K, dA, (size_t)K, dB, (size_t)N, dC, (size_t)N, nullptr, m3, false, &zm); }
  if (!rc) { hipError_t e = hipMemcpy(C, dC, sizeof(c64) * (size_t)M * N, hipMemcpyDeviceToHost); if (e != hipSuccess) { set_error("copy back: %s", hipGetErrorString(e)); rc = MA_ERR_HIP; } }
  (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC);
  return rc;
}

// The admission rule of the panel kernels as a pure function (lu_kernels.hip, "Residency"): spinning workgroups of `lds_bytes`
// of LDS and `regs` vector registers per lane that a CU can ALWAYS take, whatever offsets terminating kernels left them at.
int ma_lu_panel_slots_per_cu(int64_t lds_bytes, int32_t regs, int32_t* slots) {
  MA_REQUIRE(slots && lds_bytes > 0 && regs >= 0, MA_ERR_INVALID, "bad argument");
  *slots = lu_panel_slots_per_cu((size_t)lds_bytes, regs);
  return MA_OK;
}

// Diagnostic build only (-DMA_PANEL_STAMPS): per-phase 100 MHz tick totals of workgroup 0 of every panel kernel
int ma_lu_plan_panel_stamps(ma_lu_plan_t* P, unsigned long long* out8, int reset) {
  MA_REQUIRE(P && out8, MA_ERR_INVALID, "NULL argument");
  MA_HIP(hipSetDevice(P->device));
  MA_HIP(hipDeviceSynchronize());
  unsigned long long* d = P->pws.diagrow + 2 * 2 * LU_NB_MAX;
  MA_HIP(hipMemcpy(out8, d, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (reset) MA_HIP(hipMemset(d, 0, 8 * sizeof(unsigned long long)));
  return MA_OK;
}

// Diagnostics (tools/panel_cotenancy.py): background load on a stream of the caller's -- `repeat` launches of the trailing-update
// kernel on device buffers (tight leading dimensions), or of the matrix-core probe (d_out: 256 * blocks doubles)
int ma_diag_zgemm_dev(int32_t M, int32_t N, int32_t K, const void* dA, const void* dB, void* dC, int32_t repeat, void* stream) {
  MA_REQUIRE(dA && dB && dC && M > 0 && N > 0 && K > 0, MA_ERR_INVALID, "bad argument");
  int rc = MA_OK;
  for (int i = 0; i < repeat && !rc; ++i)
    rc = lu_launch_zgemm_sub(M, N, K, (const c64*)dA, (size_t)K, (const c64*)dB, (size_t)N, (c64*)dC, (size_t)N, (hipStream_t)stream, true);
  return rc;
}
int ma_diag_mfma_burn(void* d_out, int32_t blocks, int32_t iters, int32_t repeat, void* stream) {
  MA_REQUIRE(d_out && blocks > 0 && iters > 0, MA_ERR_INVALID, "bad argument");
  int rc = MA_OK;
  for (int i = 0; i < repeat && !rc; ++i) rc = lu_launch_mfma_probe((double*)d_out, blocks, iters, (hipStream_t)stream);
  return rc;
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

}  // extern "C"
