// lu_calu.hip — panel factorisation with TOURNAMENT pivoting (communication-avoiding LU) for gfx950.
//
// The solve behind lu_solve (math-solvers/src/direct/lu.rs:142-153) returns x only: which rows served as pivots never crosses
// that boundary. Partial pivoting (lu_panel_reg_kernel) needs one chip-wide decision per COLUMN -- 10 000 dependent exchanges
// per 10 000-row system, each a wait of every panel workgroup on every other, which is why those workgroups must be co-resident
// and why the sweep kept 64 CUs free for them. Tournament pivoting needs one decision per PANEL:
//
//   leaves      every workgroup takes 256 rows x <= 32 columns into registers (lane = row) and runs Gaussian elimination with
//               partial pivoting on ITS rows only -- no traffic between workgroups -- and names its <= 32 pivot rows;
//   tree        the candidates of 8 nodes (8 x 32 = 256 rows, read again as they stand in the matrix) go through the same
//               elimination in ONE workgroup: the workgroup that arrives LAST at a node's counter carries on as that node, the
//               others have already left. Nobody waits for anybody: no spinning, no residency rule, no admission window, no
//               reserved CUs. 10 000 rows: 40 leaves -> 5 nodes -> the root;
//   root        its elimination IS the factorisation of the panel's pivot block: winners hold the rows of L11 \ U11. It writes
//               them to the panel's top rows, moves the displaced rows to the winners' old places, and leaves LAPACK-style
//               pivots (a sequence of interchanges) and the (destination, source) row list the step kernels apply right of the panel;
//   finish      lu_calu_finish_kernel: every other row of the panel, L = A U11^-1, lane = row on registers.
//
// Same outputs as lu_launch_panel_reg (pivots, list, the left-half rows L10 of a pair's right half): everything after the panel
// is the existing schedule. Growth is bounded as for partial pivoting in practice (Grigori, Demmel, Xiang: CALU); the tests hold
// the solutions against LAPACK's and the sweep's residuals at both ends of the frequency range.
#include "lu_kernels.hpp"
#include "lu_device.hpp"

namespace ma {

namespace {

constexpr int CALU_FAN = 8;                      // children per tree node: 8 x 32 candidates = the 256 lanes of a workgroup
static_assert(CALU_FAN * LU_REG_NB == 256, "a tree node's candidates fill one workgroup");

#ifdef MA_CALU_STAMPS
// diagnostic build (tools/calu_probe.hip): 100 MHz tick totals of the workgroups that reach the root, per phase
// [0] leaf load, [1] leaf elimination, [2] publish + counter, [3] node load, [4] node elimination, [5] root: sequence + displaced reads, [6] root: writes, [7] roots
__device__ unsigned long long g_calu_stamps[16];
#define CALU_STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0) acc_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define CALU_STAMP(i) do { } while (0)
#endif

struct CaluLds {
  dc urow[2][LU_REG_NB];                         // [column parity]: the pivot row of the column (entries >= column)
  dc rinv[2];                                    // the reciprocal of its entry of the column
  __attribute__((aligned(16))) unsigned key[2][4];   // every wavefront's best key (0: no row left)
  int flag;
  PivotSeqLds seq;                               // seq.win: the node's pivot rows, in pivot order (-1: the node ran out of rows); the rest: the root only
};
static_assert(LU_REG_NB == 32, "PivotSeqLds holds 32 pivots");

// crecip_fast without a branch (the same operations on the same operands, chosen by selects): every lane forms the reciprocal of
// its own entry BESIDE the wavefront's reduction, so that no reciprocal sits between a column's barrier and its elimination
__device__ __forceinline__ dc crecip_sel(dc z) {
  const bool sw = !(__builtin_fabs(z.im) < __builtin_fabs(z.re));
  const double p = sw ? z.im : z.re, q = sw ? z.re : z.im;
  const double e = q * rcp_nr(p), g = rcp_nr(__builtin_fma(q, e, p));
  return sw ? dc_make(e * g, -g) : dc_make(g, -e * g);
}

// Gaussian elimination with partial pivoting over the <= 256 rows a workgroup holds in registers (lane = row, `rowid` = the row's
// index in the matrix), columns 0..nbc-1. Returns the column this thread's row became the pivot row of (-1: none); S.seq.win[c] = the
// pivot row of column c.
// What a column costs is LDS INSTRUCTIONS, not arithmetic (tools/lds_cost_probe.hip: the CU's LDS takes one b128 store per ~14
// clocks and one b128 load per ~7.4 however many lanes are active, for all four wavefronts together): per column ONE wavefront
// stores the pivot row (32 - c entries) and four load it (31 - c each). Hence two barriers per column -- keys, then the winner's
// row -- instead of every wavefront storing its candidate ahead of one barrier (4 x the stores: 35 -> 18 us per 32 columns).
// ONE reduction per wavefront: key = 25 bits of |re| + |im| (exponent and 13 bits of mantissa) above 7 bits that prefer the lower
// lane, so the pivot is within 1.3e-4 of the column's largest entry and ties go to the lower thread; every lane forms the
// reciprocal of its own entry beside the reduction. A row of NaNs counts as magnitude 0: a pivot is always chosen while rows are left.
// lu.rs:106-110: a pivot below 1e-30 is LuError::SingularMatrix -- its elimination is skipped, `info` (the root passes it) takes
// 1 + the first such column.
#ifdef MA_CALU_SUBSTAMPS
__device__ unsigned long long g_calu_sub[8];     // shader-clock totals inside the elimination: [0] key + reduction + reciprocal, [1] row to LDS, [2] barrier, [3] selection, [4] elimination, [5] columns
#define CALU_SUB(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); sub_[i] += now_ - slast_; slast_ = now_; } while (0)
#else
#define CALU_SUB(i) do { } while (0)
#endif

template <int NB>
__device__ __forceinline__ int calu_gepp(dc (&a)[NB], bool valid, int rowid, int nbc, CaluLds& S, int* info, int k0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int rank = -1;
  bool done = !valid;
#ifdef MA_CALU_SUBSTAMPS
  unsigned long long sub_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, slast_ = __builtin_amdgcn_s_memtime();
#endif
  // No branch in the column loop touches the row registers: a lane whose row takes no part (done, beyond the panel's columns,
  // a singular column) eliminates with a multiplier of zero. (A divergent `if` around the update made the compiler copy all 128
  // row registers per column; the predicated form is one straight block between two barriers.)
  static_for<0, NB>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr int buf = c & 1;
    const bool live = c < nbc;
    const double mag = cabs1(a[c]);
    const unsigned hi = mag == mag ? (unsigned)((u64)__double_as_longlong(mag) >> 32) : 0u;
    const unsigned key = (done || !live) ? 0u : (((hi >> 7) << 7) | (unsigned)(64 - lane));
    unsigned m = wave_umax(key);
    dc rv = crecip_sel(a[c]);
    asm volatile("" : "+v"(rv.re), "+v"(rv.im), "+s"(m));   // both chains HERE, side by side (the reciprocal is not sunk into the one lane that stores it)
    const bool mine = key != 0u && key == m;               // at most one lane: the lane is part of the key
    CALU_SUB(0);
    if (lane == 0) S.key[buf][wave] = m;
    CALU_SUB(1);
    __syncthreads();
    CALU_SUB(2);
    const uint4 k4 = *reinterpret_cast<const uint4*>(S.key[buf]);
    const unsigned k0w = __builtin_amdgcn_readfirstlane(k4.x), k1w = __builtin_amdgcn_readfirstlane(k4.y),
                   k2w = __builtin_amdgcn_readfirstlane(k4.z), k3w = __builtin_amdgcn_readfirstlane(k4.w);
    // the best wavefront: larger magnitude part, ties to the lower wavefront; a wavefront without rows has key 0 and never wins
    // against one that has rows (whose key is >= 1 even at magnitude 0)
    int bw = 0; unsigned bk = k0w;
    if (k1w != 0u && (bk == 0u || (k1w >> 7) > (bk >> 7))) { bw = 1; bk = k1w; }
    if (k2w != 0u && (bk == 0u || (k2w >> 7) > (bk >> 7))) { bw = 2; bk = k2w; }
    if (k3w != 0u && (bk == 0u || (k3w >> 7) > (bk >> 7))) { bw = 3; bk = k3w; }
    const bool any = bk != 0u;
    const bool iam = mine && wave == bw;
    if (iam) {
      static_for<c, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; S.urow[buf][j] = a[j]; });
      S.rinv[buf] = rv;
      S.seq.win[c] = rowid;
    }
    if (!any && tid == 0) S.seq.win[c] = -1;
    CALU_SUB(3);
    __syncthreads();
    rank = iam ? c : rank;
    done = done || iam;
    const dc piv = S.urow[buf][c], ri = S.rinv[buf];
    const bool singular = !(piv.re * piv.re + piv.im * piv.im >= 1e-60);
    if (info && any && singular && tid == 0) atomicCAS(info, 0, k0 + c + 1);
    const bool act = any && !done && !singular;
    {
      const dc lf = a[c] * ri;
      const dc l = dc_make(act ? lf.re : 0.0, act ? lf.im : 0.0);
      a[c].re = act ? lf.re : a[c].re; a[c].im = act ? lf.im : a[c].im;
      const double nlr = -l.re, nli = -l.im, li = l.im;
      static_for<c + 1, NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const dc u = S.urow[buf][j];
        a[j].re = __builtin_fma(li, u.im, __builtin_fma(nlr, u.re, a[j].re));
        a[j].im = __builtin_fma(nli, u.re, __builtin_fma(nlr, u.im, a[j].im));
      });
    }
    CALU_SUB(4);
#ifdef MA_CALU_SUBSTAMPS
    sub_[5] += 1;
#endif
  });
#ifdef MA_CALU_SUBSTAMPS
  if (tid == 0 && blockIdx.x == 0) for (int i = 0; i < 6; ++i) atomicAdd(&g_calu_sub[i], sub_[i]);
#endif
  return rank;
}

// One launch = one panel's tournament. Grid: ceil((n - k0) / 256) leaves of 256 threads. cand[node][32] / counters[node]: the
// plan's per-slot tree workspace, nodes numbered level by level (leaves first); counters are zero between launches (the last
// arriver of a node resets it).
template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2)))
void lu_calu_panel_kernel(dc* __restrict__ A, int n, int k0, int nbc, int* __restrict__ cand, unsigned* __restrict__ counters,
                          int* __restrict__ info, int* __restrict__ ipiv, int* __restrict__ lists, dc* __restrict__ lrows, int lcol0,
                          const int* __restrict__ run_if_nonzero) {
  __shared__ CaluLds S;
  const int tid = threadIdx.x;
  if (run_if_nonzero && __hip_atomic_load(run_if_nonzero, RLX_AGENT) == 0) return;   // the speculative panel (lu_spec.hip) was accepted: nothing to do (uniform over the grid)
  int level_n = (int)gridDim.x, base = 0, node = (int)blockIdx.x;
  int rowid = k0 + node * 256 + tid;
  bool valid = rowid < n;
  dc a[NB];
#ifdef MA_CALU_STAMPS
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memrealtime();
  bool leaf_ = true;
#endif
  auto load_row = [&]() {
    const dc* src = A + (size_t)(valid ? rowid : k0) * n + k0;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = (valid && j < nbc) ? src[j] : dc_make(0.0, 0.0); });
  };
  load_row();
  CALU_STAMP(0);
  int rank;
  for (;;) {
    const bool root = level_n == 1;
    rank = calu_gepp<NB>(a, valid, rowid, nbc, S, root ? info : nullptr, k0);
    __syncthreads();                                         // S.win is complete
#ifdef MA_CALU_STAMPS
    CALU_STAMP(leaf_ ? 1 : 4); leaf_ = false;
#endif
    if (root) break;
    // publish the node's pivot rows; the workgroup that completes the parent's set of children carries on as the parent
    // (write-through stores, counted out before the counter moves; the reader's loads bypass its own L2 the same way: the exchange
    // idiom of lu_panel_reg_kernel -- a fence here would write back the whole L2 of the XCD, which an update kernel beside us keeps dirty)
    if (tid < NB) { __hip_atomic_store(cand + (size_t)(base + node) * NB + tid, S.seq.win[tid], RLX_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();
    const int parent = node / CALU_FAN;
    const int nchild = min(CALU_FAN, level_n - CALU_FAN * parent);
    if (tid == 0) {
      unsigned* ctr = counters + base + level_n + parent;
      const unsigned old = __hip_atomic_fetch_add(ctr, 1u, RLX_AGENT);
      const bool last = old == (unsigned)(nchild - 1);
      if (last) __hip_atomic_store(ctr, 0u, RLX_AGENT);
      S.flag = last ? 1 : 0;
    }
    __syncthreads();
    if (!S.flag) return;
    CALU_STAMP(2);
    const int child = tid / NB, k = tid % NB;
    rowid = child < nchild ? __hip_atomic_load(cand + (size_t)(base + CALU_FAN * parent + child) * NB + k, RLX_AGENT) : -1;
    valid = rowid >= k0 && rowid < n;
    load_row();                                              // the candidates as they stand in the matrix (nothing has been written yet)
    CALU_STAMP(3);
    base += level_n; node = parent; level_n = (level_n + CALU_FAN - 1) / CALU_FAN;
  }

  // ---- the root: the winners' registers hold the rows of L11 \ U11
  if (tid < 64) pivot_sequence(S.seq, k0, nbc);
  // right half of a 64-column panel (lu_plan.hip, pair form): the pivot rows' entries of the LEFT half's columns [lcol0, lcol0 + 32),
  // the block L10 the step after the panel solves with (as lu_panel_reg_kernel leaves them)
  if (lrows && rank >= 0) {
    const dc* lsrc = A + (size_t)rowid * n + lcol0;
    dc* ldst = lrows + (size_t)rank * LU_REG_NB;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; ldst[j] = lsrc[j]; });
  }
  __syncthreads();
  // the rows the winners displace from the top positions: they go, as they stand, to the places the interchanges give them
  // (every read before any write)
  dc mv[4]; int mp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q, e = idx >> 5, col = idx & 31;
    mp[q] = -1; mv[q] = dc_make(0.0, 0.0);
    if (e < S.seq.next && col < nbc) {
      const int sr = S.seq.ext_src[e], ds = S.seq.ext_pos[e];
      if (sr >= k0 && sr < n && ds >= k0 && ds < n) { mv[q] = A[(size_t)sr * n + k0 + col]; mp[q] = ds; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  CALU_STAMP(5);
  if (rank >= 0) {
    dc* dst = A + (size_t)(k0 + rank) * n + k0;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if (j < nbc) dst[j] = a[j]; });
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) if (mp[q] >= 0) A[(size_t)mp[q] * n + k0 + ((tid + 256 * q) & 31)] = mv[q];
  if (tid < nbc) ipiv[k0 + tid] = S.seq.ipiv[tid];
  if (lists) {
    if (tid == 0) lists[0] = S.seq.lm;
    if (tid < S.seq.lm) { lists[1 + tid] = S.seq.ldst[tid]; lists[1 + 2 * LU_NB_MAX + tid] = S.seq.lsrc[tid]; }
  }
#ifdef MA_CALU_STAMPS
  CALU_STAMP(6);
  if (tid == 0) { acc_[7] = 1; for (int i = 0; i < 8; ++i) atomicAdd(&g_calu_stamps[i], acc_[i]); }
#endif
}

// L = A U11^-1 for the rows below the panel's pivot block: lane = row, the row's entries in registers, U11 (and the reciprocals
// of its diagonal) broadcast from LDS. Column by column the same operations, in the same order, as the elimination of calu_gepp.
template <int NB>
__global__ __launch_bounds__(256) void lu_calu_finish_kernel(dc* __restrict__ A, int n, int k0, int nbc, const int* __restrict__ run_if_nonzero) {
  if (run_if_nonzero && __hip_atomic_load(run_if_nonzero, RLX_AGENT) == 0) return;
  __shared__ __attribute__((aligned(16))) dc U[NB][NB];
  __shared__ __attribute__((aligned(16))) dc rinv[NB];
  __shared__ int sing[NB];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int i = idx / NB, j = idx % NB;
    U[i][j] = (i < nbc && j < nbc && j >= i) ? A[(size_t)(k0 + i) * n + k0 + j] : dc_make(0.0, 0.0);
  }
  __syncthreads();
  if (tid < NB) {
    const dc piv = U[tid][tid];
    const bool s = tid >= nbc || !(piv.re * piv.re + piv.im * piv.im >= 1e-60);
    sing[tid] = s ? 1 : 0;
    rinv[tid] = s ? dc_make(0.0, 0.0) : crecip_fast(piv);
  }
  __syncthreads();
  const int row = k0 + nbc + (int)blockIdx.x * 256 + tid;
  if (row >= n) return;
  dc a[NB];
  dc* p = A + (size_t)row * n + k0;
  static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = j < nbc ? p[j] : dc_make(0.0, 0.0); });
  static_for<0, NB>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (c < nbc && !sing[c]) {
      const dc l = a[c] * rinv[c];
      a[c] = l;
      const double nlr = -l.re, nli = -l.im, li = l.im;
      static_for<c + 1, NB>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const dc u = U[c][j];
        a[j].re = __builtin_fma(li, u.im, __builtin_fma(nlr, u.re, a[j].re));
        a[j].im = __builtin_fma(nli, u.re, __builtin_fma(nlr, u.im, a[j].im));
      });
    }
  });
  static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if (j < nbc) p[j] = a[j]; });
}

}  // namespace

// nodes of the tournament tree over `leaves` leaf workgroups (what a slot's workspace must hold)
int lu_calu_tree_nodes(int leaves) {
  int total = 0;
  for (int l = leaves; ; l = (l + CALU_FAN - 1) / CALU_FAN) { total += l; if (l <= 1) break; }
  return total;
}

int lu_launch_panel_calu(c64* A, int n, int k0, int nb, const LuCaluWs& ws, int* info, int* ipiv, int* lists, hipStream_t st, c64* lrows, int lcol0, const int* run_if_nonzero) {
  MA_REQUIRE(nb >= 1 && nb <= LU_REG_NB && k0 >= 0 && k0 + nb <= n, MA_ERR_INVALID, "panel [%d, %d) outside 0..%d", k0, k0 + nb, n);
  MA_REQUIRE(!lrows || (lcol0 >= 0 && lcol0 + LU_REG_NB <= k0), MA_ERR_INVALID, "left-half columns [%d, %d) not left of the panel at %d", lcol0, lcol0 + LU_REG_NB, k0);
  const int leaves = (n - k0 + 255) / 256;
  MA_REQUIRE(ws.cand && ws.counters && lu_calu_tree_nodes(leaves) <= ws.max_nodes, MA_ERR_INVALID, "tournament tree of %d leaves outside the workspace (%d nodes)", leaves, ws.max_nodes);
  hipLaunchKernelGGL(lu_calu_panel_kernel<LU_REG_NB>, dim3(leaves), dim3(256), 0, st, reinterpret_cast<dc*>(A), n, k0, nb, ws.cand, ws.counters, info, ipiv, lists,
                     reinterpret_cast<dc*>(lrows), lcol0, run_if_nonzero);
  MA_HIP(hipGetLastError());
  const int below = n - k0 - nb;
  if (below > 0) {
    hipLaunchKernelGGL(lu_calu_finish_kernel<LU_REG_NB>, dim3((below + 255) / 256), dim3(256), 0, st, reinterpret_cast<dc*>(A), n, k0, nb, run_if_nonzero);
    MA_HIP(hipGetLastError());
  }
  return MA_OK;
}

}  // namespace ma
