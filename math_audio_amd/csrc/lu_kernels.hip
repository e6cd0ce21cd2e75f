// lu_kernels.hip — dense complex128 LU (partial pivoting) and triangular solves for gfx950.
//
// Replaces the arithmetic behind lu_solve (math-solvers/src/direct/lu.rs:142-153 -> LAPACK zgesv).
// The matrix is ndarray C-order (row-major) and stays resident in HBM; pivoting is by rows
// (swaps are contiguous 16-B-per-lane row copies). Per panel of nb <= 128 columns:
//   lu_panel_kernel    co-resident workgroups, each keeping its <= 43 rows of a 64-column panel in LDS (48 KB: two
//                      systems' panels and two update workgroups share a CU); one chip-wide exchange per column (sc1
//                      write-through stores, tagged 8-byte granules gathered in two levels, sc1 loads) picks the pivot
//                      and hands every workgroup the pivot row.
//   lu_perm_kernel     one wavefront folds the panel's swap sequence into a gather list; further blocks invert the 32 x 32
//                      diagonal blocks of L11 (16 x 16 blocks through LDS: 47 registers, 26 KB);
//   lu_gather/scatter  apply the list to the columns left and right of the panel (and to the RHS).
//   lu_trsm64_kernel   U12 = L11^-1 A12 as MFMA products with the inverted diagonal blocks, panels of <= 64 columns, 94
//                      registers and 17 KB so that it runs beside a trailing update (lu_trsm_mfma_kernel: up to 128 columns);
//   lu_trsv_kernel     the nb x nb triangular solves of the backward substitution, one wavefront each.
//   zgemm3m_dma_kernel A22 -= L21 U12 on v_mfma_f64_16x16x4_f64, 3 real products per complex product, 64 x 128 tiles, operands by LDS-DMA
//   zgemm3m_sub_kernel the same with register staging and 64 x 64 tiles: K not a multiple of 8
//                      (zgemm_sub_kernel: the 4-product form, 128 x 128 tiles).
#include "lu_kernels.hpp"
#include "ma_device_math.hpp"
#include "lu_device.hpp"
#include <climits>
#include <algorithm>
#include <mutex>
#include <type_traits>

namespace ma {

// better (value, row) candidate: larger value, ties -> lower row (izamax takes the first maximum)
__device__ __forceinline__ bool cand_better(double v, int r, double bv, int br) { return v > bv || (v == bv && r < br); }

// ------------------------------------------------------------------ panel factorisation with the rows in registers (round 3)
// lane = row: a thread keeps its row's NB panel entries in registers for the whole panel (4 NB vector registers), the column
// loop is unrolled so that every register index is static, and a workgroup of 256 threads holds 256 rows: a 10 000-row panel
// is 40 workgroups instead of 233, so the per-column exchange is ONE flat sweep over <= 64 granules per lane (1.3-1.7 us on
// idle CUs against 3.7 for the two-level gather over 233) and the rank-1 update is 4 FMAs per entry straight on registers
// (no LDS traffic, no barrier between its parts). Rows never move during the panel: a thread tracks the POSITION its row
// holds under LAPACK's sequence of interchanges (`mypos`: the pivot row of column c takes position k0 + c, the row that was
// there takes the pivot's position) and writes its row to that position at the end, so the diagonal row needs no exchange
// at all. Per column and workgroup: every wavefront reduces its candidate (top 32 bits of |re| + |im|, ties to the lowest
// position -- the rule of lu_panel_kernel) with DPP; barrier; the wavefront that holds the workgroup's best row stages it
// through LDS (one lane writes, 32 lanes read) and publishes it write-through, then the granule {value, tag, position};
// wavefront 0 sweeps all granules (data = flag), fetches the winner's row into LDS; barrier; everybody eliminates.
// The kernel is meant for CUs that no throughput kernel shares (CU-masked streams, lu_plan.hip): there its exchange runs
// at the idle round trip (profiles/r03_cumask_probe.txt), and its registers (about 200 per lane, one wavefront per SIMD)
// are why it is admitted one workgroup per CU.
template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void lu_panel_reg_kernel(dc* __restrict__ A, int n, int k0, int nbc, LuPanelWs ws, int* __restrict__ ipiv, int* __restrict__ lists,
                         dc* __restrict__ lrows, int lcol0, const int* __restrict__ run_if_nonzero) {
  // the speculative panel (lu_spec.hip) was accepted: nothing to do. The word is final before this grid starts and every workgroup
  // reads the same value, so either all of them exchange or none does
  if (run_if_nonzero && __hip_atomic_load(run_if_nonzero, RLX_AGENT) == 0) return;
  __shared__ __attribute__((aligned(16))) dc s_urow[2][NB];   // pivot rows of the current and the previous column
  __shared__ __attribute__((aligned(16))) dc s_stage[NB];     // the row a workgroup sends: written by the lane that holds it, read by 32 lanes
  __shared__ unsigned s_m[4];
  __shared__ unsigned s_pos[4];
  __shared__ int s_lane[4];
  __shared__ int s_misc[4];                              // [0] pivot position, [1] fail, [2] poison seen at the start
  constexpr unsigned NONE = 0xFFFFFFu;
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, G = gridDim.x;
  const int row0 = k0 + b * 256 + tid;
  const bool valid = row0 < n;
  int mypos = row0;
  bool done = !valid;                                    // rows beyond n take no part
  dc a[NB];
  {
    const dc* src = A + (size_t)(valid ? row0 : k0) * n + k0;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = (valid && j < nbc) ? src[j] : dc_make(0.0, 0.0); });
  }
  if (tid == 0) {
    s_misc[1] = 0; s_misc[2] = (int)__hip_atomic_load(ws.timeout, RLX_AGENT);
    // the panel's interchange list (what lu_perm_kernel folds from the pivots): rows end where `mypos` says, so every thread whose
    // row moved appends (destination, source) itself. The count is cleared here, before this workgroup publishes anything: every
    // other workgroup's first append comes after a sweep that saw this workgroup's first granule
    if (b == 0 && lists) { __hip_atomic_store(lists, 0, RLX_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  __syncthreads();
  if (s_misc[2] != 0) {                                  // poisoned plan: identity pivots, nothing else (see lu_panel_kernel)
    if (b == 0) for (int j = tid; j < nbc; j += 256) ipiv[k0 + j] = k0 + j;
    return;
  }
  // this wavefront's candidate of column c: top 32 bits of |re| + |im|, ties to the lowest position
  auto candidate = [&](dc v) {
    const double mag = cabs1(v);
    const bool offer = !done && mag == mag;              // a NaN is never offered
    const unsigned hi = offer ? (unsigned)((u64)__double_as_longlong(mag) >> 32) : 0u;
    const unsigned m = wave_umax(hi);
    const unsigned pk = (offer && hi == m) ? (unsigned)mypos : NONE;
    const unsigned pmin = wave_umin(pk);
    const u64 bm = __ballot(pk == pmin && pmin != NONE);
    if (lane == 0) { s_m[wave] = m; s_pos[wave] = pmin; s_lane[wave] = bm ? (int)__builtin_ctzll(bm) : 0; }
  };
  candidate(a[0]);
  __syncthreads();                                       // B1(0)
  bool dead = false;                                     // uniform: an exchange was abandoned, the workgroup only falls through
  dc lprev = dc_make(0.0, 0.0);                          // this row's multiplier of the previous column; its rank-1 update is still due on columns > c
  bool upd_pending = false;                              // per lane: the previous column's update is due on this row
  static_for<0, NB>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (c < nbc && !dead) {
    const int gc = k0 + c, buf = c & 1;
    const unsigned want = (unsigned)(c + 1);
    // ---- after B1(c): the workgroup's candidate; the wavefront that holds it sends the row off as it stands -- with the previous
    // column's rank-1 update still due on the columns right of c (every receiver completes it on the row it fetches)
    unsigned bmax = s_m[0], bpos = s_pos[0]; int bw = 0;
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const unsigned mw = s_m[w], pw = s_pos[w];
      if (pw != NONE && (bpos == NONE || mw > bmax || (mw == bmax && pw < bpos))) { bmax = mw; bpos = pw; bw = w; }
    }
    const bool sender = wave == bw;
    if (sender && bpos != NONE) {
      // one lane's row through LDS to 32 lanes: two coalesced write-through stores instead of 64 single-lane ones (each of those
      // is a fabric write of its own: 2.0 us until every workgroup's granule was seen against 1.66 with the staging)
      if (lane == s_lane[bw]) static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; s_stage[j] = a[j]; });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane < NB) {
        const dc v = s_stage[lane];
        u64* dst = ws.candrow + ((size_t)buf * ws.max_blocks + b) * (2 * LU_NB_MAX) + 2 * lane;
        st_sc1(dst, v.re); st_sc1(dst + 1, v.im);
      }
    }
    // ---- the bulk of the previous column's rank-1 update, on registers; it overlaps the write-through of the row above
    if constexpr (c > 0) {
      if (upd_pending) {
        const dc* up = s_urow[(c - 1) & 1];
        dc u[NB];
        static_for<c + 1, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; u[j] = up[j]; });
        const double nlr = -lprev.re, nli = -lprev.im, li = lprev.im;
        static_for<c + 1, NB>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          a[j].re = __builtin_fma(li, u[j].im, __builtin_fma(nlr, u[j].re, a[j].re));
          a[j].im = __builtin_fma(nli, u[j].re, __builtin_fma(nlr, u[j].im, a[j].im));
        });
      }
    }
    if (sender) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the row is out before the granule says so
      if (lane == 0)
        __hip_atomic_store(ws.cand + ((size_t)buf * ws.max_blocks + b) * LU_GRANULE_STRIDE,
                           ((u64)(bpos != NONE ? bmax : 0u) << 32) | ((u64)want << 24) | (u64)bpos, RLX_AGENT);
    }
    // ---- wavefront 0: sweep every workgroup's granule until all carry this column's tag, reduce, fetch the winner's row
    if (wave == 0) {
      const u64 t0 = __builtin_amdgcn_s_memrealtime();
#ifdef MA_DIAGNOSTICS
      bool fail = gc == ws.test_abort_col && b == G - 1;   // diagnostic build only: this workgroup behaves as if its wait had expired
#else
      bool fail = false;
#endif
      unsigned bhi = 0, bps = NONE; int bblk = -1;
      const u64* gbase = ws.cand + (size_t)buf * ws.max_blocks * LU_GRANULE_STRIDE;
      // The row of the best candidate SO FAR is fetched while the sweep is still waiting for the other workgroups (a granule that
      // carries this column's tag is final, and its row was out before it): when the last granule arrives the winner's row is
      // usually here already or on its way -- one dependent round trip less per column. hv / hl: the prefetched row (lane j: entry j)
      // and the previous column's multiplier of that row; hb: whose it is (-1: none yet)
      dc hv = dc_make(0.0, 0.0), hl = dc_make(0.0, 0.0); int hb = -1;
      int wb = -1; unsigned p = NONE;
      const int jl = lane < NB ? lane : 0;
      auto fetch_row = [&](int blk) {
        const u64* src = ws.candrow + ((size_t)buf * ws.max_blocks + blk) * (2 * LU_NB_MAX);
        hv = dc_make(ld_sc1(src + 2 * jl), ld_sc1(src + 2 * jl + 1));
        if constexpr (c > 0) hl = dc_make(ld_sc1(src + 2 * (c - 1)), ld_sc1(src + 2 * (c - 1) + 1));
        hb = blk;
      };
      while (!fail) {
        bool ok = true; bhi = 0; bps = NONE; bblk = -1;
        for (int t = lane; t < G; t += 64) {
          const u64 g = __hip_atomic_load(gbase + (size_t)t * LU_GRANULE_STRIDE, RLX_AGENT);
          const bool here = (((unsigned)(g >> 24) & 0xFFu) == want);
          ok = ok && here;
          const unsigned h = (unsigned)(g >> 32), ps = (unsigned)g & NONE;
          if (here && ps != NONE && (bps == NONE || h > bhi || (h == bhi && ps < bps))) { bhi = h; bps = ps; bblk = t; }
        }
        const unsigned ab = __hip_atomic_load(ws.timeout, RLX_AGENT);     // the plan's abort flag rides along
        // best of the granules that have arrived
        const unsigned mh = wave_umax(bps != NONE ? bhi : 0u);
        const unsigned pk = (bps != NONE && bhi == mh) ? bps : NONE;
        p = wave_umin(pk);
        const u64 wm = __ballot(pk == p && p != NONE);
        wb = wm ? __shfl(bblk, (int)__builtin_ctzll(wm), 64) : -1;
        const bool all = __all(ok);
        if (wb >= 0 && wb != hb) fetch_row(wb);
        if (all) break;
        if (ab != 0u) { fail = true; break; }
        __builtin_amdgcn_s_sleep(LU_POLL_SLEEP);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) fail = true;   // 4 s at 100 MHz: never hang
      }
      if (!fail) {
        dc v = dc_make(0.0, 0.0);                        // no candidate anywhere: a zero pivot row, the column is skipped as singular
        if (wb >= 0) {
          v = hv;
          if constexpr (c > 0) {
            // the row was sent with the previous column's update due on the columns right of c: complete it (the sender's own copy
            // goes through the same two fused multiply-adds per component in its registers)
            const dc u = s_urow[(c - 1) & 1][jl];
            if (lane > c) {
              v.re = __builtin_fma(hl.im, u.im, __builtin_fma(-hl.re, u.re, v.re));
              v.im = __builtin_fma(-hl.im, u.re, __builtin_fma(-hl.re, u.im, v.im));
            }
          }
        }
        if (lane < NB) s_urow[buf][lane] = v;
      }
      if (lane == 0) {
        if (fail) { __hip_atomic_store(ws.timeout, 1u, RLX_AGENT); s_misc[1] = 1; }
        s_misc[0] = (wb >= 0 && p < (unsigned)n && p >= (unsigned)gc) ? (int)p : gc;
      }
    }
    __syncthreads();                                     // B2(c): pivot row and position of column c are in LDS
    if (s_misc[1]) {                                     // uniform: the whole workgroup gives up; the columns it did not reach get
      if (b == 0) for (int j = c + tid; j < nbc; j += 256) ipiv[k0 + j] = k0 + j;   // identity pivots (the plan is poisoned: MA_ERR_HIP)
      dead = true;
    } else {
    const int p = s_misc[0];
    const dc piv = s_urow[buf][c];
    // lu.rs:106-110: a pivot column whose largest |z| is below 1e-30 is LuError::SingularMatrix; its elimination is skipped
    const bool singular = !(piv.re * piv.re + piv.im * piv.im >= 1e-60);
    if (b == 0 && tid == 0) { ipiv[gc] = p; if (singular) atomicCAS(ws.info, 0, gc + 1); }
    upd_pending = false;
    if (!done) {
      if (mypos == p) { done = true; mypos = gc; }       // this row is the pivot row: it rests at position k0 + c from now on
      else {
        if (mypos == gc) mypos = p;                      // the row that sat on the diagonal takes the pivot's place
        if (!singular) {
          const dc l = a[c] * crecip_fast(piv);
          a[c] = l;
          lprev = l; upd_pending = true;
          if constexpr (c + 1 < NB) {                    // the next column at once: its candidates go out before the rest of this update
            const dc u = s_urow[buf][c + 1];
            a[c + 1].re = __builtin_fma(l.im, u.im, __builtin_fma(-l.re, u.re, a[c + 1].re));
            a[c + 1].im = __builtin_fma(-l.im, u.re, __builtin_fma(-l.re, u.im, a[c + 1].im));
          }
        }
      }
    }
    if constexpr (c + 1 < NB) {
      if (c + 1 < nbc) {
        candidate(a[c + 1]);
        __syncthreads();                                 // B1(c + 1)
      }
    }
    }
    }
  });
  if (valid && !dead) {
    dc* dst = A + (size_t)mypos * n + k0;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if (j < nbc) dst[j] = a[j]; });
    if (lists && mypos != row0) {                        // "row mypos holds what row row0 held": at most 2 nbc entries over the whole grid
      const int idx = atomicAdd(lists, 1);
      if (idx < 2 * LU_NB_MAX) { lists[1 + idx] = mypos; lists[1 + 2 * LU_NB_MAX + idx] = row0; }
    }
    // right half of a 64-column panel (lu_plan.hip, pair form): the rows that became this half's pivot rows leave their entries of
    // the LEFT half's columns [lcol0, lcol0 + 32) -- which nobody touches during this kernel -- in lrows[position - k0]: the block
    // L10 the step after the panel solves with, read there instead of from rows another workgroup of that step is permuting
    if (lrows && mypos >= k0 && mypos < k0 + nbc) {
      const dc* lsrc = A + (size_t)row0 * n + lcol0;
      dc* ldst = lrows + (size_t)(mypos - k0) * LU_REG_NB;
      static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; ldst[j] = lsrc[j]; });
    }
  }
}

// ------------------------------------------------------------------ row interchanges outside the panel
// One wavefront replays the panel's nb interchanges on an index map and emits (dst,src) row lists:
// after the sequence, row dst holds what row src held before it. m <= 2 nb entries.
//
// Blocks 1.. of the same launch invert the 32 x 32 unit-lower diagonal blocks of the panel's L11 (they are final once
// the panel kernel has ended): invd[d*32 + i][j] = (L_dd^-1)[i][j], padded with the identity beyond nb. Lane j carries
// column j of the inverse through the forward substitution; the multipliers are LDS broadcasts.
// LDS of lu_perm_kernel, shared by its two roles: the 16 x 16 blocks of one 32 x 32 diagonal block and of its inverse (26 KB),
// or the fold's index arrays (2 KB). Kept well under 31 KB: on a CU that holds two panel workgroups and a trailing-update
// workgroup that is what is left, and a launch that needs more waits for the update to drain (these launches are on
// every system's critical chain; with 35 KB they took 79 us on average and up to 4 ms instead of 17 us).
struct PermLds {
  union {
    struct { dc LA[16][17], LB[16][17], LC[16][17], XA[16][17], XB[16][17], XC[16][17]; } inv;
    struct { int top[LU_NB_MAX], ext_row[LU_NB_MAX], ext_src[LU_NB_MAX], piv[LU_NB_MAX]; } fold;
  };
};

template <bool WIDE>   // WIDE: called by a workgroup of more than 64 threads: threads >= 64 only meet the barriers
__device__ void lu_invert_diag32(PermLds& S, const dc* __restrict__ T, int ldt, int nb, int d, dc* __restrict__ invd) {
  const bool act = !WIDE || threadIdx.x < 64;
  // L = [A 0; C B] in 16 x 16 blocks:  L^-1 = [A^-1 0; -B^-1 C A^-1  B^-1].  Lanes 0-15 / 16-31 carry the columns of
  // A^-1 / B^-1 through a 16-row forward substitution (a quarter of the 32-row dependent chain); then lane (j, q)
  // forms rows 4q..4q+3 of column j of W = C A^-1 and of -B^-1 W.
  auto& V = S.inv;
  const int lane = threadIdx.x, base = d * 32;
  const int m = min(32, nb - base);
#pragma unroll 4
  for (int idx = lane; act && idx < 1024; idx += 64) {
    const int i = idx >> 5, k = idx & 31;
    if (i < 16 && k >= 16) continue;
    const dc v = (i < m && k < i) ? T[(size_t)(base + i) * ldt + base + k] : dc_make(0.0, 0.0);
    if (i < 16) V.LA[i][k] = v; else if (k >= 16) V.LB[i - 16][k - 16] = v; else V.LC[i - 16][k] = v;
  }
  __syncthreads();
  if (act && lane < 32) {
    // column j of A^-1 (lanes 0-15) / B^-1 (lanes 16-31): x_i = e_i - sum_{k<i} l_ik x_k, the x_k read back from this lane's
    // own column (registers are what this kernel must not need)
    const int j = lane & 15;
    dc (*Lm)[17] = lane < 16 ? V.LA : V.LB;
    dc (*Xm)[17] = lane < 16 ? V.XA : V.XB;
    for (int i = 0; i < 16; ++i) {
      dc acc = dc_make(i == j ? 1.0 : 0.0, 0.0);
      for (int k = j; k < i; ++k) {                       // x_k = 0 for k < j
        const dc t = Lm[i][k], xk = Xm[k][j];
        acc.re -= t.re * xk.re - t.im * xk.im; acc.im -= t.re * xk.im + t.im * xk.re;
      }
      Xm[i][j] = acc;
    }
  }
  __syncthreads();
  {
    const int j = lane & 15, q = (lane >> 4) & 3;
    dc w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = dc_make(0.0, 0.0);
#pragma unroll 1
    for (int k = 0; act && k < 16; ++k) {
      const dc a = (k >= j) ? V.XA[k][j] : dc_make(0.0, 0.0);          // A^-1 is lower triangular; above the diagonal XA was never written
#pragma unroll
      for (int r = 0; r < 4; ++r) { const dc c = V.LC[4 * q + r][k]; w[r].re += c.re * a.re - c.im * a.im; w[r].im += c.re * a.im + c.im * a.re; }
    }
    __syncthreads();                                      // LA is free now: W is parked there
    if (act) {
#pragma unroll
    for (int r = 0; r < 4; ++r) V.LA[4 * q + r][j] = w[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; act && r < 4; ++r) {
      const int i = 4 * q + r;
      dc acc = dc_make(0.0, 0.0);
#pragma unroll 2
      for (int k = 0; k <= i; ++k) {                      // B^-1 is lower triangular
        const dc bi = V.XB[i][k], wk = V.LA[k][j];
        acc.re -= bi.re * wk.re - bi.im * wk.im; acc.im -= bi.re * wk.im + bi.im * wk.re;
      }
      V.XC[i][j] = acc;
    }
  }
  __syncthreads();
#pragma unroll 4
  for (int idx = lane; act && idx < 1024; idx += 64) {
    const int i = idx >> 5, k = idx & 31;
    dc v = dc_make(0.0, 0.0);
    if (i < 16) { if (k <= i) v = V.XA[i][k]; }
    else if (k >= 16) { if (k <= i) v = V.XB[i - 16][k - 16]; }
    else v = V.XC[i - 16][k];
    invd[(size_t)base * 32 + idx] = v;
  }
}


// the panel's nb interchanges folded into (dst, src) row lists by ONE wavefront (threads >= 64 of a wider workgroup return at once)
__device__ void lu_fold_pivots(PermLds& S, const int* __restrict__ ipiv, int n, int k0, int nb, int* __restrict__ lists, unsigned* __restrict__ poison) {
  if (threadIdx.x >= 64) return;
  // An aborted panel (the plan's poison word is set) has no valid pivots: no rows are moved. The same for a pivot outside
  // [k0 + c, n) -- which a completed panel never produces: the plan is poisoned (code 2) instead of acting on it.
  if (poison && __hip_atomic_load(poison, RLX_AGENT) != 0u) { if (threadIdx.x == 0) lists[0] = 0; return; }
  int* top = S.fold.top;               // content of row k0+c
  int* ext_row = S.fold.ext_row;       // rows >= k0+nb that were touched
  int* ext_src = S.fold.ext_src;
  int* piv = S.fold.piv;               // the panel's pivots, fetched in one coalesced load (not one dependent load per column)
  const int lane = threadIdx.x;
  bool bad = false;
  for (int c = lane; c < nb; c += 64) { top[c] = k0 + c; const int pv = ipiv[k0 + c]; piv[c] = pv; bad = bad || pv < k0 + c || pv >= n; }
  if (__any(bad)) {
    if (lane == 0) { lists[0] = 0; if (poison) __hip_atomic_store(poison, 2u, RLX_AGENT); }
    return;
  }
  int next = 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int c = 0; c < nb; ++c) {
    const int p = piv[c];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (p == k0 + c) continue;
    if (p < k0 + nb) {
      if (lane == 0) { int t = top[c]; top[c] = top[p - k0]; top[p - k0] = t; }
    } else {
      // find p among the touched external rows (2 per lane)
      int hit = -1;
      for (int e = lane; e < next; e += 64) if (ext_row[e] == p) hit = e;
      unsigned long long m = __ballot(hit >= 0);
      int e;
      if (m) { int src_lane = __builtin_ctzll(m); e = __shfl(hit, src_lane, 64); }
      else { e = next; if (lane == 0) { ext_row[e] = p; ext_src[e] = p; } next += 1; }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (lane == 0) { int t = top[c]; top[c] = ext_src[e]; ext_src[e] = t; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  // compact: only rows whose content changed
  int* dst = lists + 1; int* src = lists + 1 + 2 * LU_NB_MAX;
  int m = 0;
  for (int base = 0; base < nb + next; base += 64) {
    int i = base + lane;
    int d = -1, s = -1;
    if (i < nb) { d = k0 + i; s = top[i]; }
    else if (i < nb + next) { d = ext_row[i - nb]; s = ext_src[i - nb]; }
    bool keep = (d >= 0) && (d != s);
    unsigned long long bm = __ballot(keep);
    if (keep) { int pos = m + __popcll(bm & lanemask_lt()); dst[pos] = d; src[pos] = s; }
    m += __popcll(bm);
  }
  if (lane == 0) lists[0] = m;
}

__global__ __launch_bounds__(64) void lu_perm_kernel(const int* __restrict__ ipiv, int n, int k0, int nb, int* __restrict__ lists /* [0]=m, dst[256], src[256] */,
                                                     const dc* __restrict__ T, int ldt, dc* __restrict__ invd, unsigned* __restrict__ poison) {
  __shared__ PermLds S;
  if (blockIdx.x > 0) { lu_invert_diag32<false>(S, T, ldt, nb, blockIdx.x - 1, invd); return; }
  lu_fold_pivots(S, ipiv, n, k0, nb, lists, poison);
}

// tmp[idx][q] = A[src[idx]][col(q)] over the column set [x0, x1) U [y0, y1) plus the nrhs RHS "columns"
__global__ __launch_bounds__(256) void lu_gather_rows_kernel(const dc* __restrict__ A, int n, const int* __restrict__ lists, dc* __restrict__ tmp, int tstride,
                                                             int x0, int x1, int y0, int y1, const dc* __restrict__ B, int nrhs) {
  const int m = lists[0];
  const int idx = blockIdx.y;
  if (idx >= m) return;
  const int s = lists[1 + 2 * LU_NB_MAX + idx];
  if (s < 0 || s >= n) return;
  const int nx = x1 - x0, nxy = nx + (y1 - y0), ncol = nxy + nrhs;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < ncol; q += gridDim.x * 256) {
    dc v;
    if (q < nxy) { const int col = q < nx ? x0 + q : y0 + (q - nx); v = A[(size_t)s * n + col]; }
    else v = B[(size_t)(q - nxy) * n + s];
    tmp[(size_t)idx * tstride + q] = v;
  }
}

__global__ __launch_bounds__(256) void lu_scatter_rows_kernel(dc* __restrict__ A, int n, const int* __restrict__ lists, const dc* __restrict__ tmp, int tstride,
                                                              int x0, int x1, int y0, int y1, dc* __restrict__ B, int nrhs) {
  const int m = lists[0];
  const int idx = blockIdx.y;
  if (idx >= m) return;
  const int d = lists[1 + idx];
  if (d < 0 || d >= n) return;
  const int nx = x1 - x0, nxy = nx + (y1 - y0), ncol = nxy + nrhs;
  for (int q = blockIdx.x * 256 + threadIdx.x; q < ncol; q += gridDim.x * 256) {
    const dc v = tmp[(size_t)idx * tstride + q];
    if (q < nxy) { const int col = q < nx ? x0 + q : y0 + (q - nx); A[(size_t)d * n + col] = v; }
    else B[(size_t)(q - nxy) * n + d] = v;
  }
}

// ------------------------------------------------------------------ one launch between two panels of a block column (round 3)
// For the columns [x0, x0 + ncols) right of panel (k0, nb <= 32) -- the rest of its block column: the panel's row interchanges
// and U = L11^-1 A[k0 : k0 + nb, columns], one workgroup per strip of 32 columns: the <= 2 nb moved rows of the strip are read
// into registers (every read before any write), the rows that land below the panel are written, the panel rows' entries are put
// together in LDS and go through a right-looking forward substitution with L11 (one barrier per row of L11: 32 x 32 entries per
// strip need no matrix cores). The LAST workgroup inverts the diagonal block of L11 for the main lane's MFMA triangular solves,
// beside the strips. Replaces four launches of the per-panel chain (lu_perm_kernel, gather, scatter, lu_trsm64_kernel).
struct LaneStepLds {
  union {
    struct { dc B[32][33], L[32][33]; int dst[2 * LU_NB_MAX], src[2 * LU_NB_MAX]; } strip;
    PermLds perm;
  };
};
__global__ __launch_bounds__(256) void lu_lane_step_kernel(dc* __restrict__ A, int n, int k0, int nb, const int* __restrict__ lists, int x0, int ncols,
                                                           dc* __restrict__ invd, const unsigned* __restrict__ poison) {
  __shared__ LaneStepLds S;
  const int tid = threadIdx.x;
  const int nstrips = (ncols + 31) / 32;
  if ((int)blockIdx.x >= nstrips) {                      // the extra workgroup: inverted diagonal block (identity-padded beyond nb)
    lu_invert_diag32<true>(S.perm, A + (size_t)k0 * n + k0, n, nb, 0, invd);
    return;
  }
  if (poison && __hip_atomic_load(poison, RLX_AGENT) != 0u) return;      // an aborted panel has no valid pivots: no rows are moved
  auto& V = S.strip;
  int m = lists[0];
  // a list longer than the 2 nb entries a panel of nb columns can produce (the register stash below holds 64) is a corrupted or stale
  // list: the panel counts as abandoned -- the plan is poisoned (MA_ERR_HIP at ma_lu_plan_status) and no rows move, as lu_fold_pivots
  // does for a pivot outside its range
  if (m < 0 || m > 2 * nb) { if (tid == 0 && poison) __hip_atomic_store(const_cast<unsigned*>(poison), 2u, RLX_AGENT); return; }
  for (int i = tid; i < m; i += 256) { V.dst[i] = lists[1 + i]; V.src[i] = lists[1 + 2 * LU_NB_MAX + i]; }
  const int c0 = x0 + 32 * (int)blockIdx.x;              // first column of the strip
  const int wcols = min(32, x0 + ncols - c0);
  // L11 (strictly lower part; the diagonal is 1) and the panel rows' entries as they stand
  for (int idx = tid; idx < 32 * 32; idx += 256) {
    const int i = idx >> 5, k = idx & 31;
    V.L[i][k] = (i < nb && k < i) ? A[(size_t)(k0 + i) * n + k0 + k] : dc_make(0.0, 0.0);
    V.B[i][k] = (i < nb && k < wcols) ? A[(size_t)(k0 + i) * n + c0 + k] : dc_make(0.0, 0.0);
  }
  __syncthreads();
  // moved rows: element e = (list entry, column) -> registers
  dc mv[8]; int md[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int e = tid + 256 * q, idx = e >> 5, col = e & 31;
    md[q] = -1; mv[q] = dc_make(0.0, 0.0);
    if (idx < m && col < wcols) {
      const int sr = V.src[idx], ds = V.dst[idx];
      if (sr >= 0 && sr < n && ds >= 0 && ds < n) { mv[q] = A[(size_t)sr * n + c0 + col]; md[q] = ds; }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every read of the strip has returned ...
  __syncthreads();                                       // ... in every thread, before the first write
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int col = (tid + 256 * q) & 31;
    if (md[q] < 0) continue;
    if (md[q] >= k0 && md[q] < k0 + nb) V.B[md[q] - k0][col] = mv[q];      // lands in the panel rows: joins the triangular solve
    else A[(size_t)md[q] * n + c0 + col] = mv[q];
  }
  __syncthreads();
  // right-looking forward substitution: row k is final once rows < k have been subtracted from it
  {
    const int col = tid & 31, r0 = tid >> 5;             // 8 rows at a time
    for (int k = 0; k + 1 < nb; ++k) {
      const dc xk = V.B[k][col];
      for (int i = k + 1 + r0; i < nb; i += 8) {
        const dc l = V.L[i][k];
        dc v = V.B[i][col];
        v.re = __builtin_fma(l.im, xk.im, __builtin_fma(-l.re, xk.re, v.re));
        v.im = __builtin_fma(-l.im, xk.re, __builtin_fma(-l.re, xk.im, v.im));
        V.B[i][col] = v;
      }
      __syncthreads();
    }
  }
  for (int idx = tid; idx < 32 * 32; idx += 256) {
    const int i = idx >> 5, k = idx & 31;
    if (i < nb && k < wcols) A[(size_t)(k0 + i) * n + c0 + k] = V.B[i][k];
  }
}

// The same step for a 64-column panel that was factored as TWO register half-panels (lu_plan.hip, launch_panel pair form): ONE launch
// for what the 64-column chain did in six (the right half's interchanges on the left half's columns, lu_perm_kernel, gather,
// scatter, lu_trsm64_kernel). Workgroups by role:
//   [0, nstrips)   a strip of 32 columns of [x0, x0 + ncols): the left half's interchange list, then the right half's (each: every
//                  read before any write; panel rows live in LDS), then U = L11^-1 B with the 64 x 64 L11 in three 32 x 32 pieces
//                  that share one LDS buffer (solve, subtract L10 X0, solve);
//   nstrips        the right half's interchanges on the left half's columns [k0, k0 + 32);
//   nstrips + 1, 2 the inverted diagonal blocks of L11 for the main lane;
//   nstrips + 3    the 64 pivots folded into the list the main lane applies left and right of the block.
struct LaneStep2Lds {
  union {
    struct { dc B[64][33], L[32][33]; int dst[2][2 * LU_REG_NB], src[2][2 * LU_REG_NB]; int m[2]; } strip;
    PermLds perm;
  };
};
__global__ __launch_bounds__(256) void lu_lane_step2_kernel(dc* __restrict__ A, int n, int k0, int nb, const int* __restrict__ lists1, const int* __restrict__ lists2,
                                                            int x0, int ncols, const int* __restrict__ ipiv, int* __restrict__ lists64, dc* __restrict__ invd,
                                                            unsigned* __restrict__ poison, const dc* __restrict__ l10 /* rows 32.. of L11 x columns 0..31, row-major 32 wide */) {
  __shared__ LaneStep2Lds S;
  const int tid = threadIdx.x;
  const int nstrips = (ncols + 31) / 32;
  const int role = (int)blockIdx.x - nstrips;
  if (role == 1 || role == 2) { if (32 * (role - 1) < nb) lu_invert_diag32<true>(S.perm, A + (size_t)k0 * n + k0, n, nb, role - 1, invd); return; }
  if (role == 3) { lu_fold_pivots(S.perm, ipiv, n, k0, nb, lists64, poison); return; }
  if (poison && __hip_atomic_load(poison, RLX_AGENT) != 0u) return;      // an aborted panel has no valid pivots: no rows are moved
  auto& V = S.strip;
  const int h1 = min(nb, LU_REG_NB);
  if (tid < 2) {
    int m = (tid == 0 ? lists1 : lists2)[0];
    if (tid == 1 && nb <= LU_REG_NB) m = 0;
    // (as in lu_lane_step_kernel: a list beyond what a half-panel can produce poisons the plan and moves no rows)
    if (m < 0 || m > 2 * LU_REG_NB) { m = -1; if (poison) __hip_atomic_store(poison, 2u, RLX_AGENT); }
    V.m[tid] = m;
  }
  __syncthreads();
  if (V.m[0] < 0 || V.m[1] < 0) return;
  for (int q = 0; q < 2; ++q) {
    const int* ls = q == 0 ? lists1 : lists2;
    for (int i = tid; i < V.m[q]; i += 256) { V.dst[q][i] = ls[1 + i]; V.src[q][i] = ls[1 + 2 * LU_NB_MAX + i]; }
  }
  const bool left = role == 0;                            // the left half's own columns: only the right half's interchanges, no solve
  const int c0 = left ? k0 : x0 + 32 * (int)blockIdx.x;
  const int wcols = left ? h1 : min(32, x0 + ncols - c0);
  if (!left)
    for (int idx = tid; idx < 64 * 32; idx += 256) {       // the panel rows' entries of the strip as they stand
      const int i = idx >> 5, k = idx & 31;
      V.B[i][k] = (i < nb && k < wcols) ? A[(size_t)(k0 + i) * n + c0 + k] : dc_make(0.0, 0.0);
    }
  __syncthreads();
  // a row of the strip: from LDS if it is a panel row of a solving strip, else from the matrix
  auto get = [&](int row, int col) -> dc { return (!left && row >= k0 && row < k0 + nb) ? V.B[row - k0][col] : A[(size_t)row * n + c0 + col]; };
  auto put = [&](int row, int col, dc v) { if (!left && row >= k0 && row < k0 + nb) V.B[row - k0][col] = v; else A[(size_t)row * n + c0 + col] = v; };
  for (int q = left ? 1 : 0; q < 2; ++q) {
    const int m = V.m[q];
    if (m == 0) continue;
    dc mv[8]; int md[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int e = tid + 256 * t, idx = e >> 5, col = e & 31;
      md[t] = -1; mv[t] = dc_make(0.0, 0.0);
      if (idx < m && col < wcols) {
        const int sr = V.src[q][idx], ds = V.dst[q][idx];
        if (sr >= 0 && sr < n && ds >= 0 && ds < n) { mv[t] = get(sr, col); md[t] = ds; }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // every read of the strip has returned ...
    __syncthreads();                                       // ... in every thread, before the first write
#pragma unroll
    for (int t = 0; t < 8; ++t) if (md[t] >= 0) put(md[t], (tid + 256 * t) & 31, mv[t]);
    __syncthreads();                                       // (workgroup-scope: the next list reads what this one wrote)
  }
  if (left) return;
  // U = L11^-1 B, right-looking, 32 rows of L11 at a time through one LDS buffer
  const int col = tid & 31, r0 = tid >> 5;
  auto load_L = [&](int rb, int cb) {                      // rows [32 rb, +32) x columns [32 cb, +32) of L11 (strictly lower part when rb == cb)
    __syncthreads();
    for (int idx = tid; idx < 32 * 32; idx += 256) {
      const int i = idx >> 5, k = idx & 31, gi = 32 * rb + i, gk = 32 * cb + k;
      // block (1, 0) lies in the left half's columns, whose rows the `left` workgroup of THIS launch is permuting: it comes from the
      // copy the right half's panel kernel left (its pivot rows' left-half entries)
      V.L[i][k] = (gi < nb && gk < nb && gk < gi) ? ((rb == 1 && cb == 0) ? l10[(size_t)i * LU_REG_NB + k] : A[(size_t)(k0 + gi) * n + k0 + gk]) : dc_make(0.0, 0.0);
    }
    __syncthreads();
  };
  auto solve32 = [&](int rb) {                             // rows [32 rb, +32) of B against the diagonal block in V.L
    const int rows = min(32, nb - 32 * rb);
    for (int k = 0; k + 1 < rows; ++k) {
      const dc xk = V.B[32 * rb + k][col];
      for (int i = k + 1 + r0; i < rows; i += 8) {
        const dc l = V.L[i][k];
        dc v = V.B[32 * rb + i][col];
        v.re = __builtin_fma(l.im, xk.im, __builtin_fma(-l.re, xk.re, v.re));
        v.im = __builtin_fma(-l.im, xk.re, __builtin_fma(-l.re, xk.im, v.im));
        V.B[32 * rb + i][col] = v;
      }
      __syncthreads();
    }
  };
  load_L(0, 0); solve32(0);
  if (nb > 32) {
    load_L(1, 0);
    for (int i = r0; i < nb - 32; i += 8) {                // B1 -= L10 X0
      dc v = V.B[32 + i][col];
      for (int k = 0; k < 32; ++k) {
        const dc l = V.L[i][k], xk = V.B[k][col];
        v.re = __builtin_fma(l.im, xk.im, __builtin_fma(-l.re, xk.re, v.re));
        v.im = __builtin_fma(-l.im, xk.re, __builtin_fma(-l.re, xk.im, v.im));
      }
      V.B[32 + i][col] = v;
    }
    load_L(1, 1); solve32(1);
  }
  __syncthreads();
  for (int idx = tid; idx < 64 * 32; idx += 256) {
    const int i = idx >> 5, k = idx & 31;
    if (i < nb && k < wcols) A[(size_t)(k0 + i) * n + c0 + k] = V.B[i][k];
  }
}

// ------------------------------------------------------------------ U12 = L11^-1 A12 on the f64 matrix cores
// Blocked forward substitution with inverted 32 x 32 diagonal blocks (lu_invert_diag32): per 32-row block
//   X_b = D_b^-1 B_b,   B_below -= L_(below,b) X_b,
// everything a 16x16x4 MFMA. One wavefront owns 16 columns and keeps its whole nb x 16 block of A12 in
// accumulator registers (8 tiles x re/im); a result tile's register r holds rows 4r..4r+3 in exactly the layout
// of the MFMA B operand for that k-slice, so solved rows feed the next products without leaving the registers.
// The A operands (L11's current 32-column slab with D_b^-1 in place of its diagonal block) are staged through
// LDS as separate re/im planes (pitch 34: conflict-free 8-byte reads). In place: a workgroup reads and writes
// its own columns only. The last workgroup of the launch (if nc2 > 0) does the same for the right-hand sides,
// addressed with their own strides (row stride 1, column stride ldb): the forward substitution rides along.
#define TM_PITCH 34
// The same solve for panels of <= 64 columns with the four 16-row tiles of the wavefront's block of A12 in NAMED accumulators
// (an indexed array of tiles makes the compiler carry the whole array through every update): few registers, so the
// wavefront fits beside two update wavefronts on a SIMD instead of queueing behind a running trailing update -- where these
// launches, which sit on every system's critical chain, took 0.2-0.35 ms instead of 20 us.
__global__ __launch_bounds__(128, 5) void lu_trsm64_kernel(const dc* __restrict__ T, int ldt, int nb, const dc* __restrict__ invd,
                                                           dc* __restrict__ X, size_t xrs, size_t xcs, int nc, int nmain,
                                                           dc* __restrict__ X2, size_t x2rs, size_t x2cs, int nc2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* Lre = reinterpret_cast<double*>(smem);
  double* Lim = Lre + 32 * TM_PITCH;                    // two planes of 32 rows: 17 KB, loaded three times (see below)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const bool extra = (int)blockIdx.x >= nmain;
  dc* Xp = extra ? X2 : X;
  const size_t rs = extra ? x2rs : xrs, cs = extra ? x2cs : xcs;
  const int ncols = extra ? nc2 : nc;
  const int c0 = extra ? wave * 16 : ((int)blockIdx.x * 2 + wave) * 16;
  const bool active = c0 < ncols;
  const int col = c0 + li;
  const int NT = (nb + 15) >> 4;
  const v4d zero = (v4d){0, 0, 0, 0};
  v4d b0r = zero, b0i = zero, b1r = zero, b1i = zero, b2r = zero, b2i = zero, b3r = zero, b3i = zero;
  auto load_tile = [&](int t, v4d& br, v4d& bi) {
    if (t < NT && active && col < ncols) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * t + lk + 4 * r;
        if (row < nb) { const dc v = Xp[(size_t)row * rs + (size_t)col * cs]; br[r] = v.re; bi[r] = v.im; }
      }
    }
  };
  auto store_tile = [&](int t, const v4d& br, const v4d& bi) {
    if (t < NT && active && col < ncols) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * t + lk + 4 * r;
        if (row < nb) Xp[(size_t)row * rs + (size_t)col * cs] = dc_make(br[r], bi[r]);
      }
    }
  };
  load_tile(0, b0r, b0i); load_tile(1, b1r, b1i);   // tiles 2 and 3 are fetched once tiles 0 and 1 are solved: 64 accumulator registers live at a time
  // 32 rows of LDS at a time -- the inverted diagonal block `blk` (what = 0) or L's rows 32..63 below diagonal block 0
  // (what = 1) -- so that the launch needs 17 KB, not 35: it has to fit on CUs that hold panel and update workgroups
  auto load_rows = [&](int what, int blk) {
    __syncthreads();
    for (int idx = tid; idx < 32 * 32; idx += 128) {
      const int rr = idx >> 5, c = idx & 31;
      dc v;
      if (what == 0) v = invd[((size_t)blk * 32 + rr) * 32 + c];
      else v = (32 + rr < nb && c < nb) ? T[(size_t)(32 + rr) * ldt + c] : dc_make(0.0, 0.0);
      Lre[rr * TM_PITCH + c] = v.re; Lim[rr * TM_PITCH + c] = v.im;
    }
    __syncthreads();
  };
  // (Ba; Bb) <- D (Ba; Bb) with the unit-lower-triangular inverse D: X1 = D[16:32, 0:32] (Ba; Bb), X0 = D[0:16, 0:16] Ba
  auto solve_pair = [&](v4d& ar_, v4d& ai_, v4d& br_, v4d& bi_) {
    v4d x0r = zero, x0i = zero, x1r = zero, x1i = zero;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const double ar = Lre[(16 + li) * TM_PITCH + ks * 4 + lk], ai = Lim[(16 + li) * TM_PITCH + ks * 4 + lk];
      const double br = (ks < 4) ? ar_[ks & 3] : br_[ks & 3], bi = (ks < 4) ? ai_[ks & 3] : bi_[ks & 3];
      x1r = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, x1r, 0, 0, 0);
      x1i = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, x1i, 0, 0, 0);
      x1r = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, x1r, 0, 0, 0);
      x1i = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, x1i, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double ar = Lre[li * TM_PITCH + ks * 4 + lk], ai = Lim[li * TM_PITCH + ks * 4 + lk];
      const double br = ar_[ks], bi = ai_[ks];
      x0r = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, x0r, 0, 0, 0);
      x0i = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, x0i, 0, 0, 0);
      x0r = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, x0r, 0, 0, 0);
      x0i = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, x0i, 0, 0, 0);
    }
    ar_ = x0r; ai_ = x0i; br_ = x1r; bi_ = x1i;
  };
  // Bt -= L[tile t rows, columns 0..31] (X0; X1); tile t (2 or 3) sits in LDS rows 16 (t - 2) + li
  auto update_tile = [&](int t, v4d& tr, v4d& ti, const v4d& x0r, const v4d& x0i, const v4d& x1r, const v4d& x1i) {
    const int lr = (t - 2) * 16 + li;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const double ar = Lre[lr * TM_PITCH + ks * 4 + lk], ai = Lim[lr * TM_PITCH + ks * 4 + lk];
      const double xr = (ks < 4) ? x0r[ks & 3] : x1r[ks & 3], xi = (ks < 4) ? x0i[ks & 3] : x1i[ks & 3];
      tr = __builtin_amdgcn_mfma_f64_16x16x4f64(-ar, xr, tr, 0, 0, 0);
      ti = __builtin_amdgcn_mfma_f64_16x16x4f64(-ar, xi, ti, 0, 0, 0);
      tr = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, xi, tr, 0, 0, 0);
      ti = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, xr, ti, 0, 0, 0);
    }
  };
  load_rows(0, 0);
  if (active) solve_pair(b0r, b0i, b1r, b1i);
  store_tile(0, b0r, b0i); store_tile(1, b1r, b1i);
  if (nb > 32) {
    load_tile(2, b2r, b2i); load_tile(3, b3r, b3i);
    load_rows(1, 0);
    if (active) {
      update_tile(2, b2r, b2i, b0r, b0i, b1r, b1i);
      if (NT > 3) update_tile(3, b3r, b3i, b0r, b0i, b1r, b1i);
    }
    load_rows(0, 1);
    if (active) solve_pair(b2r, b2i, b3r, b3i);
    store_tile(2, b2r, b2i); store_tile(3, b3r, b3i);
  }
}

// ------------------------------------------------------------------ triangular solves for the right-hand sides
// x <- T^-1 x for one vector of nb <= 128 entries; one wavefront per right-hand side, lane l owns
// rows l and l + 64 and streams its own rows of T (row-major: contiguous per lane); the solved
// entry is broadcast with a cross-lane read. UPPER=false: unit lower; UPPER=true: non-unit upper.
template <bool UPPER>
__global__ __launch_bounds__(64) void lu_trsv_kernel(const dc* __restrict__ T, int ldt, int nb, dc* __restrict__ B, size_t ldb) {
  const int lane = threadIdx.x;
  dc* b = B + (size_t)blockIdx.x * ldb;
  const int r0 = lane, r1 = lane + 64;
  dc v0 = r0 < nb ? b[r0] : dc_make(0.0, 0.0);
  dc v1 = r1 < nb ? b[r1] : dc_make(0.0, 0.0);
  const dc* T0 = T + (size_t)(r0 < nb ? r0 : 0) * ldt;
  const dc* T1 = T + (size_t)(r1 < nb ? r1 : 0) * ldt;
  if (!UPPER) {
    for (int p0 = 0; p0 < nb; p0 += 8) {
      dc t0[8], t1[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int p = min(p0 + q, nb - 1); t0[q] = T0[p]; t1[q] = T1[p]; }   // 16 loads in flight
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int p = p0 + q;
        if (p < nb) {
          const int src = p & 63;
          const double xr = __shfl(p < 64 ? v0.re : v1.re, src, 64), xi = __shfl(p < 64 ? v0.im : v1.im, src, 64);
          if (r0 > p && r0 < nb) { v0.re -= t0[q].re * xr - t0[q].im * xi; v0.im -= t0[q].re * xi + t0[q].im * xr; }
          if (r1 > p && r1 < nb) { v1.re -= t1[q].re * xr - t1[q].im * xi; v1.im -= t1[q].re * xi + t1[q].im * xr; }
        }
      }
    }
  } else {
    // reciprocals of this lane's two diagonal entries up front: the division leaves the dependent chain
    const dc rd0 = r0 < nb ? crecip(T0[r0]) : dc_make(0.0, 0.0);
    const dc rd1 = r1 < nb ? crecip(T1[r1]) : dc_make(0.0, 0.0);
    for (int p0 = nb - 1; p0 >= 0; p0 -= 8) {
      dc t0[8], t1[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int p = max(p0 - q, 0); t0[q] = T0[p]; t1[q] = T1[p]; }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int p = p0 - q;
        if (p >= 0) {
          const int src = p & 63;
          if (r0 == p) v0 = v0 * rd0;                         // the owner scales by 1 / u_pp first
          if (r1 == p) v1 = v1 * rd1;
          const double xr = __shfl(p < 64 ? v0.re : v1.re, src, 64), xi = __shfl(p < 64 ? v0.im : v1.im, src, 64);
          if (r0 < p) { v0.re -= t0[q].re * xr - t0[q].im * xi; v0.im -= t0[q].re * xi + t0[q].im * xr; }
          if (r1 < p && r1 < nb) { v1.re -= t1[q].re * xr - t1[q].im * xi; v1.im -= t1[q].re * xi + t1[q].im * xr; }
        }
      }
    }
  }
  if (r0 < nb) b[r0] = v0;
  if (r1 < nb) b[r1] = v1;
}

#define ZG_BK 8                              // k-slice of an LDS stage of the register-staged update kernel
// ------------------------------------------------------------------ C -= A * B, 3M form
// C -= A B (row-major, leading dimensions in elements) with three real products per complex product:
//   T1 = Ar Br, T2 = Ai Bi, T3 = (Ar + Ai)(Br + Bi);  Re = T1 - T2,  Im = T3 - T1 - T2.
// 25 % fewer matrix-core instructions than the 4-product form; normwise (not componentwise)
// backward stable, which is what the LU update needs. Workgroup tile 64 x 64, 4 wavefronts as
// 2 (M) x 2 (N), 32 x 32 per wavefront = 2 x 2 MFMA tiles x 3 accumulators. Two or three such workgroups
// share a CU (165 VGPRs, 32 KB LDS each), so that one's C read-modify-write epilogue runs under the
// others' MFMA loops (with one 128 x 64 workgroup per CU the epilogue cost 22 % of the kernel); the sums Ar+Ai and Br+Bi are formed once per fragment load.
#define Z3_BM 64
#define Z3_BN 64
#define Z3_STAGES 3                          // LDS stages of 16 KB: three workgroups of 48 KB share a CU

#ifndef MA_ZGEMM_MAXWAVES
#define MA_ZGEMM_MAXWAVES 2
#endif
// One tile per workgroup, tile = blockIdx. (Rounds 2-3 also had a form whose workgroups DREW their tiles XCD by XCD: half the fabric
// traffic, no time: profiles/r02_*_tiles_drawn.json.)
__device__ __forceinline__ void zgemm3m_body(int M, int N, int K, const dc* __restrict__ A, size_t lda, const dc* __restrict__ B, size_t ldb, dc* __restrict__ C, size_t ldc,
                                             dc (*As)[ZG_BK][Z3_BM], dc (*Bs)[ZG_BK][Z3_BN]) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  const int m0 = blockIdx.y * Z3_BM, n0 = blockIdx.x * Z3_BN;

  v4d t1[2][2], t2[2][2], t3[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) { t1[a][b] = (v4d){0, 0, 0, 0}; t2[a][b] = (v4d){0, 0, 0, 0}; t3[a][b] = (v4d){0, 0, 0, 0}; }

  dc ra[2], rb[2];
  auto load_stage = [&](int k0) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int e = tid + 256 * s;
      const int row = e >> 3, kk = e & 7;                       // A: 8 consecutive k of one row = 128 B
      const int gm = m0 + row, gk = k0 + kk;
      ra[s] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : dc_make(0.0, 0.0);
      const int bk = e >> 6, bn = e & 63;                       // B: 64 consecutive n of one k-row
      const int gn = n0 + bn, gk2 = k0 + bk;
      rb[s] = (gn < N && gk2 < K) ? B[(size_t)gk2 * ldb + gn] : dc_make(0.0, 0.0);
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int e = tid + 256 * s;
      As[buf][e & 7][(e >> 3) ^ (e & 7)] = ra[s];   // XOR swizzle (see zgemm_sub_kernel): conflict-free transposed store
      Bs[buf][e >> 6][e & 63] = rb[s];
    }
  };

  // Three LDS stages. The loads of stage st + 2 are issued at the top of iteration st and go to LDS at the top of iteration
  // st + 1: a whole stage of matrix-core work (x the wavefronts sharing the SIMD) lies between a load and the first instruction
  // that needs it, where the two-stage form waited for its loads at the END of the stage they were issued in (one stage is
  // 0.65 us of MFMA time, an L2 miss under load 1-2 us). The staging registers are the same 16: free again once written to LDS.
  // One barrier per stage as before: it publishes stage st + 1 (written an iteration ago by now) and fences the reuse of buffer
  // st mod 3, which is written next at the top of iteration st + 2.
  const int nstage = (K + ZG_BK - 1) / ZG_BK;
  load_stage(0);
  store_stage(0);
  if (nstage > 1) load_stage(ZG_BK);
  __syncthreads();
  int buf = 0;
  for (int st = 0; st < nstage; ++st) {
    const int nxt = buf == Z3_STAGES - 1 ? 0 : buf + 1;
    if (st + 1 < nstage) store_stage(nxt);
    if (st + 2 < nstage) load_stage((st + 2) * ZG_BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = ks * 4 + lk;
      dc af[2], bf[2];
      double as[2], bs[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) { af[a] = As[buf][kk][(wm * 32 + a * 16 + li) ^ kk]; as[a] = af[a].re + af[a].im; }
#pragma unroll
      for (int b = 0; b < 2; ++b) { bf[b] = Bs[buf][kk][wn * 32 + b * 16 + li]; bs[b] = bf[b].re + bf[b].im; }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          t1[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].re, bf[b].re, t1[a][b], 0, 0, 0);
          t2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].im, bf[b].im, t2[a][b], 0, 0, 0);
          t3[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[a], bs[b], t3[a][b], 0, 0, 0);
        }
    }
    __syncthreads();
    buf = nxt;
  }
  // C read-modify-write, eight entries of C in flight per lane: the loads of one half (a) of the wavefront's tile are issued
  // back to back from clamped (always valid) addresses, without a branch between them, then combined and stored under the bounds
  // test. (Written entry by entry -- load, wait, store, sixteen times -- the compiler kept that order, and the epilogue was sixteen
  // memory round trips per tile: as long as the K = 256 main loop itself.)
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    dc cv[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = min(m0 + wm * 32 + a * 16 + lk + 4 * r, M - 1);
        const int gn = min(n0 + wn * 32 + b * 16 + li, N - 1);
        cv[b][r] = C[(size_t)gm * ldc + gn];
      }
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm * 32 + a * 16 + lk + 4 * r;
        const int gn = n0 + wn * 32 + b * 16 + li;
        const double p1 = t1[a][b][r], p2 = t2[a][b][r];
        dc c = cv[b][r];
        c.re -= p1 - p2; c.im -= t3[a][b][r] - p1 - p2;
        if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = c;
      }
  }
}
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(1, MA_ZGEMM_MAXWAVES))) void zgemm3m_sub_kernel(int M, int N, int K, const dc* __restrict__ A, size_t lda,
                                                             const dc* __restrict__ B, size_t ldb, dc* __restrict__ C, size_t ldc) {
  __shared__ __attribute__((aligned(16))) dc As[Z3_STAGES][ZG_BK][Z3_BM];
  __shared__ __attribute__((aligned(16))) dc Bs[Z3_STAGES][ZG_BK][Z3_BN];
  zgemm3m_body(M, N, K, A, lda, B, ldb, C, ldc, As, Bs);
}
// ------------------------------------------------------------------ C -= A * B, 3M form, operands by LDS-DMA
// What the matrix cores lose in zgemm3m_sub_kernel is its LDS traffic (tools/mfma_loop_probe.hip, profiles/r03_mfma_loop_probe.txt:
// the bare loop of 24 MFMAs per stage runs at 76 TFLOP/s of 77; with the fragment reads 72; with the four ds_write_b128 of the
// register staging 63; with the four global loads 59 -- and three workgroups per CU hide none of it). This kernel halves that
// traffic per MFMA and takes the staging out of the register file:
//   - a wavefront computes 32 x 64 of C (2 x 4 MFMA tiles x 3 accumulators = 192 accumulation registers, in the AGPR half of the
//     file; 6 fragment reads per 24 MFMAs instead of 4 per 12), WM x WN wavefronts a tile of 32 WM x 64 WN;
//   - A and B tiles go global -> LDS by global_load_lds_dwordx4 (no staging registers, no ds_write), three stages of 8 k, the
//     DMAs of stage st + 2 issued at the top of iteration st and waited for (counted vmcnt) before the barrier that ends
//     iteration st + 1 -- one raw s_barrier per stage, nothing drains early;
//   - an LDS-DMA writes 64 lanes x 16 B CONTIGUOUSLY, so the layout that keeps the fragment reads conflict-free is made on the
//     SOURCE side: lane l of the DMA for rows 8c .. 8c + 7 of the A tile fetches (row 8c + (l & 7), k = 2 (l >> 4) + ((l >> 3) & 1)):
//     8 rows x 128 B per instruction, the same lines as a plain row-major load; an A fragment read of one ds_read_b128 lane group
//     ({0-3, 12-15, 20-27}: 8 rows x 2 consecutive k) then touches 16 different 16-B slots mod 16. B rows are k-major as they are.
// Edges: rows >= M and columns >= N are fetched from the last valid row / column (they only reach C entries that are not stored);
// K must be a multiple of 8 (the launcher falls back to zgemm3m_sub_kernel otherwise).
#define ZD_BK 8
__device__ __forceinline__ void zd_glds16(const void* g, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
#define ZD_MFMA(acc, x, y) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))

// BIG: the same code as a kernel of its own for the big trailing updates on the caller's stream (K = panels-per-update x 64, nine
// tenths of a factorisation's flops), so that a kernel trace tells them from the K = 32 / 64 in-block updates on the lanes.
template <int WM, int WN, bool BIG>
__global__ __launch_bounds__(64 * WM * WN, 512 / (64 * WM * WN)) void zgemm3m_dma_kernel(int M, int N, int K, const dc* __restrict__ A, size_t lda, const dc* __restrict__ B, size_t ldb,
                                                                   dc* __restrict__ C, size_t ldc, int xcd_order) {
  constexpr int NW = WM * WN, TM = 32 * WM, TN = 64 * WN;
  constexpr int ACH = TM / 8, BSEG = TN / 64, BCH = 8 * BSEG;        // 1-KiB pieces of one stage: 8 rows of A each / 64 columns of one k-row of B each
  constexpr int CA = ACH / NW, CB = BCH / NW;                          // pieces per wavefront per stage
  static_assert(ACH % NW == 0 && BCH % NW == 0, "pieces must divide among the wavefronts");
  constexpr int STAGE = (TM + TN) * ZD_BK;                             // entries per stage: A part, then B part
  extern __shared__ __attribute__((aligned(16))) dc zd_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 15, lk = lane >> 4;
  // Tile of this workgroup. xcd_order (a one-dimensional grid of 8 ceil(tiles / 8) workgroups) deals the tiles out XCD by XCD: workgroup b goes to XCD b mod 8
  // when the chip is free to place it so (a tendency, not a rule: every tile is computed exactly once whatever the placement),
  // XCD x takes the x-th eighth of the tile sequence, and that sequence runs through blocks of 4 x 4 tiles, so that the ~50
  // workgroups an XCD runs at a time share 4 row strips of A and 4 column strips of B per 16 tiles in its 4 MB of L2 instead of
  // fetching 2 strips per tile over the fabric.
  int m0, n0;
  if (!xcd_order) { m0 = blockIdx.y * TM; n0 = blockIdx.x * TN; }
  else {
    const int gx = (N + TN - 1) / TN, gy = (M + TM - 1) / TM, T = gx * gy;
    const int per = (T + 7) >> 3;
    const int t = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if (t >= T) return;                                                   // (uniform: before any barrier)
    const int bw = (gx + 3) >> 2;                                         // blocks of 4 x 4 tiles, block-row-major; the last column / row of blocks may be narrower
    const int full_rows = gy >> 2;                                        // block rows with 4 tile rows
    int ty, tx;
    if (t < full_rows * 4 * gx) {
      const int br = t / (4 * gx), r = t - br * 4 * gx;                   // block row, index inside it
      const int full_cols = gx >> 2;
      if (r < full_cols * 16) { const int bc = r >> 4, q = r & 15; ty = br * 4 + (q >> 2); tx = bc * 4 + (q & 3); }
      else { const int q = r - full_cols * 16, w = gx - full_cols * 4; ty = br * 4 + q / w; tx = full_cols * 4 + q % w; }
    } else { const int q = t - full_rows * 4 * gx; ty = full_rows * 4 + q / gx; tx = q % gx; }   // the last (short) block row: row-major
    (void)bw;
    m0 = ty * TM; n0 = tx * TN;
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)zd_lds;

  // per-lane sources of this wavefront's pieces (advanced by one stage per issue)
  const dc* pa[CA]; const dc* pb[CB];
#pragma unroll
  for (int j = 0; j < CA; ++j) {
    const int c = wave * CA + j;
    const int row = min(m0 + 8 * c + (lane & 7), M - 1);
    pa[j] = A + (size_t)row * lda + (((lane >> 4) << 1) | ((lane >> 3) & 1));
  }
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int q = wave * CB + j, k = q / BSEG, seg = q - k * BSEG;
    const int col = min(n0 + seg * 64 + lane, N - 1);
    pb[j] = B + (size_t)k * ldb + col;
  }
  auto issue = [&](int buf) {
    const unsigned base = lds0 + (unsigned)(buf * STAGE * (int)sizeof(dc));
#pragma unroll
    for (int j = 0; j < CA; ++j) { zd_glds16(pa[j], base + (unsigned)((wave * CA + j) * 1024)); pa[j] += ZD_BK; }
#pragma unroll
    for (int j = 0; j < CB; ++j) { zd_glds16(pb[j], base + (unsigned)(TM * ZD_BK * 16 + (wave * CB + j) * 1024)); pb[j] += (size_t)ZD_BK * ldb; }
  };

  v4d t1[2][4], t2[2][4], t3[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { t1[a][b] = (v4d){0, 0, 0, 0}; t2[a][b] = (v4d){0, 0, 0, 0}; t3[a][b] = (v4d){0, 0, 0, 0}; }

  // fragment positions (entries inside a stage): A piece (m >> 3), slot (k >> 1) 16 + (k & 1) 8 + (m & 7); B piece k BSEG + (n >> 6), slot n & 63
  int aoff[2][2], boff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int k = ks * 4 + lk;
#pragma unroll
    for (int a = 0; a < 2; ++a) { const int m = wm * 32 + a * 16 + li; aoff[ks][a] = (m >> 3) * 64 + (k >> 1) * 16 + (k & 1) * 8 + (m & 7); }
    boff[ks] = TM * ZD_BK + (k * BSEG + wn) * 64 + li;
  }

  const int nstage = K / ZD_BK;
  issue(0);
  if (nstage > 1) issue(1);
  if (nstage > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CA + CB) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int buf = 0;
  for (int st = 0; st < nstage; ++st) {
    const int nxt = buf == 2 ? 0 : buf + 1;
    if (st + 2 < nstage) issue(nxt == 2 ? 0 : nxt + 1);
    const dc* S = zd_lds + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      dc af[2], bf[4];
      double as[2], bs[4];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = S[aoff[ks][a]];
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = S[boff[ks] + b * 16];
#pragma unroll
      for (int a = 0; a < 2; ++a) as[a] = af[a].re + af[a].im;
#pragma unroll
      for (int b = 0; b < 4; ++b) bs[b] = bf[b].re + bf[b].im;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          ZD_MFMA(t1[a][b], af[a].re, bf[b].re);
          ZD_MFMA(t2[a][b], af[a].im, bf[b].im);
          ZD_MFMA(t3[a][b], as[a], bs[b]);
        }
    }
    // stage st + 1 has landed (the pieces of st + 2, issued above, may stay in flight); the barrier publishes it and fences buffer st mod 3
    if (st + 2 < nstage) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CA + CB) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf = nxt;
  }

  // C read-modify-write, eight entries in flight per lane: the loads of a quarter (a, two b) of the wavefront's tile are issued
  // back to back from clamped (always valid) addresses, then combined and stored under the bounds test (sixteen at a time
  // spilled 28 registers)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int bh = 0; bh < 2; ++bh) {
      dc cv[2][4];
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gm = min(m0 + wm * 32 + a * 16 + lk + 4 * r, M - 1);
          const int gn = min(n0 + wn * 64 + (bh * 2 + b2) * 16 + li, N - 1);
          cv[b2][r] = C[(size_t)gm * ldc + gn];
        }
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int b = bh * 2 + b2;
          const int gm = m0 + wm * 32 + a * 16 + lk + 4 * r;
          const int gn = n0 + wn * 64 + b * 16 + li;
          const double p1 = t1[a][b][r], p2 = t2[a][b][r];
          dc c = cv[b2][r];
          c.re -= p1 - p2; c.im -= t3[a][b][r] - p1 - p2;
          if (gm < M && gn < N) C[(size_t)gm * ldc + gn] = c;
        }
    }
}

// thin-N variant for the right-hand sides (N = nrhs small): y[m] -= sum_k A[m][k] x[k]; one wave per row
__global__ __launch_bounds__(256) void zgemv_sub_kernel(int M, int K, const dc* __restrict__ A, size_t lda, const dc* __restrict__ x, dc* __restrict__ y) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  if (m >= M) return;
  double sr = 0.0, si = 0.0;
  for (int k = lane; k < K; k += 64) {
    const dc a = A[(size_t)m * lda + k]; const dc v = x[k];
    sr += a.re * v.re - a.im * v.im; si += a.re * v.im + a.im * v.re;
  }
  sr = wave_sum(sr); si = wave_sum(si);
  if (lane == 0) { dc c = y[m]; c.re -= sr; c.im -= si; y[m] = c; }
}

// MFMA f64 issue-rate probe: each wave runs `iters` x 16 independent v_mfma_f64_16x16x4_f64
__global__ __launch_bounds__(256) void mfma_f64_probe_kernel(double* out, int iters) {
  // 12 independent accumulators (what one k-step of the update kernel issues), the instruction written out so that the
  // accumulators stay where they are (the builtin form moved them between the two register files every iteration: 34 TFLOP/s)
  v4d acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 12; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ------------------------------------------------------------------ launchers
// ---- Residency of the panel kernels (the argument behind the admission rule below)
// A panel kernel exchanges by spinning: none of its workgroups finishes before ALL of them are resident. Several panel
// kernels may be in flight on a device (the systems of a batch, other plans, other host threads), next to kernels that
// always terminate (updates, trsm, row moves). A spinning workgroup that is resident never leaves, so the question is
// whether every workgroup of every ADMITTED grid can always be placed once the terminating kernels have drained.
//   LDS is allocated to a workgroup as ONE contiguous range of a CU's 160 KB. A CU that holds j spinning workgroups of
//   at most s bytes each has 160 KB - j s free, in at most j + 1 holes (the other kernels' ranges come and go and leave
//   the spinning ones at arbitrary offsets), so its largest hole is >= (160 KB - j s) / (j + 1). That is >= s as long as
//   (2 j + 1) s <= 160 KB. With p(s) = floor((160 KB / s + 1) / 2) slots per CU, any CU holding fewer than p(s) spinning
//   workgroups can therefore ALWAYS take one more. If the admitted grids together have at most p(s_max) x ncu workgroups
//   (s_max: the largest of their LDS sizes), a workgroup can only be left without a place when every CU already holds
//   >= p of them, i.e. when all of them are resident. The same holds for the vector registers (a wave's registers are
//   one contiguous range of the SIMD's 512): p_v(r) = floor((512 / r + 1) / 2) waves of r registers.
// The first version of this rule (round 1, commit 9958a60) counted slots as floor(160 KB / s): with 32 rows of a
// 128-column panel per workgroup (MA_LU_RPB=32: s = 70 KB) it admitted two grids of 256 workgroups = 2 per CU, which only
// fit if every CU packs them at offsets 0 and 70 KB. A workgroup that landed behind a departing 32 KB update workgroup
// (offset 32 KB) left two holes of 32 and 58 KB: that CU could never take its second panel workgroup, the grids stayed
// partially resident and ran into the 4 s limit of the spin; the aborted panels left pivots unwritten (d_ipiv was not
// even initialised then) and the interchange kernels indexed rows with them -- the memory-access fault recorded in
// gpurun_out/bench_rpb32.log (a wild address, 0xc6290bf7a000). By the rule above p(70 KB) = 1.
// Now: (1) admission by p(s) and p_v(r), checked against the occupancy the runtime reports for the kernel;
// (2) a grid that does not fit the chip ALONE is refused with a status (never launched); (3) an expired wait poisons the
// plan (LuPanelWs::timeout): every poll reads the word, all workgroups of all the plan's panel kernels leave within one
// sweep, the columns not reached get identity pivots, lu_perm_kernel moves no rows for a poisoned plan and poisons it
// itself on any pivot outside [k0 + c, n); ma_lu_plan_status reports MA_ERR_HIP. No kernel indexes memory with a value
// read from an aborted panel.
namespace {
constexpr int kSeqRing = 256;                          // launches remembered per device
struct PanelLaunch { hipEvent_t ev = nullptr; hipStream_t st = nullptr; int nblk = 0; size_t lds = 0; int regs = 0; int ncu = 0; bool used = false; };
// Locking rule: ONE mutex per device guards that device's ring and count (host threads driving different devices never meet:
// ma_bem_solve_sweep_multi runs a thread per device, each launching ~300 panels per frequency); it is held from admit() to
// commit() -- admission and the launch it admits are one step for the other threads of the SAME device -- but never across a
// host wait: when the host is a whole ring ahead of the device, admit() drops the lock, waits for the old launch's event and
// looks again. The kernels' register counts and the occupancy checks (process-wide facts) sit behind a small mutex of their own.
struct DeviceSequencer {
  std::mutex mu;
  PanelLaunch ring[kSeqRing];
  bool made = false;
  unsigned long long count = 0;
};
DeviceSequencer g_seq_dev[16];
struct KernelFacts {
  std::mutex mu;
  int regs = 0;              // vector registers per lane of lu_panel_reg_kernel
  bool occ_checked_reg = false;
};
KernelFacts g_seq;
constexpr size_t kLdsPerCu = 160 * 1024;
}  // namespace

// slots per CU for spinning workgroups of `lds` bytes and `regs` vector registers per lane (see the argument above)
int lu_panel_slots_per_cu(size_t lds, int regs) {
  if (lds == 0) return 0;
  int p = (int)((kLdsPerCu / (double)lds + 1.0) / 2.0);
  if (regs > 0) { const int r8 = (regs + 7) & ~7; const int pv = (int)((512.0 / r8 + 1.0) / 2.0); if (pv < p) p = pv; }
  if (p > 4) p = 4;
  return p;
}

size_t lu_panel_granule_bytes(int max_blocks) { return sizeof(unsigned long long) * LU_GRANULE_STRIDE * (2 * (size_t)max_blocks + 2 * LU_GROUPS); }

int lu_panel_regs() {
  std::lock_guard<std::mutex> lock(g_seq.mu);
  if (g_seq.regs == 0) {
    hipFuncAttributes fa;
    MA_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(lu_panel_reg_kernel<LU_REG_NB>)));
    g_seq.regs = fa.numRegs > 0 ? fa.numRegs : 128;
  }
  return g_seq.regs;
}

// ---- the admission window as a guard any SPINNING kernel of the library goes through (LU panels, the flag-driven Gauss-Seidel
// sweep, the one-launch Gram-Schmidt step): admit() holds the sequencer until commit() has recorded the launch's event
int SpinLaunch::admit(hipStream_t st_, int nblk_, size_t lds_, int regs_, int ncu_) {
  int dev_ = 0;
  MA_HIP(hipGetDevice(&dev_));
  MA_REQUIRE(dev_ >= 0 && dev_ < 16, MA_ERR_UNSUPPORTED, "device index %d beyond the sequencer table", dev_);
  {
    const int p = lu_panel_slots_per_cu(lds_ ? lds_ : 1, regs_);
    MA_REQUIRE(p >= 1 && (long long)nblk_ <= (long long)p * ncu_, MA_ERR_UNSUPPORTED,
               "a spinning grid of %d workgroups (%zu B LDS, %d registers) cannot be co-resident on %d CUs (%d per CU)", nblk_, lds_, regs_, ncu_, p);
  }
  DeviceSequencer& D = g_seq_dev[dev_];
  D.mu.lock(); locked = true;
  dev = dev_; st = st_; nblk = nblk_; lds = lds_ ? lds_ : 1; regs = regs_; ncu = ncu_;
  if (!D.made) {
    for (int i = 0; i < kSeqRing; ++i) {
      const hipError_t e = hipEventCreateWithFlags(&D.ring[i].ev, hipEventDisableTiming);
      if (e != hipSuccess) { set_error("sequencer events: %s", hipGetErrorString(e)); abandon(); return MA_ERR_HIP; }
    }
    D.made = true;
  }
  // Admission. Streams are in order, so at most ONE spinning kernel per stream runs at any time, and what may run beside this
  // launch is, per other stream, one of that stream's earlier launches (later launches do their own admission and count this
  // one). Per other stream take the largest grid and LDS size among its launches of the last 256 launches; if this launch plus
  // one such grid per stream fits p(s_max) x ncu workgroups, nothing has to be waited for. Otherwise the launch waits for the
  // LATEST launch of the stream whose latest launch is oldest (that stream then contributes nothing: all its earlier launches are
  // over when this kernel starts), and so on until the rest fits. For equal shapes on three lanes this is "wait for the launch
  // before the previous one"; the running set is always bounded by what its newest member computed, so the residency argument
  // above applies to it. With CU-masked streams in the window the CUs counted are those of the SMALLEST set any member may use
  // (grids on a mask share its CUs with every unmasked grid): a grid that needs more than that runs on its own, which the check
  // above has already allowed.
  PanelLaunch* ring = D.ring;
  // an event counts as pending only while the runtime says "not ready": anything else (success, or an error because the
  // stream it was recorded on has been destroyed since -- plans come and go, the table is per device) means its work is over
  auto pending = [](hipEvent_t ev) { const hipError_t q = hipEventQuery(ev); (void)hipGetLastError(); return q == hipErrorNotReady; };
  // launch i - 256 not finished yet: the host is that far ahead of the device. The wait happens WITHOUT the lock (the ring's
  // events live as long as the process; another thread of this device may launch meanwhile, so the slot is looked up again)
  for (;;) {
    PanelLaunch& slot = ring[D.count % kSeqRing];
    if (!(slot.used && pending(slot.ev))) break;
    const hipEvent_t ev = slot.ev;
    D.mu.unlock();
    (void)hipEventSynchronize(ev);
    (void)hipGetLastError();
    D.mu.lock();
  }
  const unsigned long long i = D.count;
  const unsigned long long oldest = i >= (unsigned long long)(kSeqRing - 1) ? i - (kSeqRing - 1) : 0ull;
  struct Lane { hipStream_t st; unsigned long long latest; long long nblk; size_t lds; int regs; int ncu; };
  Lane lanes[16]; int nlanes = 0;
  for (unsigned long long j = i; j-- > oldest;) {             // newest first: the first hit of a stream is its latest launch
    const PanelLaunch& L = ring[j % kSeqRing];
    if (!L.used || L.st == st) continue;
    int q = 0;
    while (q < nlanes && lanes[q].st != L.st) ++q;
    if (q == nlanes) { if (nlanes == 16) continue; lanes[nlanes++] = {L.st, j, L.nblk, L.lds, L.regs, L.ncu}; }
    else { if (L.nblk > lanes[q].nblk) lanes[q].nblk = L.nblk; if (L.lds > lanes[q].lds) lanes[q].lds = L.lds; if (L.regs > lanes[q].regs) lanes[q].regs = L.regs; if (L.ncu < lanes[q].ncu) lanes[q].ncu = L.ncu; }
  }
  bool pruned = false;
  for (;;) {
    long long tot = nblk; size_t smax = lds; int rmax = regs; int cmin = ncu; int victim = -1;
    for (int q = 0; q < nlanes; ++q) {
      if (!lanes[q].st) continue;
      tot += lanes[q].nblk; if (lanes[q].lds > smax) smax = lanes[q].lds; if (lanes[q].regs > rmax) rmax = lanes[q].regs; if (lanes[q].ncu < cmin) cmin = lanes[q].ncu;
      if (victim < 0 || lanes[q].latest < lanes[victim].latest) victim = q;
    }
    if (victim < 0 || tot <= (long long)lu_panel_slots_per_cu(smax, rmax) * cmin) break;
    if (!pruned) {                                            // over capacity: forget the streams whose latest launch is over (idle lanes, plans of the past)
      pruned = true;
      for (int q = 0; q < nlanes; ++q) if (lanes[q].st && !pending(ring[lanes[q].latest % kSeqRing].ev)) lanes[q].st = nullptr;
      continue;
    }
    const PanelLaunch& L = ring[lanes[victim].latest % kSeqRing];
    if (pending(L.ev)) {
      const hipError_t e = hipStreamWaitEvent(st, L.ev, 0);
      if (e != hipSuccess) { set_error("hipStreamWaitEvent failed: %s", hipGetErrorString(e)); abandon(); return MA_ERR_HIP; }
    }
    lanes[victim].st = nullptr;                               // everything that stream launched before is over when this kernel starts
  }
  return MA_OK;
}
int SpinLaunch::commit() {
  if (!locked) return MA_OK;
  DeviceSequencer& D = g_seq_dev[dev];
  const unsigned long long i = D.count;
  PanelLaunch& me = D.ring[i % kSeqRing];
  const hipError_t e = hipEventRecord(me.ev, st);
  if (e == hipSuccess) { me.nblk = nblk; me.lds = lds; me.regs = regs; me.ncu = ncu; me.st = st; me.used = true; D.count = i + 1; }
  else set_error("hipEventRecord failed: %s", hipGetErrorString(e));
  abandon();
  return e == hipSuccess ? MA_OK : MA_ERR_HIP;
}
void SpinLaunch::abandon() { if (locked) { locked = false; g_seq_dev[dev].mu.unlock(); } }
SpinLaunch::~SpinLaunch() { abandon(); }

// device-wide "a spinning kernel gave up a wait" word: every such kernel raises it beside its own status word, every Krylov driver
// reads (and clears) it once at its end -- a result computed across an abandoned wait is never returned with MA_OK
namespace { unsigned* g_spin_err[16] = {}; std::mutex g_spin_err_mu; }
unsigned* spin_error_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lock(g_spin_err_mu);
  if (!g_spin_err[dev]) {
    if (hipMalloc(&g_spin_err[dev], 64) != hipSuccess) { (void)hipGetLastError(); g_spin_err[dev] = nullptr; return nullptr; }
    (void)hipMemset(g_spin_err[dev], 0, 64);
    (void)hipDeviceSynchronize();
  }
  return g_spin_err[dev];
}
int spin_error_check(const char* what) {
  unsigned* w = spin_error_word();
  if (!w) return MA_OK;
  unsigned v = 0;
  MA_HIP(hipMemcpy(&v, w, sizeof(unsigned), hipMemcpyDeviceToHost));
  if (!v) return MA_OK;
  (void)hipMemset(w, 0, sizeof(unsigned));
  set_error("%s: a kernel that exchanges between its workgroups abandoned a wait (its grid was not co-resident within 2 s); the result is not valid", what);
  return MA_ERR_HIP;
}

// LDS the register panel kernel declares (static: pivot row, staging row, a few words)
static size_t lu_panel_reg_lds() { return 2 * (size_t)LU_REG_NB * sizeof(dc) + 64; }

// admission + launch of the register-resident panel kernel: 256 rows per workgroup, nb <= LU_REG_NB columns; `ncu` = the CUs `st` may use
int lu_launch_panel_reg(c64* A, int n, int k0, int nb, int nblk, int ncu, const LuPanelWs& ws, int* ipiv, int* lists, bool clear_tags, hipStream_t st, c64* lrows, int lcol0,
                        const int* run_if_nonzero) {
  MA_REQUIRE(!lrows || (lcol0 >= 0 && lcol0 + LU_REG_NB <= k0), MA_ERR_INVALID, "left-half columns [%d, %d) not left of the panel at %d", lcol0, lcol0 + LU_REG_NB, k0);
  int dev = 0;
  MA_HIP(hipGetDevice(&dev));
  MA_REQUIRE(dev >= 0 && dev < 16, MA_ERR_UNSUPPORTED, "device index %d beyond the panel sequencer table", dev);
  MA_REQUIRE(nb >= 1 && nb <= LU_REG_NB, MA_ERR_INVALID, "panel of %d columns", nb);
  MA_REQUIRE(nblk >= 1 && nblk <= ws.max_blocks, MA_ERR_INVALID, "panel grid of %d workgroups outside the workspace (%d)", nblk, ws.max_blocks);
  MA_REQUIRE((long long)k0 + (long long)(nblk - 1) * 256 < n && (long long)k0 + (long long)nblk * 256 >= n, MA_ERR_INVALID,
             "panel grid (%d workgroups of 256 rows from row %d) does not tile the %d rows", nblk, k0, n);
  MA_REQUIRE(n < 0xFFFFFF, MA_ERR_UNSUPPORTED, "row positions travel in 24 bits of the exchange granule");
  const size_t lds = lu_panel_reg_lds();
  const int regs = lu_panel_regs();
  {
    const int p = lu_panel_slots_per_cu(lds, regs);
    MA_REQUIRE(p >= 1 && (long long)nblk <= (long long)p * ncu, MA_ERR_UNSUPPORTED,
               "panel grid of %d workgroups (%d columns) cannot be co-resident on %d CUs (%d per CU)", nblk, nb, ncu, p);
  }
  {
    std::lock_guard<std::mutex> lock(g_seq.mu);
    if (!g_seq.occ_checked_reg) {
      // the runtime's own occupancy figure must not be below the slots the rule assumes (registers, waves, LDS granularity)
      int occ = 0;
      MA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(lu_panel_reg_kernel<LU_REG_NB>), 256, 0));
      MA_REQUIRE(occ >= lu_panel_slots_per_cu(lds, regs), MA_ERR_UNSUPPORTED, "panel kernel occupancy %d per CU is below the %d slots the admission rule assumes", occ, lu_panel_slots_per_cu(lds, regs));
      g_seq.occ_checked_reg = true;
    }
  }
  SpinLaunch guard;
  { const int arc = guard.admit(st, nblk, lds, regs, ncu); if (arc) return arc; }
  // Stale tags must not match. A workgroup rewrites its granule every column, so only columns 0 and 1 of a launch can
  // see the previous launch's granules, which carry that launch's last two tags (nb and nb - 1): they differ from the
  // wanted 1 and 2 whenever the previous panel of this workspace had >= 4 columns. Otherwise (and at the start of a
  // factorisation, whose predecessor may have been aborted) the granules are cleared.
  if (clear_tags) MA_HIP(hipMemsetAsync(ws.cand, 0, lu_panel_granule_bytes(ws.max_blocks), st));
  hipLaunchKernelGGL(lu_panel_reg_kernel<LU_REG_NB>, dim3(nblk), dim3(256), 0, st, reinterpret_cast<dc*>(A), n, k0, nb, ws, ipiv, lists, reinterpret_cast<dc*>(lrows), lcol0, run_if_nonzero);
  MA_HIP(hipGetLastError());
  return guard.commit();
}

// a stream is about to be destroyed (its work is over): the sequencer must not wait on, or query, events recorded on it
void lu_panel_forget_stream(int dev, hipStream_t st) {
  if (dev < 0 || dev >= 16 || !st) return;
  DeviceSequencer& D = g_seq_dev[dev];
  std::lock_guard<std::mutex> lock(D.mu);
  if (!D.made) return;
  for (int i = 0; i < kSeqRing; ++i) if (D.ring[i].used && D.ring[i].st == st) { D.ring[i].used = false; D.ring[i].st = nullptr; }
}

// the step between two panels of a block column: interchanges + U = L11^-1 A12 on the columns [x0, x0 + ncols), and the inverted
// diagonal block of L11 into invd (lists: what lu_launch_panel_reg wrote)
int lu_launch_lane_step(c64* A, int n, int k0, int nb, const int* lists, int x0, int ncols, c64* invd, const unsigned* poison, hipStream_t st) {
  MA_REQUIRE(nb >= 1 && nb <= 32 && k0 >= 0 && k0 + nb <= n && ncols >= 0 && x0 >= 0 && x0 + ncols <= n, MA_ERR_INVALID, "lane step outside the matrix");
  hipLaunchKernelGGL(lu_lane_step_kernel, dim3((ncols + 31) / 32 + 1), dim3(256), 0, st, reinterpret_cast<dc*>(A), n, k0, nb, lists, x0, ncols, reinterpret_cast<dc*>(invd), poison);
  MA_HIP(hipGetLastError());
  return MA_OK;
}
// MA_OK when a register-panel grid of nblk workgroups can be co-resident on ncu CUs on its own
int lu_panel_reg_admissible(int nblk, int ncu) {
  const int p = lu_panel_slots_per_cu(lu_panel_reg_lds(), lu_panel_regs());
  MA_REQUIRE(p >= 1 && (long long)nblk <= (long long)p * ncu, MA_ERR_UNSUPPORTED, "register panel grid of %d workgroups cannot be co-resident on %d CUs (%d per CU)", nblk, ncu, p);
  return MA_OK;
}
// Apply panel (k0, nb)'s interchanges to the columns [x0, x1) U [y0, y1) of A and to the nrhs right-hand sides.
// `tmp` holds 2 nb rows of `tstride` >= (x1-x0)+(y1-y0)+nrhs entries. With `invd`, blocks 1.. of the first launch
// also invert the 32 x 32 diagonal blocks of the panel's L11 for lu_trsm_mfma_kernel.
// fold the panel's interchange sequence into its gather lists and invert the 32 x 32 diagonal blocks of L11
int lu_launch_perm(const c64* A, int n, int k0, int nb, const int* ipiv, int* lists, c64* invd, unsigned* poison, hipStream_t st) {
  MA_REQUIRE(nb >= 1 && nb <= LU_NB_MAX && k0 >= 0 && k0 + nb <= n, MA_ERR_INVALID, "panel [%d, %d) outside 0..%d", k0, k0 + nb, n);
  hipLaunchKernelGGL(lu_perm_kernel, dim3(invd ? 1 + (nb + 31) / 32 : 1), dim3(64), 0, st, ipiv, n, k0, nb, lists,
                     reinterpret_cast<const dc*>(A + (size_t)k0 * n + k0), n, reinterpret_cast<dc*>(invd), poison);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// apply the lists of lu_launch_perm to the columns [x0, x1) U [y0, y1) and to the right-hand sides
int lu_launch_row_moves(c64* A, int n, int nb, const int* lists, c64* tmp, int tstride, int x0, int x1, int y0, int y1, c64* B, int nrhs, hipStream_t st) {
  const int ncol = (x1 - x0) + (y1 - y0) + nrhs;
  if (ncol <= 0) return MA_OK;
  MA_REQUIRE(ncol <= tstride, MA_ERR_INVALID, "interchange staging rows too short (%d > %d)", ncol, tstride);
  int gx = (ncol + 255) / 256; if (gx > 64) gx = 64;
  dim3 grid(gx, 2 * nb);
  hipLaunchKernelGGL(lu_gather_rows_kernel, grid, dim3(256), 0, st, reinterpret_cast<const dc*>(A), n, lists, reinterpret_cast<dc*>(tmp), tstride, x0, x1, y0, y1,
                     reinterpret_cast<const dc*>(B), nrhs);
  MA_HIP(hipGetLastError());
  hipLaunchKernelGGL(lu_scatter_rows_kernel, grid, dim3(256), 0, st, reinterpret_cast<dc*>(A), n, lists, reinterpret_cast<const dc*>(tmp), tstride, x0, x1, y0, y1,
                     reinterpret_cast<dc*>(B), nrhs);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// the folded lists of a block's np panels (lists + j * lstride) applied in order to the columns [x0, x1) U [y0, y1) and the right-hand
// sides: one launch (lu_block_row_moves_kernel). Panel j's list goes to a column c of [x0, x1) only if c < k0s[j].

int lu_launch_swaps(c64* A, int n, int k0, int nb, const int* ipiv, int* lists, c64* tmp, int tstride, int x0, int x1, int y0, int y1, c64* B, int nrhs,
                    c64* invd, unsigned* poison, hipStream_t st) {
  int rc = lu_launch_perm(A, n, k0, nb, ipiv, lists, invd, poison, st);
  if (rc) return rc;
  return lu_launch_row_moves(A, n, nb, lists, tmp, tstride, x0, x1, y0, y1, B, nrhs, st);
}

int lu_trsm_configure() { return MA_OK; }

// X (nb <= 64 rows x ncols, row stride ldx) <- L11^-1 X and the nrhs right-hand sides b_r = B + r*ldb (nb entries each) <- L11^-1 b_r,
// with the inverted diagonal blocks `invd` of lu_launch_lane_step2 / lu_launch_swaps
int lu_launch_trsm_mfma(const c64* T, int ldt, int nb, const c64* invd, c64* X, size_t ldx, int ncols, c64* B, size_t ldb, int nrhs, hipStream_t st) {
  if (nb <= 0 || (ncols <= 0 && nrhs <= 0)) return MA_OK;
  MA_REQUIRE(nb <= 64 && nrhs <= 32, MA_ERR_DIM, "trsm: nb %d / nrhs %d beyond the kernel's tiles", nb, nrhs);
  const int nmain = ncols > 0 ? (ncols + 31) / 32 : 0;
  // LDS for a 32-row slab (17 KB): the launch fits on a CU that already holds panel and update workgroups
  const size_t lds = 2 * (size_t)32 * TM_PITCH * 8;
  hipLaunchKernelGGL(lu_trsm64_kernel, dim3(nmain + (nrhs > 0 ? 1 : 0)), dim3(128), lds, st, reinterpret_cast<const dc*>(T), ldt, nb,
                     reinterpret_cast<const dc*>(invd), reinterpret_cast<dc*>(X), ldx, (size_t)1, ncols, nmain, reinterpret_cast<dc*>(B), (size_t)1, ldb, nrhs);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// the step after a 64-column panel factored as two register half-panels: both halves' interchanges + U12 on [x0, x0 + ncols), the
// right half's interchanges on the left half's columns, the inverted diagonal blocks and the folded 64-pivot list for the main lane
int lu_launch_lane_step2(c64* A, int n, int k0, int nb, const int* lists1, const int* lists2, int x0, int ncols, const int* ipiv, int* lists64, c64* invd, unsigned* poison,
                         const c64* l10, hipStream_t st) {
  MA_REQUIRE(nb >= 1 && nb <= 2 * LU_REG_NB && k0 >= 0 && k0 + nb <= n && ncols >= 0 && x0 >= 0 && x0 + ncols <= n, MA_ERR_INVALID, "lane step outside the matrix");
  hipLaunchKernelGGL(lu_lane_step2_kernel, dim3((ncols + 31) / 32 + 4), dim3(256), 0, st, reinterpret_cast<dc*>(A), n, k0, nb, lists1, lists2, x0, ncols, ipiv, lists64,
                     reinterpret_cast<dc*>(invd), poison, reinterpret_cast<const dc*>(l10));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// nrhs vectors b_r = B + r*ldb (nb entries each): b_r <- T^-1 b_r
int lu_launch_trsv(bool upper, const c64* T, int ldt, int nb, c64* B, size_t ldb, int nrhs, hipStream_t st) {
  if (nrhs <= 0 || nb <= 0) return MA_OK;
  if (upper) hipLaunchKernelGGL(lu_trsv_kernel<true>, dim3(nrhs), dim3(64), 0, st, reinterpret_cast<const dc*>(T), ldt, nb, reinterpret_cast<dc*>(B), ldb);
  else hipLaunchKernelGGL(lu_trsv_kernel<false>, dim3(nrhs), dim3(64), 0, st, reinterpret_cast<const dc*>(T), ldt, nb, reinterpret_cast<dc*>(B), ldb);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// C -= A B. big: the trailing update on the caller's stream (its own kernel instantiation, so that a trace tells it from the lanes'
// updates). dma = false: the register-staged kernel (what K not a multiple of 8 gets anyway); the two are bit-identical.
int lu_launch_zgemm_sub(int M, int N, int K, const c64* A, size_t lda, const c64* B, size_t ldb, c64* C, size_t ldc, hipStream_t st, bool big, bool dma) {
  if (M <= 0 || N <= 0 || K <= 0) return MA_OK;
  if (dma && K % ZD_BK == 0) {
    // more than 64 KB of LDS per workgroup: the limit is raised per function AND per device (a process may drive several)
    {
      static std::mutex mu;
      static bool done[16] = {};
      int dev = 0;
      MA_HIP(hipGetDevice(&dev));
      MA_REQUIRE(dev >= 0 && dev < 16, MA_ERR_UNSUPPORTED, "device index %d beyond the table", dev);
      std::lock_guard<std::mutex> lock(mu);
      if (!done[dev]) {
        MA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(zgemm3m_dma_kernel<2, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (64 + 128) * ZD_BK * 16));
        MA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(zgemm3m_dma_kernel<2, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (64 + 128) * ZD_BK * 16));
        done[dev] = true;
      }
    }
    // large updates: a one-dimensional grid, tiles dealt out XCD by XCD in blocks of 4 x 4 (see the kernel)
    const int T22 = ((N + 127) / 128) * ((M + 63) / 64);
    const bool xcd_order = T22 >= 512;
    const dim3 g22 = xcd_order ? dim3(8 * ((T22 + 7) / 8), 1) : dim3((N + 127) / 128, (M + 63) / 64);
    if (big) hipLaunchKernelGGL((zgemm3m_dma_kernel<2, 2, true>), g22, dim3(256), 3 * (64 + 128) * ZD_BK * 16, st, M, N, K,
                                reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(B), ldb, reinterpret_cast<dc*>(C), ldc, xcd_order ? 1 : 0);
    else hipLaunchKernelGGL((zgemm3m_dma_kernel<2, 2, false>), g22, dim3(256), 3 * (64 + 128) * ZD_BK * 16, st, M, N, K,
                            reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(B), ldb, reinterpret_cast<dc*>(C), ldc, xcd_order ? 1 : 0);
    MA_HIP(hipGetLastError());
    return MA_OK;
  }
  dim3 g3((N + Z3_BN - 1) / Z3_BN, (M + Z3_BM - 1) / Z3_BM);
  hipLaunchKernelGGL(zgemm3m_sub_kernel, g3, dim3(256), 0, st, M, N, K, reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(B), ldb, reinterpret_cast<dc*>(C), ldc);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int lu_launch_zgemv_sub(int M, int K, const c64* A, size_t lda, const c64* x, c64* y, hipStream_t st) {
  if (M <= 0 || K <= 0) return MA_OK;
  hipLaunchKernelGGL(zgemv_sub_kernel, dim3((M + 3) / 4), dim3(256), 0, st, M, K, reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(x),
                     reinterpret_cast<dc*>(y));
  MA_HIP(hipGetLastError());
  return MA_OK;
}

// ---- does a CU-masked stream do what the plan assumes? (round 4: the mask's bit layout was taken from tools/cumask_probe.hip and never
// checked at run time.) A census: 4096 one-wavefront workgroups that each stay ~20 us record the (XCC, SE, SH, CU) they run on; the stream
// must have used exactly `expect_cus` different CUs, the same number in each of the 8 XCDs. Once per (device, expectation).
__global__ __launch_bounds__(64) void lu_cu_census_kernel(unsigned* __restrict__ out, int spin_ticks) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (u64)spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu);     // cu_id[11:8], sh_id[12], se_id[15:13] of HW_ID
}
int lu_cumask_selfcheck(hipStream_t masked, int expect_cus, bool* ok) {
  *ok = false;
  int dev = 0;
  MA_HIP(hipGetDevice(&dev));
  static std::mutex mu;
  static std::vector<std::pair<long long, bool>> seen;
  const long long key = (long long)dev * 100000 + expect_cus;
  {
    std::lock_guard<std::mutex> lock(mu);
    for (auto& e : seen) if (e.first == key) { *ok = e.second; return MA_OK; }
  }
  constexpr int NB = 4096;
  unsigned* d = nullptr;
  MA_HIP(hipMalloc(&d, sizeof(unsigned) * NB));
  MA_HIP(hipDeviceSynchronize());                            // the census counts the CUs that take its workgroups: on an otherwise idle device
  hipLaunchKernelGGL(lu_cu_census_kernel, dim3(NB), dim3(64), 0, masked, d, 2000);          // 20 us at 100 MHz
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(masked);
  std::vector<unsigned> h(NB);
  if (e == hipSuccess) e = hipMemcpy(h.data(), d, sizeof(unsigned) * NB, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { set_error("CU-mask census failed: %s", hipGetErrorString(e)); return MA_ERR_HIP; }
  std::sort(h.begin(), h.end());
  h.erase(std::unique(h.begin(), h.end()), h.end());
  int per_xcc[16] = {};
  for (unsigned v : h) per_xcc[(v >> 8) & 0xf] += 1;
  // the mask holds if no more than the expected CUs were used (an ignored mask shows all of them), none of the XCDs more than its share,
  // and the census did not miss more than a CU per XCD (a CU the dispatcher happened to skip must not cost the plan its schedule)
  bool good = expect_cus % 8 == 0 && (int)h.size() <= expect_cus && (int)h.size() >= expect_cus - 8;
  for (int x = 0; x < 8 && good; ++x) good = per_xcc[x] <= expect_cus / 8 && per_xcc[x] >= expect_cus / 8 - 1;
  // only a POSITIVE verdict is remembered: a census that ran while another host thread's sweep occupied the device may
  // miss CUs; a plan that gets a negative one runs unsplit (ma_lu_plan_cu_split reports 0 CUs), the next plan asks again
  if (good) { std::lock_guard<std::mutex> lock(mu); seen.push_back({key, true}); }
  *ok = good;
  return MA_OK;
}

int lu_launch_mfma_probe(double* out, int blocks, int iters, hipStream_t st) {
  hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, st, out, iters);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

}  // namespace ma
