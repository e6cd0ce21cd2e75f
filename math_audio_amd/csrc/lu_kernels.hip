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

// ------------------------------------------------------------------ panel factorisation
// Dynamic LDS: P[rpb][nb+1] | urow[2][nb] | drow[nb] | small scalars.
// Per column c every workgroup publishes its best pivot candidate (value, row, the row's nb panel
// entries) and, if it owns it, the current diagonal row; the granules are gathered in two levels (a leader per group of
// <= 32 workgroups reduces its group, every workgroup sweeps the 8 group granules), all reduce to the same pivot and fetch its
// row. Wavefront 0 carries this chain; wavefronts 1-3 do the bulk of the rank-1 update. The candidate of
// column c+1 is published BEFORE the bulk of step c's rank-1 update: only column c+1 is brought up to date first, the
// rows go out as they stand and the receivers finish step c's update on the one row they fetch. The chip-wide wait
// therefore overlaps the local update, and nothing but the scan of column c+1 sits between a pivot and the next publish.
struct PanelCand { double v; int row; };

__device__ __forceinline__ PanelCand wave_best(PanelCand c) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double ov = __shfl_xor(c.v, off, 64); int orow = __shfl_xor(c.row, off, 64);
    if (cand_better(ov, orow, c.v, c.row)) { c.v = ov; c.row = orow; }
  }
  return c;
}

__global__ __launch_bounds__(256, 1) void lu_panel_kernel(dc* __restrict__ A, int n, int k0, int nb, int rpb, LuPanelWs ws,
                                                          int* __restrict__ ipiv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pitch = nb + 1;
  dc* P = reinterpret_cast<dc*>(smem);
  dc* urow_a = P + (size_t)rpb * pitch;                  // pivot rows of the current and the previous column (ping-pong)
  dc* urow_b = urow_a + nb;
  dc* drow = urow_b + nb;
  double* s_wv = reinterpret_cast<double*>(drow + nb);   // [4] wave maxima
  int* s_wr = reinterpret_cast<int*>(s_wv + 4);          // [4] rows
  int* s_misc = s_wr + 4;                                // [0] best row, [3] fail
  double* s_bestv = reinterpret_cast<double*>(s_misc + 4);

  // The panel is a chain of short, latency-critical steps; when it shares a CU with throughput-bound
  // wavefronts (another frequency's trailing update) its instructions should issue first.
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, nblk = gridDim.x;
  const int r0 = k0 + b * rpb;
  const int nrows = min(rpb, n - r0);
  const int myrow = r0 + tid;                            // thread-per-row phases

  for (int idx = tid; idx < nrows * nb; idx += 256) {
    int rr = idx / nb, j = idx - rr * nb;
    P[rr * pitch + j] = A[(size_t)(r0 + rr) * n + k0 + j];
  }
  if (tid == 0) { s_misc[3] = 0; s_misc[2] = (int)__hip_atomic_load(ws.timeout, RLX_AGENT); }
  __syncthreads();
  // A plan whose earlier panel was aborted (exchange timed out, see below) is poisoned: every later panel kernel leaves at
  // once and records identity pivots, so that nothing downstream ever sees an unwritten pivot. Workgroups that miss the
  // flag here meet it in their first sweep.
  if (s_misc[2] != 0) {
    if (b == 0) for (int j = tid; j < nb; j += 256) ipiv[k0 + j] = k0 + j;
    return;
  }

  // local candidate of column `col` among rows >= k0+col -> s_bestv[0], s_misc[0] (after the barriers)
  auto scan_column = [&](int col) {
    PanelCand cd; cd.v = -1.0; cd.row = INT_MAX;
    if (nrows <= 64) {                                   // one wavefront sees every row: no cross-wave step
      if (wave == 0) {
        if (tid < nrows && myrow >= k0 + col) { cd.v = cabs1(P[tid * pitch + col]); cd.row = myrow; }
        cd = wave_best(cd);
        if (lane == 0) { s_bestv[0] = cd.v; s_misc[0] = cd.row; }
      }
      __syncthreads();
      return;
    }
    if (tid < nrows && myrow >= k0 + col) { cd.v = cabs1(P[tid * pitch + col]); cd.row = myrow; }
    cd = wave_best(cd);
    if (lane == 0) { s_wv[wave] = cd.v; s_wr[wave] = cd.row; }
    __syncthreads();
    if (tid == 0) {
      double v = s_wv[0]; int row = s_wr[0];
      for (int w = 1; w < 4; ++w) if (cand_better(s_wv[w], s_wr[w], v, row)) { v = s_wv[w]; row = s_wr[w]; }
      s_bestv[0] = v; s_misc[0] = row;
    }
    __syncthreads();
  };
  // publish the candidate (and the diagonal row, if owned) of column `col`. The 8-byte granule
  // {high 32 bits of |re|+|im|, tag = col+1, row} is both the data and the flag: it is stored last,
  // after every wave has drained the row payload (write-through stores), by ONE lane.
  // Pivot selection therefore compares magnitudes to 21 significant bits (exponent + 20 mantissa
  // bits); ties go to the lower row. The pivot is within 1e-6 relative of the column maximum.
  auto publish = [&](int col) {
    const int buf = col & 1;
    const double bv = s_bestv[0]; const int br = s_misc[0];
    const int gd = k0 + col;
    const bool own_diag = gd >= r0 && gd < r0 + nrows;
    // Wavefront 0 alone publishes (a row of <= 128 columns is 4 doubles per lane): it reads the rows from LDS, releases
    // the other wavefronts to the bulk update with one barrier, and only then waits for its write-through stores.
    double cv[4], dv[4];
    if (wave == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = lane + 64 * u;
        cv[u] = (br != INT_MAX && t < 2 * nb) ? reinterpret_cast<const double*>(P + (size_t)(br - r0) * pitch)[t] : 0.0;
        dv[u] = (own_diag && t < 2 * nb) ? reinterpret_cast<const double*>(P + (size_t)(gd - r0) * pitch)[t] : 0.0;
      }
    }
    __syncthreads();
    if (wave != 0) return;
    if (br != INT_MAX) {
      u64* dst = ws.candrow + ((size_t)buf * ws.max_blocks + b) * (2 * LU_NB_MAX);
#pragma unroll
      for (int u = 0; u < 4; ++u) if (lane + 64 * u < 2 * nb) st_sc1(dst + lane + 64 * u, cv[u]);
    }
    if (own_diag) {
      u64* dst = ws.diagrow + (size_t)buf * (2 * LU_NB_MAX);
#pragma unroll
      for (int u = 0; u < 4; ++u) if (lane + 64 * u < 2 * nb) st_sc1(dst + lane + 64 * u, dv[u]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the payload is out before the granule says so
    if (lane == 0) {
      const u64 hi = br != INT_MAX ? ((u64)__double_as_longlong(bv) >> 32) : 0ull;
      const u64 lo = ((u64)(unsigned)(col + 1) << 24) | (u64)(br != INT_MAX ? (unsigned)br : 0xFFFFFFu);
      __hip_atomic_store(ws.cand + ((size_t)buf * ws.max_blocks + b) * LU_GRANULE_STRIDE, (hi << 32) | lo, RLX_AGENT);
    }
  };

  // group leader (the group's last member), wavefront 0: gather the group's granules of column `col` and publish the group's
  // granule. Runs right after the leader's own publish, in place of its share of the bulk update: every workgroup waits for it.
  const int g_ngrp = nblk < LU_GROUPS ? nblk : LU_GROUPS;
  const int g_grp = b % g_ngrp;
  const int g_per = (nblk - g_grp + g_ngrp - 1) / g_ngrp;  // members g_grp, g_grp + g_ngrp, ...
  const bool lead = b == g_grp + (g_per - 1) * g_ngrp;
  auto leader_gather = [&](int col) {
    const int buf = col & 1;
    const unsigned want = (unsigned)(col + 1);
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    const u64* mbase = ws.cand + ((size_t)buf * ws.max_blocks + g_grp) * LU_GRANULE_STRIDE;
    unsigned bhi = 0, brow = 0xFFFFFFu; bool fail = false;
    for (;;) {
      bool ok = true; bhi = 0; brow = 0xFFFFFFu;
      unsigned ab = 0u;
      if (lane < g_per) {
        const u64 g = __hip_atomic_load(mbase + (size_t)lane * g_ngrp * LU_GRANULE_STRIDE, RLX_AGENT);
        ok = ((unsigned)(g >> 24) & 0xFFu) == want;
        bhi = (unsigned)(g >> 32); brow = (unsigned)g & 0xFFFFFFu;
      } else if (lane == 63) ab = __hip_atomic_load(ws.timeout, RLX_AGENT);          // the plan's abort flag rides along
      if (__all(ok)) break;
      if (__any(ab != 0u)) { fail = true; break; }
      __builtin_amdgcn_s_sleep(1);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) { fail = true; break; }   // 4 s at 100 MHz: never hang (the sweep below reports it)
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
      const unsigned ohi = __shfl_xor(bhi, off, 64), orow = __shfl_xor(brow, off, 64);
      if (ohi > bhi || (ohi == bhi && orow < brow)) { bhi = ohi; brow = orow; }
    }
    if (lane == 0 && !fail)
      __hip_atomic_store(ws.cand + ((size_t)2 * ws.max_blocks + (size_t)buf * LU_GROUPS + g_grp) * LU_GRANULE_STRIDE,
                         ((u64)bhi << 32) | ((u64)want << 24) | (u64)brow, RLX_AGENT);
  };

  scan_column(0);
  publish(0);
  if (lead && wave == 0) leader_gather(0);
#ifdef MA_PANEL_STAMPS
  u64 stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u64 stamp_t = __builtin_amdgcn_s_memrealtime();
#define MA_STAMP(i) do { if (tid == 0) { u64 now_ = __builtin_amdgcn_s_memrealtime(); stamp_acc[i] += now_ - stamp_t; stamp_t = now_; } } while (0)
#else
#define MA_STAMP(i) do { } while (0)
#endif

  bool pending = false;                                  // the previous column's rank-1 update is missing from the published rows
  for (int c = 0; c < nb; ++c) {
    const int gc = k0 + c;
    const int buf = c & 1;
    // ---- wavefront 0 sweeps every workgroup's granule until all carry this column's tag, and
    // reduces them on the way (the data is the flag: no counter, no second round trip)
    // ---- two-level gather of the candidates: workgroup b belongs to group b % 8 (its XCD under the round-robin placement);
    // the group's last member gathers the group's <= 32 granules and publishes the group's granule, every workgroup
    // sweeps only the 8 group granules (128 B apart). The leader's wavefront 0 starts its sweep right after publishing
    // (it takes no part in the bulk update, see below): the group result is on the critical path of every workgroup.
    if (wave == 0) {
      const unsigned want = (unsigned)(c + 1);
      const u64 t0 = __builtin_amdgcn_s_memrealtime();
      const int ngrp = nblk < LU_GROUPS ? nblk : LU_GROUPS;
      bool fail = false;
      unsigned bhi = 0, brow = 0xFFFFFFu;
      const u64* gbase = ws.cand + ((size_t)2 * ws.max_blocks + (size_t)buf * LU_GROUPS) * LU_GRANULE_STRIDE;
      if (gc == ws.test_abort_col && b == nblk - 1) fail = true;   // test hook: this workgroup behaves as if its wait had expired
      while (!fail) {
        bool ok = true; bhi = 0; brow = 0xFFFFFFu;
        unsigned ab = 0u;
        if (lane < ngrp) {
          const u64 g = __hip_atomic_load(gbase + (size_t)lane * LU_GRANULE_STRIDE, RLX_AGENT);
          ok = ((unsigned)(g >> 24) & 0xFFu) == want;
          bhi = (unsigned)(g >> 32); brow = (unsigned)g & 0xFFFFFFu;
        } else if (lane == 63) ab = __hip_atomic_load(ws.timeout, RLX_AGENT);        // another workgroup (or system of the plan) gave up
        if (__all(ok)) break;
        if (__any(ab != 0u)) { fail = true; break; }
        __builtin_amdgcn_s_sleep(LU_POLL_SLEEP);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) fail = true;
      }
#pragma unroll
      for (int off = 4; off > 0; off >>= 1) {
        const unsigned ohi = __shfl_xor(bhi, off, 64), orow = __shfl_xor(brow, off, 64);
        if (ohi > bhi || (ohi == bhi && orow < brow)) { bhi = ohi; brow = orow; }
      }
      if (lane == 0) {
        u64 best = brow; int bblk;
        if (fail) { __hip_atomic_store(ws.timeout, 1u, RLX_AGENT); s_misc[3] = 1; }
        if (fail || best >= (u64)n || best < (u64)gc) { best = (u64)gc; bblk = -1; if (!fail && b == 0) atomicCAS(ws.info, 0, gc + 1); }
        else bblk = ((int)best - k0) / rpb;              // rows are dealt to the workgroups in runs of rpb
        s_misc[1] = (int)best; s_misc[2] = bblk;
      }
    }
    __syncthreads();
#ifdef MA_PANEL_STAMPS
    if (c == 0 && tid == 0) stamp_acc[7] += __builtin_amdgcn_s_memrealtime() - stamp_t;   // residency wait: first column only
#endif
    MA_STAMP(0);
    if (s_misc[3]) {                                     // uniform: the whole workgroup leaves; the columns it did not reach get
      if (b == 0) for (int j = c + tid; j < nb; j += 256) ipiv[k0 + j] = k0 + j;   // identity pivots (the plan is poisoned: MA_ERR_HIP)
      return;
    }
    const int p = s_misc[1], wb = s_misc[2];
    MA_STAMP(1);
    // ---- fetch the pivot row (and the displaced diagonal row) with sc1 loads. Rows are published as they stood BEFORE
    // the bulk of the previous column's rank-1 update (publish() below), so the receiver finishes that update itself:
    // row[j] -= row[c-1] * u_{c-1}[j] for j > c, with the previous pivot row still in LDS. Every workgroup does the
    // same arithmetic on the same data, so all hold the same pivot row.
    dc* urow = (c & 1) ? urow_b : urow_a;
    const dc* uprev = (c & 1) ? urow_a : urow_b;
    {
      // wb < 0 (no candidate anywhere): the diagonal row stands in as the pivot row
      const u64* src = wb >= 0 ? ws.candrow + ((size_t)buf * ws.max_blocks + wb) * (2 * LU_NB_MAX) : ws.diagrow + (size_t)buf * (2 * LU_NB_MAX);
      const u64* s2 = ws.diagrow + (size_t)buf * (2 * LU_NB_MAX);
      for (int j = tid; j < nb; j += 256) {
        dc v = dc_make(ld_sc1(src + 2 * j), ld_sc1(src + 2 * j + 1));
        dc d = dc_make(0.0, 0.0);
        if (p != gc) d = dc_make(ld_sc1(s2 + 2 * j), ld_sc1(s2 + 2 * j + 1));
        if (pending && j > c) {
          const dc u = uprev[j];
          const dc lv = dc_make(ld_sc1(src + 2 * (c - 1)), ld_sc1(src + 2 * (c - 1) + 1));
          v.re -= lv.re * u.re - lv.im * u.im; v.im -= lv.re * u.im + lv.im * u.re;
          if (p != gc) {
            const dc ld = dc_make(ld_sc1(s2 + 2 * (c - 1)), ld_sc1(s2 + 2 * (c - 1) + 1));
            d.re -= ld.re * u.re - ld.im * u.im; d.im -= ld.re * u.im + ld.im * u.re;
          }
        }
        urow[j] = v;
        if (p != gc) drow[j] = d;
      }
    }
    __syncthreads();
    MA_STAMP(2);
    // ---- interchange inside the panel
    if (p != gc && p >= r0 && p < r0 + nrows) for (int t = tid; t < nb; t += 256) P[(size_t)(p - r0) * pitch + t] = drow[t];
    // the diagonal slot always takes the pivot row as every workgroup holds it (also when p == gc: the owner's own copy went
    // through the bulk update, the shared one through the receivers' completion above -- keep the one that was used)
    if (gc >= r0 && gc < r0 + nrows) for (int t = tid; t < nb; t += 256) P[(size_t)(gc - r0) * pitch + t] = urow[t];
    if (b == 0 && tid == 0) ipiv[gc] = p;
    const dc piv = urow[c];
    // lu.rs:106-110: a pivot column whose largest |z| is below 1e-30 is LuError::SingularMatrix (an exact zero is zgetf2's
    // INFO); the elimination of that column is skipped either way
    const bool singular = !(piv.re * piv.re + piv.im * piv.im >= 1e-60);
    if (singular && b == 0 && tid == 0) atomicCAS(ws.info, 0, gc + 1);   // first such pivot, 1-based
    __syncthreads();
    // ---- multipliers l = a / pivot (reciprocal scaling, zgetf2) and the update of column c+1 only
    const bool more = c + 1 < nb;
    dc anext = dc_make(0.0, 0.0);                        // this thread's row, column c+1, after the update
    if (tid < nrows && myrow > gc) {
      if (more) anext = P[tid * pitch + c + 1];
      if (!singular) {
        const dc l = P[tid * pitch + c] * crecip(piv);
        P[tid * pitch + c] = l;
        if (more) {
          const dc u = urow[c + 1];
          anext.re -= l.re * u.re - l.im * u.im; anext.im -= l.re * u.im + l.im * u.re;
          P[tid * pitch + c + 1] = anext;
        }
      }
    }
    MA_STAMP(3);
    if (more) {
      if (nrows <= 64) {
        // one wavefront holds every row: the next column's candidate straight from the registers. Magnitudes compare on
        // their top 32 bits (what the granule carries anyway), ties go to the lowest row = lowest lane.
        if (wave == 0) {
          const double mag = cabs1(anext);
          const bool valid = tid < nrows && myrow > gc && mag == mag;      // a NaN is never offered (as in scan_column)
          const unsigned hi = valid ? (unsigned)((u64)__double_as_longlong(mag) >> 32) : 0u;
          const unsigned m = wave_umax(hi);
          const u64 mask = __ballot(valid && hi == m);
          if (lane == 0) {
            s_misc[0] = mask ? r0 + (int)__builtin_ctzll(mask) : INT_MAX;
            s_bestv[0] = __longlong_as_double((long long)((u64)m << 32));
          }
        }
        __syncthreads();
      } else {
        scan_column(c + 1);
      }
      MA_STAMP(4);
      publish(c + 1);
      MA_STAMP(5);
      if (lead && wave == 0) leader_gather(c + 1);
    }
    // ---- bulk rank-1 update (overlaps the other workgroups' arrival): lane = row (the row pitch of
    // nb+1 complex spreads the lanes over all LDS banks); each wavefront takes every 4th group of 4
    // columns, loads the group before touching it so the LDS latency is paid once per group
    const int bwn = 3, bw = wave - 1;                    // wavefront 0 is busy publishing (and, in a leader, gathering)
    if (!singular && more && bw >= 0) {
      for (int rbase = 0; rbase < nrows; rbase += 64) {
        const int rr = rbase + lane;
        const int gr = r0 + rr;
        const bool on = rr < nrows && gr > gc;
        const dc l = on ? P[rr * pitch + c] : dc_make(0.0, 0.0);
        dc* Pr = P + (size_t)(on ? rr : 0) * pitch;
        for (int j0 = c + 2 + 4 * bw; j0 < nb; j0 += 4 * bwn) {
          dc u[4], a[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int j = min(j0 + q, nb - 1); u[q] = urow[j]; a[q] = Pr[j]; }
#pragma unroll
          for (int q = 0; q < 4; ++q) { a[q].re -= l.re * u[q].re - l.im * u[q].im; a[q].im -= l.re * u[q].im + l.im * u[q].re; }
          if (on) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (j0 + q < nb) Pr[j0 + q] = a[q];
          }
        }
      }
    }
    pending = !singular && more;
    // LDS-only barrier: the granule store of publish() may still be in flight (write-through ack ~1 us)
    // and nothing in the next column depends on it, so do not drain the vector-memory counter here.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    MA_STAMP(6);
  }
#ifdef MA_PANEL_STAMPS
  if (tid == 0 && b == 0) for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(ws.diagrow) + 2 * 2 * LU_NB_MAX + i, stamp_acc[i]);
#endif
  for (int idx = tid; idx < nrows * nb; idx += 256) {
    int rr = idx / nb, j = idx - rr * nb;
    A[(size_t)(r0 + rr) * n + k0 + j] = P[rr * pitch + j];
  }
}

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
#ifdef MA_PANEL_STAMPS
  // diagnostic build: [0] sender wave: B1 -> granule stored (ticks), [1] sender count, [2] wave 0: B1 -> sweep starts, [3] sweep time,
  // [4] sweeps, [5] wave 0: B2 -> B1 of the next column, [6] columns x workgroups, [7] sender: B1 -> row stores issued
  u64 stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u64 stamp_b1 = __builtin_amdgcn_s_memrealtime(), stamp_b2 = stamp_b1;
#define MA_NOW() __builtin_amdgcn_s_memrealtime()
#endif
#define MA_RSTAMP(i) do { } while (0)
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
#ifdef MA_PANEL_STAMPS
    stamp_b1 = MA_NOW();
#endif
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
#ifdef MA_PANEL_STAMPS
      if (lane == 0) stamp_acc[7] += MA_NOW() - stamp_b1;
#endif
    }
    MA_RSTAMP(0);
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
#ifdef MA_PANEL_STAMPS
      if (lane == 0) { stamp_acc[0] += MA_NOW() - stamp_b1; stamp_acc[1] += 1; }
#endif
    }
    // ---- wavefront 0: sweep every workgroup's granule until all carry this column's tag, reduce, fetch the winner's row
    if (wave == 0) {
      for (int q = 0; q < ws.diag_sleep; ++q) __builtin_amdgcn_s_sleep(8);   // diagnostic only (0 in production): 512 cycles each
      const u64 t0 = __builtin_amdgcn_s_memrealtime();
#ifdef MA_PANEL_STAMPS
      if (lane == 0) { stamp_acc[2] += t0 - stamp_b1; stamp_acc[6] += 1; }
#endif
      bool fail = gc == ws.test_abort_col && b == G - 1;   // test hook: this workgroup behaves as if its wait had expired
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
#ifdef MA_PANEL_STAMPS
        if (lane == 0) stamp_acc[4] += 1;
#endif
        if (wb >= 0 && wb != hb) fetch_row(wb);
        if (all) break;
        if (ab != 0u) { fail = true; break; }
        __builtin_amdgcn_s_sleep(LU_POLL_SLEEP);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) fail = true;   // 4 s at 100 MHz: never hang
      }
#ifdef MA_PANEL_STAMPS
      if (lane == 0) stamp_acc[3] += MA_NOW() - t0;
#endif
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
#ifdef MA_PANEL_STAMPS
    stamp_b2 = MA_NOW();
#endif
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
#ifdef MA_PANEL_STAMPS
        if (tid == 0) stamp_acc[5] += MA_NOW() - stamp_b2;
#endif
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
#ifdef MA_PANEL_STAMPS
  if (lane == 0) for (int i = 0; i < 8; ++i) if (stamp_acc[i]) atomicAdd(reinterpret_cast<unsigned long long*>(ws.diagrow) + 2 * 2 * LU_NB_MAX + i, stamp_acc[i]);
#endif
}

// ------------------------------------------------------------------ panel factorisation of several systems, a wavefront each
// The systems of a lock-step batch (frequencies of a sweep: same n, same panel) share ONE co-resident grid. A first form walked
// the systems one after the other inside the workgroup (all four wavefronts on one system's column step, then the next system):
// measured, it LOSES (0.63 ms per 32 columns of three systems = 6.6 us per system column against 5.8 alone; 99.9 ms per
// frequency against 63.1 with a panel kernel per system) -- with the exchange hidden, the bulk rank-1 update that the
// single-system kernel tucks under the exchange wait lands on the critical path of every step, and the steps' own memory
// round trips (pivot-row fetch, write-through publish) are paid system after system.
// Here every system of the batch has its OWN wavefront in the workgroup (<= 64 rows per workgroup: lane = row): the wavefront
// runs the whole column step for its system -- poll, fetch, interchange, multipliers, next candidate, publish, rank-1 update
// -- with no workgroup barrier anywhere, so the systems' chains advance independently and their latencies overlap on the
// CU's SIMDs; one co-resident grid instead of nsys grids that slow each other down, and the per-system arithmetic (and
// therefore every pivot, factor and solution) is the single-system kernel's at the same panel width.
struct LuPanelBatch { int nsys; dc* A[LU_BATCH_MAX]; int* ipiv[LU_BATCH_MAX]; LuPanelWs ws[LU_BATCH_MAX]; };

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256, 1) void lu_panel_wave_kernel(LuPanelBatch B, int n, int k0, int nb, int rpb, unsigned sys_lds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int pitch = nb + 1;
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63;
  const int sy = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // this wavefront's system
  const int b = blockIdx.x, nblk = gridDim.x;
  const int r0 = k0 + b * rpb;
  const int nrows = min(rpb, n - r0);                    // <= 64: lane = row
  const int myrow = r0 + lane;
  const int g_ngrp = nblk < LU_GROUPS ? nblk : LU_GROUPS;
  const int g_grp = b % g_ngrp;
  const int g_per = (nblk - g_grp + g_ngrp - 1) / g_ngrp;
  const bool lead = b == g_grp + (g_per - 1) * g_ngrp;
  const LuPanelWs ws = B.ws[sy];
  dc* __restrict__ A = B.A[sy];
  int* __restrict__ ipiv = B.ipiv[sy];
  unsigned* const poison = ws.timeout;

  dc* const P = reinterpret_cast<dc*>(smem + (size_t)sy * sys_lds);
  dc* const urow_a = P + (size_t)rpb * pitch;
  dc* const urow_b = urow_a + nb;
  dc* const drow = urow_b + nb;

  for (int idx = lane; idx < nrows * nb; idx += 64) {
    int rr = idx / nb, j = idx - rr * nb;
    P[rr * pitch + j] = A[(size_t)(r0 + rr) * n + k0 + j];
  }
  if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(poison, RLX_AGENT)) != 0) {
    if (b == 0) for (int j = lane; j < nb; j += 64) ipiv[k0 + j] = k0 + j;
    return;
  }
  wave_sync_lds();

  // candidate of a column from a per-lane magnitude: top 32 bits compared, ties to the lowest row (= lowest lane)
  auto pick = [=](bool valid, double mag, int* row_out, double* val_out) {
    const unsigned hi = valid ? (unsigned)((u64)__double_as_longlong(mag) >> 32) : 0u;
    const unsigned m = wave_umax(hi);
    const u64 mask = __ballot(valid && hi == m);
    *row_out = mask ? r0 + (int)__builtin_ctzll(mask) : INT_MAX;
    *val_out = __longlong_as_double((long long)((u64)m << 32));
  };
  auto publish = [&](int col, int br, double bv) {
    const int buf = col & 1;
    const int gd = k0 + col;
    const bool own_diag = gd >= r0 && gd < r0 + nrows;
    if (br != INT_MAX) {
      const double* src = reinterpret_cast<const double*>(P + (size_t)(br - r0) * pitch);
      u64* dst = ws.candrow + ((size_t)buf * ws.max_blocks + b) * (2 * LU_NB_MAX);
      for (int t = lane; t < 2 * nb; t += 64) st_sc1(dst + t, src[t]);
    }
    if (own_diag) {
      const double* src = reinterpret_cast<const double*>(P + (size_t)(gd - r0) * pitch);
      u64* dst = ws.diagrow + (size_t)buf * (2 * LU_NB_MAX);
      for (int t = lane; t < 2 * nb; t += 64) st_sc1(dst + t, src[t]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the payload is out before the granule says so
    if (lane == 0) {
      const u64 hi = br != INT_MAX ? ((u64)__double_as_longlong(bv) >> 32) : 0ull;
      const u64 lo = ((u64)(unsigned)(col + 1) << 24) | (u64)(br != INT_MAX ? (unsigned)br : 0xFFFFFFu);
      __hip_atomic_store(ws.cand + ((size_t)buf * ws.max_blocks + b) * LU_GRANULE_STRIDE, (hi << 32) | lo, RLX_AGENT);
    }
  };
  auto leader_gather = [&](int col) {
    const int buf = col & 1;
    const unsigned want = (unsigned)(col + 1);
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    const u64* mbase = ws.cand + ((size_t)buf * ws.max_blocks + g_grp) * LU_GRANULE_STRIDE;
    unsigned bhi = 0, brow = 0xFFFFFFu; bool fail = false;
    for (;;) {
      bool ok = true; bhi = 0; brow = 0xFFFFFFu;
      unsigned ab = 0u;
      if (lane < g_per) {
        const u64 g = __hip_atomic_load(mbase + (size_t)lane * g_ngrp * LU_GRANULE_STRIDE, RLX_AGENT);
        ok = ((unsigned)(g >> 24) & 0xFFu) == want;
        bhi = (unsigned)(g >> 32); brow = (unsigned)g & 0xFFFFFFu;
      } else if (lane == 63) ab = __hip_atomic_load(poison, RLX_AGENT);
      if (__all(ok)) break;
      if (__any(ab != 0u)) { fail = true; break; }
      __builtin_amdgcn_s_sleep(1);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) { fail = true; break; }
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
      const unsigned ohi = __shfl_xor(bhi, off, 64), orow = __shfl_xor(brow, off, 64);
      if (ohi > bhi || (ohi == bhi && orow < brow)) { bhi = ohi; brow = orow; }
    }
    if (lane == 0 && !fail)
      __hip_atomic_store(ws.cand + ((size_t)2 * ws.max_blocks + (size_t)buf * LU_GROUPS + g_grp) * LU_GRANULE_STRIDE,
                         ((u64)bhi << 32) | ((u64)want << 24) | (u64)brow, RLX_AGENT);
  };

  {
    int br; double bv;
    // column 0: the full-magnitude comparison of lu_panel_kernel's scan_column (cand_better), lowest row on ties
    PanelCand cd; cd.v = -1.0; cd.row = INT_MAX;
    if (lane < nrows && myrow >= k0) { cd.v = cabs1(P[lane * pitch]); cd.row = myrow; }
    cd = wave_best(cd);
    br = __builtin_amdgcn_readfirstlane(cd.row); bv = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(cd.v)), __builtin_amdgcn_readfirstlane(__double2loint(cd.v)));
    publish(0, br, bv);
    if (lead) leader_gather(0);
  }

#ifdef MA_PANEL_STAMPS
  u64 wst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u64 wst_t = __builtin_amdgcn_s_memrealtime();
#define MA_WSTAMP(i) do { if (b == 0 && sy == 0) { u64 now_ = __builtin_amdgcn_s_memrealtime(); wst_acc[i] += now_ - wst_t; wst_t = now_; } } while (0)
#else
#define MA_WSTAMP(i) do { } while (0)
#endif
  bool pending = false;
  for (int c = 0; c < nb; ++c) {
    const int gc = k0 + c;
    const int buf = c & 1;
    int p, wb; bool fail = false;
    {
      const unsigned want = (unsigned)(c + 1);
      const u64 t0 = __builtin_amdgcn_s_memrealtime();
      const int ngrp = nblk < LU_GROUPS ? nblk : LU_GROUPS;
      unsigned bhi = 0, brow = 0xFFFFFFu;
      const u64* gbase = ws.cand + ((size_t)2 * ws.max_blocks + (size_t)buf * LU_GROUPS) * LU_GRANULE_STRIDE;
      if (gc == ws.test_abort_col && b == nblk - 1 && sy == B.nsys - 1) fail = true;
      while (!fail) {
        bool ok = true; bhi = 0; brow = 0xFFFFFFu;
        unsigned ab = 0u;
        if (lane < ngrp) {
          const u64 g = __hip_atomic_load(gbase + (size_t)lane * LU_GRANULE_STRIDE, RLX_AGENT);
          ok = ((unsigned)(g >> 24) & 0xFFu) == want;
          bhi = (unsigned)(g >> 32); brow = (unsigned)g & 0xFFFFFFu;
        } else if (lane == 63) ab = __hip_atomic_load(poison, RLX_AGENT);
        if (__all(ok)) break;
        if (__any(ab != 0u)) { fail = true; break; }
        __builtin_amdgcn_s_sleep(LU_POLL_SLEEP);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 400000000ull) fail = true;
      }
#pragma unroll
      for (int off = 4; off > 0; off >>= 1) {
        const unsigned ohi = __shfl_xor(bhi, off, 64), orow = __shfl_xor(brow, off, 64);
        if (ohi > bhi || (ohi == bhi && orow < brow)) { bhi = ohi; brow = orow; }
      }
      const unsigned best = (unsigned)__builtin_amdgcn_readfirstlane((int)brow);
      if (fail) {                                         // this system's chain gives up: poison the plan, identity pivots for what is left
        if (lane == 0) __hip_atomic_store(poison, 1u, RLX_AGENT);
        if (b == 0) for (int j = c + lane; j < nb; j += 64) ipiv[k0 + j] = k0 + j;
        return;                                           // the other wavefronts (systems) meet the word in their next poll
      }
      if (best >= (unsigned)n || best < (unsigned)gc) { p = gc; wb = -1; if (b == 0 && lane == 0) atomicCAS(ws.info, 0, gc + 1); }
      else { p = (int)best; wb = (p - k0) / rpb; }
    }
    MA_WSTAMP(0);
    // ---- fetch the pivot row (and the displaced diagonal row); finish the previous column's pending update on them
    dc* urow = (c & 1) ? urow_b : urow_a;
    const dc* uprev = (c & 1) ? urow_a : urow_b;
    {
      const u64* src = wb >= 0 ? ws.candrow + ((size_t)buf * ws.max_blocks + wb) * (2 * LU_NB_MAX) : ws.diagrow + (size_t)buf * (2 * LU_NB_MAX);
      const u64* s2 = ws.diagrow + (size_t)buf * (2 * LU_NB_MAX);
      for (int j = lane; j < nb; j += 64) {
        dc v = dc_make(ld_sc1(src + 2 * j), ld_sc1(src + 2 * j + 1));
        dc d = dc_make(0.0, 0.0);
        if (p != gc) d = dc_make(ld_sc1(s2 + 2 * j), ld_sc1(s2 + 2 * j + 1));
        if (pending && j > c) {
          const dc u = uprev[j];
          const dc lv = dc_make(ld_sc1(src + 2 * (c - 1)), ld_sc1(src + 2 * (c - 1) + 1));
          v.re -= lv.re * u.re - lv.im * u.im; v.im -= lv.re * u.im + lv.im * u.re;
          if (p != gc) {
            const dc ld = dc_make(ld_sc1(s2 + 2 * (c - 1)), ld_sc1(s2 + 2 * (c - 1) + 1));
            d.re -= ld.re * u.re - ld.im * u.im; d.im -= ld.re * u.im + ld.im * u.re;
          }
        }
        urow[j] = v;
        if (p != gc) drow[j] = d;
      }
    }
    wave_sync_lds();
    MA_WSTAMP(1);
    // ---- interchange inside the panel
    if (p != gc && p >= r0 && p < r0 + nrows) for (int t = lane; t < nb; t += 64) P[(size_t)(p - r0) * pitch + t] = drow[t];
    if (gc >= r0 && gc < r0 + nrows) for (int t = lane; t < nb; t += 64) P[(size_t)(gc - r0) * pitch + t] = urow[t];
    if (b == 0 && lane == 0) ipiv[gc] = p;
    const dc piv = urow[c];
    const bool singular = !(piv.re * piv.re + piv.im * piv.im >= 1e-60);
    if (singular && b == 0 && lane == 0) atomicCAS(ws.info, 0, gc + 1);
    wave_sync_lds();
    // ---- multipliers and the update of column c+1
    const bool more = c + 1 < nb;
    const bool below = lane < nrows && myrow > gc;
    dc l = dc_make(0.0, 0.0), anext = dc_make(0.0, 0.0);
    if (below) {
      if (more) anext = P[lane * pitch + c + 1];
      if (!singular) {
        l = P[lane * pitch + c] * crecip(piv);
        P[lane * pitch + c] = l;
        if (more) {
          const dc u = urow[c + 1];
          anext.re -= l.re * u.re - l.im * u.im; anext.im -= l.re * u.im + l.im * u.re;
          P[lane * pitch + c + 1] = anext;
        }
      }
    }
    if (more) {
      const double mag = cabs1(anext);
      int br; double bv;
      pick(below && mag == mag, mag, &br, &bv);
      wave_sync_lds();
      MA_WSTAMP(2);
      publish(c + 1, br, bv);
      MA_WSTAMP(3);
      if (lead) leader_gather(c + 1);
      MA_WSTAMP(4);
      // ---- bulk rank-1 update of this system's rows (hidden behind the other systems' steps and this one's exchange)
      if (!singular && below) {
        dc* Pr = P + (size_t)lane * pitch;
        for (int j0 = c + 2; j0 < nb; j0 += 4) {
          dc u[4], a[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int j = min(j0 + q, nb - 1); u[q] = urow[j]; a[q] = Pr[j]; }
#pragma unroll
          for (int q = 0; q < 4; ++q) { a[q].re -= l.re * u[q].re - l.im * u[q].im; a[q].im -= l.re * u[q].im + l.im * u[q].re; }
#pragma unroll
          for (int q = 0; q < 4; ++q) if (j0 + q < nb) Pr[j0 + q] = a[q];
        }
      }
    }
    pending = !singular && more;
    wave_sync_lds();
    MA_WSTAMP(5);
  }
#ifdef MA_PANEL_STAMPS
  if (lane == 0 && b == 0 && sy == 0) for (int i = 0; i < 8; ++i) atomicAdd(reinterpret_cast<unsigned long long*>(ws.diagrow) + 2 * 2 * LU_NB_MAX + i, wst_acc[i]);
#endif
  for (int idx = lane; idx < nrows * nb; idx += 64) {
    int rr = idx / nb, j = idx - rr * nb;
    A[(size_t)(r0 + rr) * n + k0 + j] = P[rr * pitch + j];
  }
}

// (A third form kept the rows in REGISTERS -- lane = row with its 32 panel entries in 128 vector registers, columns picked with
// compare-and-select chains, published rows gathered with v_readlane, 200 VGPRs, no scratch once the 64 values were named
// scalars instead of arrays: bit-identical again, but 61.9 ms of panel time per frequency against 31.7-37.5 for the LDS form
// above: ~1500 vector instructions per column step, most of them selects and lane reads, cost more than the LDS traffic they
// replace. Removed; the measurements are in DESIGN.md 4.)

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

struct LuBlockPanels { int np; int k0[8]; int nb[8]; };   // a block's panels: first columns and widths

// ------------------------------------------------------------------ a whole block's interchanges in one launch (round 4)
// The main lane's per-panel gather + scatter (2 launches per panel, 12 per block of six) as ONE launch per block: the folded lists
// of the block's `np` panels (lists + j * lstride: [0] = m <= 2 nb, dst[], src[]) applied one after the other to the columns
// [x0, x1) U [y0, y1) and to the nrhs right-hand sides. Columns are independent, so a workgroup owns a strip of 32 columns and walks
// the panels on its own: per panel every moved entry of the strip is read into registers (16 per thread: 128 rows x 32 columns over
// 256 threads), then -- after a workgroup barrier, i.e. every read before any write -- written to its destination row; the next
// panel's reads see these writes (one CU, workgroup scope). The last workgroup takes the right-hand sides (row stride 1, one
// "column" per right-hand side). A poisoned plan moves nothing (the folded lists of an abandoned panel are empty anyway).
// Columns of [x0, x1) that lie inside the block receive only the interchanges of the panels to their right (P.k0[j] > column).
template <int NT>   // 8 NT = the most list entries a panel may have: 128 (panels of <= 64 columns: 16 registers of moved entries per thread) or 256
__global__ __launch_bounds__(256) void lu_block_row_moves_kernel(dc* __restrict__ A, int n, const int* __restrict__ lists, int lstride, LuBlockPanels P, int x0, int x1, int y0, int y1,
                                                                 dc* __restrict__ B, int nrhs, const unsigned* __restrict__ poison) {
  __shared__ int s_dst[2 * LU_NB_MAX], s_src[2 * LU_NB_MAX];
  __shared__ int s_m;
  if (poison && __hip_atomic_load(poison, RLX_AGENT) != 0u) return;
  const int tid = threadIdx.x;
  const int nx = x1 - x0, nxy = nx + (y1 - y0);
  const int nstrips = (nxy + 31) / 32;
  const bool rhs = (int)blockIdx.x >= nstrips;
  if (rhs && nrhs <= 0) return;
  // a strip never straddles the two column ranges' seam in a way that matters: column q of the set is x0 + q or y0 + (q - nx)
  const int q0 = 32 * (int)blockIdx.x;
  const int col_l = tid & 31;
  const int qc = q0 + col_l;
  const bool col_ok = rhs ? col_l < nrhs : qc < nxy;
  const size_t coff = rhs ? (size_t)col_l * (size_t)n : (size_t)(qc < nx ? x0 + qc : y0 + (qc - nx));
  const size_t rstride = rhs ? 1 : (size_t)n;
  dc* base = rhs ? B : A;
  const bool xpart = !rhs && q0 < nx;                       // (uniform: strips are 32 columns from x0, the panels' first columns multiples of 32 from x0 -- checked by the launcher)
  for (int j = 0; j < P.np; ++j) {
    // inside the block a panel's interchanges go to the columns LEFT of the panel only: its own columns were permuted by the panel
    // kernel, the block's columns right of it by the lane's step kernel
    if (xpart && x0 + q0 >= P.k0[j]) continue;
    const int* ls = lists + (size_t)j * lstride;
    if (tid == 0) { int m = ls[0]; if (m < 0 || m > 8 * NT) m = 0; s_m = m; }
    __syncthreads();
    const int m = s_m;
    if (m == 0) { __syncthreads(); continue; }
    for (int i = tid; i < m; i += 256) { s_dst[i] = ls[1 + i]; s_src[i] = ls[1 + 2 * LU_NB_MAX + i]; }
    __syncthreads();
    dc mv[NT]; int md[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int idx = (tid >> 5) + 8 * t;
      md[t] = -1; mv[t] = dc_make(0.0, 0.0);
      if (idx < m && col_ok) {
        const int sr = s_src[idx], ds = s_dst[idx];
        if (sr >= 0 && sr < n && ds >= 0 && ds < n) { mv[t] = base[(size_t)sr * rstride + coff]; md[t] = ds; }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every read of the strip has returned ...
    __syncthreads();                                       // ... in every thread, before the first write
#pragma unroll
    for (int t = 0; t < NT; ++t) if (md[t] >= 0) base[(size_t)md[t] * rstride + coff] = mv[t];
    __syncthreads();                                       // workgroup scope: the next panel's reads see these writes
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
__global__ __launch_bounds__(128) void lu_trsm_mfma_kernel(const dc* __restrict__ T, int ldt, int nb, const dc* __restrict__ invd,
                                                           dc* __restrict__ X, size_t xrs, size_t xcs, int nc, int nmain,
                                                           dc* __restrict__ X2, size_t x2rs, size_t x2cs, int nc2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* Lre = reinterpret_cast<double*>(smem);
  double* Lim = Lre + (size_t)max(32, ((nb + 15) >> 4) * 16) * TM_PITCH;   // the launch sizes the two planes for this nb
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

  v4d bre[8], bim[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    bre[t] = (v4d){0, 0, 0, 0}; bim[t] = (v4d){0, 0, 0, 0};
    if (t < NT && active) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * t + lk + 4 * r;
        if (row < nb && col < ncols) { const dc v = Xp[(size_t)row * rs + (size_t)col * cs]; bre[t][r] = v.re; bim[t][r] = v.im; }
      }
    }
  }

#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    if (blk * 32 >= nb) break;
    // slab rows: the (identity-padded) inverted diagonal block and everything below it, zero beyond nb up to the
    // tile boundary: padded rows of a tile feed later products as k-slices and must stay exactly zero
    const int rows = max(32, NT * 16 - blk * 32);
    __syncthreads();
    for (int idx = tid; idx < rows * 32; idx += 128) {
      const int rr = idx >> 5, c = idx & 31;
      dc v;
      if (rr < 32) v = invd[((size_t)blk * 32 + rr) * 32 + c];
      else v = (blk * 32 + rr < nb && blk * 32 + c < nb) ? T[(size_t)(blk * 32 + rr) * ldt + blk * 32 + c] : dc_make(0.0, 0.0);
      Lre[rr * TM_PITCH + c] = v.re; Lim[rr * TM_PITCH + c] = v.im;
    }
    __syncthreads();
    if (!active) continue;
    const int t0 = 2 * blk, t1 = 2 * blk + 1;
    v4d x0r = (v4d){0, 0, 0, 0}, x0i = x0r, x1r = x0r, x1i = x0r;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                     // X1 = D[16:32, 0:32] (B_t0; B_t1)
      const double ar = Lre[(16 + li) * TM_PITCH + ks * 4 + lk], ai = Lim[(16 + li) * TM_PITCH + ks * 4 + lk];
      const double br = bre[t0 + (ks >> 2)][ks & 3], bi = bim[t0 + (ks >> 2)][ks & 3];
      x1r = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, x1r, 0, 0, 0);
      x1i = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, x1i, 0, 0, 0);
      x1r = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, x1r, 0, 0, 0);
      x1i = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, x1i, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                     // X0 = D[0:16, 0:16] B_t0
      const double ar = Lre[li * TM_PITCH + ks * 4 + lk], ai = Lim[li * TM_PITCH + ks * 4 + lk];
      const double br = bre[t0][ks], bi = bim[t0][ks];
      x0r = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, br, x0r, 0, 0, 0);
      x0i = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, bi, x0i, 0, 0, 0);
      x0r = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, bi, x0r, 0, 0, 0);
      x0i = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, br, x0i, 0, 0, 0);
    }
    bre[t0] = x0r; bim[t0] = x0i;
    if (t1 < 8) { bre[t1 < 8 ? t1 : 7] = x1r; bim[t1 < 8 ? t1 : 7] = x1i; }
#pragma unroll
    for (int tj = t1 + 1; tj < 8; ++tj) {                // B_tj -= L[tj rows, slab] (X0; X1)
      if (tj >= NT) break;
      const int lr = tj * 16 - blk * 32 + li;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const double ar = Lre[lr * TM_PITCH + ks * 4 + lk], ai = Lim[lr * TM_PITCH + ks * 4 + lk];
        const double xr = (ks < 4) ? x0r[ks & 3] : x1r[ks & 3], xi = (ks < 4) ? x0i[ks & 3] : x1i[ks & 3];
        bre[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ar, xr, bre[tj], 0, 0, 0);
        bim[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ar, xi, bim[tj], 0, 0, 0);
        bre[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, xi, bre[tj], 0, 0, 0);
        bim[tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, xr, bim[tj], 0, 0, 0);
      }
    }
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t >= NT) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * t + lk + 4 * r;
        if (row < nb && col < ncols) Xp[(size_t)row * rs + (size_t)col * cs] = dc_make(bre[t][r], bim[t][r]);
      }
    }
  }
}

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

// ------------------------------------------------------------------ a whole block row of U in one launch (round 4)
// The main lane's work on block g right of the block -- per panel j: U_j = L_jj^-1 A[p_j rows, e:n) (lu_trsm64_kernel), the
// right-hand side's rows, and A[a_{j+1}:e, e:n) -= L[a_{j+1}:e, p_j] U_j (a K = 64 update launch), 3 launches per panel, 18 per
// block of six -- as ONE launch: columns are independent, so a wavefront owns 16 columns of [e, n) and takes them through the block's
// np <= 8 panels LEFT-looking: for panel i it fetches its 64 x 16 piece B_i of A12 (four accumulator tiles), subtracts
// L[p_i rows, p_j columns] X_j for the panels j < i it has already solved (X_j comes back from memory in the accumulator layout,
// which IS the MFMA B-operand layout; L in 64 x 32 pieces through LDS as re / im planes), solves with the inverted 32 x 32
// diagonal blocks exactly as lu_trsm64_kernel does, and stores. Four wavefronts (64 columns) share the LDS pieces. The last
// workgroup does the same for the right-hand sides (row stride 1): the forward substitution rides along, the rows below the
// block get theirs from one zgemv afterwards.
#define TB_ROWS 64
__global__ __launch_bounds__(256) void lu_block_trsm_kernel(const dc* __restrict__ A, int lda, LuBlockPanels P, const dc* __restrict__ invd, int invd_stride,
                                                            dc* __restrict__ X, size_t xrs, size_t xcs, int nc, int nmain,
                                                            dc* __restrict__ X2, size_t x2rs, size_t x2cs, int nc2) {
  __shared__ __attribute__((aligned(16))) double Lre[TB_ROWS * TM_PITCH];
  __shared__ __attribute__((aligned(16))) double Lim[TB_ROWS * TM_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const bool extra = (int)blockIdx.x >= nmain;
  dc* Xp = extra ? X2 : X;                                   // row 0 = the block's first row (P.k0[0])
  const size_t rs = extra ? x2rs : xrs, cs = extra ? x2cs : xcs;
  const int ncols = extra ? nc2 : nc;
  const int c0 = extra ? wave * 16 : ((int)blockIdx.x * 4 + wave) * 16;
  const bool active = c0 < ncols;
  const int col = c0 + li;
  const bool colok = active && col < ncols;
  const int a0 = P.k0[0];
  const v4d zero = (v4d){0, 0, 0, 0};
  // tile t (16 rows) of panel q's rows, this wavefront's 16 columns: register r of lane (li, lk) = row 16 t + lk + 4 r, column li
  auto load_tile = [&](int q, int t, v4d& br, v4d& bi) {
    br = zero; bi = zero;
    if (colok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * t + lk + 4 * r;
        if (row < P.nb[q]) { const dc v = Xp[(size_t)(P.k0[q] - a0 + row) * rs + (size_t)col * cs]; br[r] = v.re; bi[r] = v.im; }
      }
    }
  };
  auto store_tile = [&](int q, int t, const v4d& br, const v4d& bi) {
    if (colok) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * t + lk + 4 * r;
        if (row < P.nb[q]) Xp[(size_t)(P.k0[q] - a0 + row) * rs + (size_t)col * cs] = dc_make(br[r], bi[r]);
      }
    }
  };
  // LDS <- `rows` x 32 entries: what 0: the inverted diagonal block `blk` of panel q; 1: L[p_q rows r0.., p_j columns c0j..c0j+31] (zero beyond the panels)
  auto load_rows = [&](int what, int q, int blk, int j, int r0, int cj, int rows) {
    __syncthreads();
    for (int idx = tid; idx < rows * 32; idx += 256) {
      const int rr = idx >> 5, c = idx & 31;
      dc v;
      if (what == 0) v = invd[(size_t)q * invd_stride + ((size_t)blk * 32 + rr) * 32 + c];
      else v = (r0 + rr < P.nb[q] && cj + c < P.nb[j]) ? A[(size_t)(P.k0[q] + r0 + rr) * lda + P.k0[j] + cj + c] : dc_make(0.0, 0.0);
      Lre[rr * TM_PITCH + c] = v.re; Lim[rr * TM_PITCH + c] = v.im;
    }
    __syncthreads();
  };
  // T -= L[LDS rows lr0 + li, 32 columns] (Xa; Xb)
  auto update_tile = [&](int lr0, v4d& tr, v4d& ti, const v4d& xar, const v4d& xai, const v4d& xbr, const v4d& xbi) {
    const int lr = lr0 + li;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const double ar = Lre[lr * TM_PITCH + ks * 4 + lk], ai = Lim[lr * TM_PITCH + ks * 4 + lk];
      const double xr = (ks < 4) ? xar[ks & 3] : xbr[ks & 3], xi = (ks < 4) ? xai[ks & 3] : xbi[ks & 3];
      tr = __builtin_amdgcn_mfma_f64_16x16x4f64(-ar, xr, tr, 0, 0, 0);
      ti = __builtin_amdgcn_mfma_f64_16x16x4f64(-ar, xi, ti, 0, 0, 0);
      tr = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, xi, tr, 0, 0, 0);
      ti = __builtin_amdgcn_mfma_f64_16x16x4f64(-ai, xr, ti, 0, 0, 0);
    }
  };
  // (Ba; Bb) <- D (Ba; Bb) with the unit-lower-triangular inverse D in LDS rows 0..31 (lu_trsm64_kernel's solve_pair)
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
  for (int i = 0; i < P.np; ++i) {
    v4d b0r, b0i, b1r, b1i, b2r, b2i, b3r, b3i;
    load_tile(i, 0, b0r, b0i); load_tile(i, 1, b1r, b1i); load_tile(i, 2, b2r, b2i); load_tile(i, 3, b3r, b3i);
    const int nti = (P.nb[i] + 15) >> 4;
    for (int j = 0; j < i; ++j) {
      v4d x0r, x0i, x1r, x1i;
#pragma unroll 1
      for (int cb = 0; cb < 2; ++cb) {
        if (32 * cb >= P.nb[j]) break;                     // (uniform)
        // the solved rows of panel j came back from this wavefront's own stores: same lanes, same addresses
        load_tile(j, 2 * cb, x0r, x0i); load_tile(j, 2 * cb + 1, x1r, x1i);
        load_rows(1, i, 0, j, 0, 32 * cb, TB_ROWS);
        if (active) {
          update_tile(0, b0r, b0i, x0r, x0i, x1r, x1i);
          if (nti > 1) update_tile(16, b1r, b1i, x0r, x0i, x1r, x1i);
          if (nti > 2) update_tile(32, b2r, b2i, x0r, x0i, x1r, x1i);
          if (nti > 3) update_tile(48, b3r, b3i, x0r, x0i, x1r, x1i);
        }
      }
    }
    load_rows(0, i, 0, 0, 0, 0, 32);
    if (active) solve_pair(b0r, b0i, b1r, b1i);
    if (P.nb[i] > 32) {
      load_rows(1, i, 0, i, 32, 0, 32);                    // L10 of panel i: rows 32.., columns 0..31
      if (active) {
        update_tile(0, b2r, b2i, b0r, b0i, b1r, b1i);
        if (nti > 3) update_tile(16, b3r, b3i, b0r, b0i, b1r, b1i);
      }
      load_rows(0, i, 1, 0, 0, 0, 32);
      if (active) solve_pair(b2r, b2i, b3r, b3i);
    }
    store_tile(i, 0, b0r, b0i); store_tile(i, 1, b1r, b1i); store_tile(i, 2, b2r, b2i); store_tile(i, 3, b3r, b3i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next panels read these rows back
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

// ------------------------------------------------------------------ C -= A * B on the f64 matrix cores
// A: M x K (lda), B: K x N (ldb), C: M x N (ldc), row-major complex128. Workgroup tile 128 x 128,
// K in steps of 8 (two v_mfma_f64_16x16x4 k-steps), 512 threads = 8 wavefronts as 4 (M) x 2 (N);
// each wavefront owns 32 x 64 = 2 x 4 MFMA tiles with separate real/imaginary accumulators.
// LDS: As[2][8][128] (k-major: the A fragment of a tile is 16 consecutive complex), Bs[2][8][128].
// Operand fragment of v_mfma_f64_16x16x4_f64: lane l holds A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; result register r of lane l is D[(l >> 4) + 4 r][l & 15].
#define ZG_BM 128
#define ZG_BN 128
#define ZG_BK 8

__global__ __launch_bounds__(512, 1) void zgemm_sub_kernel(int M, int N, int K, const dc* __restrict__ A, size_t lda,
                                                           const dc* __restrict__ B, size_t ldb, dc* __restrict__ C, size_t ldc) {
  __shared__ __attribute__((aligned(16))) dc As[2][ZG_BK][ZG_BM];
  __shared__ __attribute__((aligned(16))) dc Bs[2][ZG_BK][ZG_BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;           // 4 x 2 wave grid
  const int m0 = blockIdx.y * ZG_BM, n0 = blockIdx.x * ZG_BN;
  const int li = lane & 15, lk = lane >> 4;

  v4d accr[2][4], acci[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { accr[a][b] = (v4d){0, 0, 0, 0}; acci[a][b] = (v4d){0, 0, 0, 0}; }

  // staging assignment: 1024 A elements and 1024 B elements per stage, 2 + 2 per thread
  dc ra[2], rb[2];
  auto load_stage = [&](int k0) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int e = tid + 512 * s;
      const int row = e >> 3, kk = e & 7;                       // A: 8 consecutive k of one row = 128 B
      const int gm = m0 + row, gk = k0 + kk;
      ra[s] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : dc_make(0.0, 0.0);
      const int bk = e >> 7, bn = e & 127;                      // B: 128 consecutive n of one k-row
      const int gn = n0 + bn, gk2 = k0 + bk;
      rb[s] = (gn < N && gk2 < K) ? B[(size_t)gk2 * ldb + gn] : dc_make(0.0, 0.0);
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int e = tid + 512 * s;
      As[buf][e & 7][(e >> 3) ^ (e & 7)] = ra[s];   // XOR swizzle: the 8 lanes of one row (k = 0..7, 1 KB apart) hit 8 different bank groups
      Bs[buf][e >> 7][e & 127] = rb[s];
    }
  };

  const int nstage = (K + ZG_BK - 1) / ZG_BK;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstage) load_stage((st + 1) * ZG_BK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = ks * 4 + lk;
      dc af[2], bf[4];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = As[buf][kk][(wm * 32 + a * 16 + li) ^ kk];
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = Bs[buf][kk][wn * 64 + b * 16 + li];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double nai = -af[a].im;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          accr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].re, bf[b].re, accr[a][b], 0, 0, 0);
          accr[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, bf[b].im, accr[a][b], 0, 0, 0);
          acci[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].re, bf[b].im, acci[a][b], 0, 0, 0);
          acci[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a].im, bf[b].re, acci[a][b], 0, 0, 0);
        }
      }
    }
    if (st + 1 < nstage) store_stage(buf ^ 1);
    __syncthreads();
  }
  // epilogue: C -= acc
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm * 32 + a * 16 + lk + 4 * r;
        const int gn = n0 + wn * 64 + b * 16 + li;
        if (gm < M && gn < N) {
          dc* pc = C + (size_t)gm * ldc + gn;
          dc c = *pc;
          c.re -= accr[a][b][r]; c.im -= acci[a][b][r];
          *pc = c;
        }
      }
}

// ------------------------------------------------------------------ C -= A * B, 3M form
// Same contract as zgemm_sub_kernel with three real products per complex product:
//   T1 = Ar Br, T2 = Ai Bi, T3 = (Ar + Ai)(Br + Bi);  Re = T1 - T2,  Im = T3 - T1 - T2.
// 25 % fewer matrix-core instructions than the 4-product form; normwise (not componentwise)
// backward stable, which is what the LU update needs. Workgroup tile 64 x 64, 4 wavefronts as
// 2 (M) x 2 (N), 32 x 32 per wavefront = 2 x 2 MFMA tiles x 3 accumulators. Two or three such workgroups
// share a CU (165 VGPRs, 32 KB LDS each), so that one's C read-modify-write epilogue runs under the
// others' MFMA loops (with one 128 x 64 workgroup per CU the epilogue cost 22 % of the kernel); the sums Ar+Ai and Br+Bi are formed once per fragment load.
#define Z3_BM 64
#define Z3_BN 64
#ifndef MA_ZGEMM_DMA_DEFAULT
#define MA_ZGEMM_DMA_DEFAULT 1
#endif
#define Z3_STAGES 3                          // LDS stages of 16 KB: three workgroups of 48 KB share a CU

#ifndef MA_ZGEMM_MAXWAVES
#define MA_ZGEMM_MAXWAVES 2
#endif
// Tile order. ctr == nullptr: workgroup (bx, by) takes tile (bx, by). ctr != nullptr (large updates): a grid of <= 2 workgroups
// per CU DRAWS its tiles: the tiles are grouped in blocks of 8 x 8, block s belongs to XCD s mod 8, and a workgroup asks the
// counter of the XCD it runs on (XCC_ID, read at run time -- the dispatcher's placement is not a function of blockIdx once other
// kernels are in flight) for the next tile of that XCD's blocks; 64 consecutive draws are one block, so the <= 64 workgroups
// an XCD runs at a time share 8 row panels of A and 8 column panels of B in that XCD's L2 instead of fetching each over the
// fabric (every tile needs 2 x 64 x K entries of A and B for 64 x 64 of C). An XCD whose blocks are exhausted draws from the
// next XCD's. ctr[0..7] draws, ctr[8] workgroups that have left; the last one zeroes the counters for the launch that
// reuses them.
__device__ __forceinline__ unsigned zg_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}
// DRAW = false is the plain kernel (one tile per workgroup, tile = blockIdx): the tile loop and everything of the draw fold away
template <bool DRAW>
__device__ __forceinline__ void zgemm3m_body(int M, int N, int K, const dc* __restrict__ A, size_t lda, const dc* __restrict__ B, size_t ldb, dc* __restrict__ C, size_t ldc,
                                             unsigned* __restrict__ ctr, int one_tile, dc (*As)[ZG_BK][Z3_BM], dc (*Bs)[ZG_BK][Z3_BN], int* s_tile) {
  const int tid0 = threadIdx.x;
  const int gx = (N + Z3_BN - 1) / Z3_BN, gy = (M + Z3_BM - 1) / Z3_BM;
  const int sgx = (gx + 7) >> 3, NS = sgx * ((gy + 7) >> 3);
  const unsigned myx = DRAW ? zg_xcc_id() : 0u;
  for (bool first = true;; first = false) {
  // the per-thread index arithmetic is redone for every tile (the opaque copy keeps it from being hoisted out of the tile loop
  // and carried in registers across it: the kernel has to stay within 176 of them to share a SIMD with two panel wavefronts)
  int tid = tid0;
  if (DRAW) asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  int m0, n0;
  if (!DRAW) {
    if (!first) break;
    m0 = blockIdx.y * Z3_BM; n0 = blockIdx.x * Z3_BN;
  } else {
    if (tid == 0) {
      int ty = -1, tx = -1;
      for (int v = 0; v < 8 && ty < 0; ++v) {
        const int x = (int)((myx + (unsigned)v) & 7u);
        const unsigned space = (unsigned)((NS - x + 7) / 8) * 64u;          // draws of XCD x: its blocks s = x, x + 8, ... times 64 tiles
        for (;;) {
          const unsigned r = atomicAdd(ctr + x, 1u);
          if (r >= space) break;
          const int sblk = x + 8 * (int)(r >> 6), t = (int)(r & 63u);
          const int sy = sblk / sgx, sx = sblk - sy * sgx;
          const int y = sy * 8 + (t >> 3), xx = sx * 8 + (t & 7);
          if (y < gy && xx < gx) { ty = y; tx = xx; break; }
        }
      }
      s_tile[0] = ty; s_tile[1] = tx;
    }
    __syncthreads();
    const int ty = __builtin_amdgcn_readfirstlane(s_tile[0]), tx = __builtin_amdgcn_readfirstlane(s_tile[1]);   // uniform: keep the tile origin on the scalar side
    __syncthreads();                                                        // s_tile is rewritten by the next draw
    if (ty < 0) {
      if (tid == 0 && atomicAdd(ctr + 8, 1u) == gridDim.x - 1u) { for (int i = 0; i < 9; ++i) ctr[i] = 0u; }
      break;
    }
    m0 = ty * Z3_BM; n0 = tx * Z3_BN;
  }

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
  if (DRAW) {
    __syncthreads();                                                        // the next tile's first stage overwrites the LDS buffers
    if (one_tile) {                                                         // one tile per workgroup (grid = tiles): leave, counting out
      if (tid == 0 && atomicAdd(ctr + 8, 1u) == gridDim.x - 1u) { for (int i = 0; i < 9; ++i) ctr[i] = 0u; }
      break;
    }
  }
  }
}
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(1, MA_ZGEMM_MAXWAVES))) void zgemm3m_sub_kernel(int M, int N, int K, const dc* __restrict__ A, size_t lda,
                                                             const dc* __restrict__ B, size_t ldb, dc* __restrict__ C, size_t ldc) {
  __shared__ __attribute__((aligned(16))) dc As[Z3_STAGES][ZG_BK][Z3_BM];
  __shared__ __attribute__((aligned(16))) dc Bs[Z3_STAGES][ZG_BK][Z3_BN];
  zgemm3m_body<false>(M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, As, Bs, nullptr);
}
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(1, MA_ZGEMM_MAXWAVES))) void zgemm3m_sub_drawn_kernel(int M, int N, int K, const dc* __restrict__ A, size_t lda,
                                                             const dc* __restrict__ B, size_t ldb, dc* __restrict__ C, size_t ldc, unsigned* __restrict__ ctr, int one_tile) {
  __shared__ __attribute__((aligned(16))) dc As[Z3_STAGES][ZG_BK][Z3_BM];
  __shared__ __attribute__((aligned(16))) dc Bs[Z3_STAGES][ZG_BK][Z3_BN];
  __shared__ int s_tile[2];
  zgemm3m_body<true>(M, N, K, A, lda, B, ldb, C, ldc, ctr, one_tile, As, Bs, s_tile);
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
size_t lu_panel_lds_bytes(int nb, int rpb) {
  return (size_t)rpb * (nb + 1) * sizeof(dc) + 3 * (size_t)nb * sizeof(dc) + 4 * sizeof(double) + 12 * sizeof(int) + sizeof(double) + 64;
}

int lu_panel_configure() {
  MA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lu_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  MA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lu_panel_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return MA_OK;
}

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
  int regs[3] = {0, 0, 0};   // vector registers per lane of lu_panel_kernel, lu_panel_wave_kernel, lu_panel_reg_kernel
  int occ_checked_lds = 0;
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

int lu_panel_regs(int kind) {
  std::lock_guard<std::mutex> lock(g_seq.mu);
  if (g_seq.regs[0] == 0) {
    const void* f[3] = {reinterpret_cast<const void*>(lu_panel_kernel), reinterpret_cast<const void*>(lu_panel_wave_kernel), reinterpret_cast<const void*>(lu_panel_reg_kernel<LU_REG_NB>)};
    for (int q = 0; q < 3; ++q) {
      hipFuncAttributes fa;
      MA_HIP(hipFuncGetAttributes(&fa, f[q]));
      g_seq.regs[q] = fa.numRegs > 0 ? fa.numRegs : 128;
    }
  }
  return g_seq.regs[kind >= 0 && kind <= 2 ? kind : 0];
}

// MA_OK when a grid of nblk workgroups with this panel shape can be co-resident on ncu CUs on its own
int lu_panel_admissible(int nb, int rpb, int nblk, int ncu) {
  const size_t lds = lu_panel_lds_bytes(nb, rpb);
  const int regs = lu_panel_regs(0);
  const int p = lu_panel_slots_per_cu(lds, regs);
  MA_REQUIRE(p >= 1 && (long long)nblk <= (long long)p * ncu, MA_ERR_UNSUPPORTED,
             "panel grid of %d workgroups x %zu B LDS (%d columns, %d rows each) cannot be co-resident on %d CUs (%d per CU)", nblk, lds, nb, rpb, ncu, p);
  return MA_OK;
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

// admission + launch of a panel kernel. kind 0: lu_panel_kernel on (A[0], ws[0], ipiv[0]); kind 1: lu_panel_wave_kernel over nsys
// systems; kind 2: lu_panel_reg_kernel (rows in registers, 256 rows per workgroup, <= LU_REG_NB columns). `ncu` is the number of
// CUs the stream may use (a CU-masked stream: the CUs of its mask).
static int launch_panel_any(int kind, int nsys, c64* const* As, int n, int k0, int nb, int rpb, int nblk, int ncu, const LuPanelWs* wss, int* const* ipivs, bool clear_tags, hipStream_t st,
                            int* reg_lists = nullptr, c64* reg_lrows = nullptr, int reg_lcol0 = 0, const int* reg_run_if_nonzero = nullptr) {
  int dev = 0;
  MA_HIP(hipGetDevice(&dev));
  MA_REQUIRE(dev >= 0 && dev < 16, MA_ERR_UNSUPPORTED, "device index %d beyond the panel sequencer table", dev);
  MA_REQUIRE(nsys >= 1 && nsys <= LU_GROUP_MAX, MA_ERR_INVALID, "%d systems per panel kernel", nsys);
  MA_REQUIRE(kind == 2 ? (nsys == 1 && rpb == 256 && nb >= 1 && nb <= LU_REG_NB) : (kind == (nsys == 1 ? 0 : 1)), MA_ERR_INVALID, "panel kernel kind %d with %d systems, %d rows per workgroup, %d columns", kind, nsys, rpb, nb);
  const LuPanelWs& ws = wss[0];
  MA_REQUIRE(nblk >= 1 && nblk <= ws.max_blocks, MA_ERR_INVALID, "panel grid of %d workgroups outside the workspace (%d)", nblk, ws.max_blocks);
  MA_REQUIRE((long long)k0 + (long long)(nblk - 1) * rpb < n && (long long)k0 + (long long)nblk * rpb >= n, MA_ERR_INVALID,
             "panel grid (%d workgroups of %d rows from row %d) does not tile the %d rows", nblk, rpb, k0, n);
  MA_REQUIRE(n < 0xFFFFFF, MA_ERR_UNSUPPORTED, "row positions travel in 24 bits of the exchange granule");
  const size_t sys_lds = (lu_panel_lds_bytes(nb, rpb) + 15) & ~(size_t)15;
  const size_t lds = kind == 2 ? lu_panel_reg_lds() : (kind == 0 ? lu_panel_lds_bytes(nb, rpb) : sys_lds * (size_t)nsys);
  const int regs = lu_panel_regs(kind);
  {
    const int p = lu_panel_slots_per_cu(lds, regs);
    MA_REQUIRE(p >= 1 && (long long)nblk <= (long long)p * ncu, MA_ERR_UNSUPPORTED,
               "panel grid of %d workgroups x %zu B LDS (%d systems, %d columns, %d rows each) cannot be co-resident on %d CUs (%d per CU)", nblk, lds, nsys, nb, rpb, ncu, p);
  }
  {
    std::lock_guard<std::mutex> lock(g_seq.mu);
    if (kind == 2 ? !g_seq.occ_checked_reg : (int)lds > g_seq.occ_checked_lds) {
      // the runtime's own occupancy figure must not be below the slots the rule assumes (registers, waves, LDS granularity)
      int occ = 0;
      if (kind == 0) MA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(lu_panel_kernel), 256, lds));
      else if (kind == 1) MA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(lu_panel_wave_kernel), 64 * nsys, lds));
      else MA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void*>(lu_panel_reg_kernel<LU_REG_NB>), 256, 0));
      MA_REQUIRE(occ >= lu_panel_slots_per_cu(lds, regs), MA_ERR_UNSUPPORTED, "panel kernel occupancy %d per CU at %zu B LDS is below the %d slots the admission rule assumes",
                 occ, lds, lu_panel_slots_per_cu(lds, regs));
      if (kind == 2) g_seq.occ_checked_reg = true; else g_seq.occ_checked_lds = (int)lds;
    }
  }
  SpinLaunch guard;
  { const int arc = guard.admit(st, nblk, lds, regs, ncu); if (arc) return arc; }
  // Stale tags must not match. A workgroup rewrites its granule every column, so only columns 0 and 1 of a launch can
  // see the previous launch's granules, which carry that launch's last two tags (nb and nb - 1): they differ from the
  // wanted 1 and 2 whenever the previous panel of this workspace had >= 4 columns. Otherwise (and at the start of a
  // factorisation, whose predecessor may have been aborted) the granules are cleared.
  if (clear_tags) for (int t = 0; t < nsys; ++t) MA_HIP(hipMemsetAsync(wss[t].cand, 0, lu_panel_granule_bytes(wss[t].max_blocks), st));
  if (kind == 2) hipLaunchKernelGGL(lu_panel_reg_kernel<LU_REG_NB>, dim3(nblk), dim3(256), 0, st, reinterpret_cast<dc*>(As[0]), n, k0, nb, ws, ipivs[0], reg_lists,
                                    reinterpret_cast<dc*>(reg_lrows), reg_lcol0, reg_run_if_nonzero);
  else if (nsys == 1) hipLaunchKernelGGL(lu_panel_kernel, dim3(nblk), dim3(256), lds, st, reinterpret_cast<dc*>(As[0]), n, k0, nb, rpb, ws, ipivs[0]);
  else {
    LuPanelBatch B;
    B.nsys = nsys;
    for (int t = 0; t < LU_BATCH_MAX; ++t) { const int q = t < nsys ? t : 0; B.A[t] = reinterpret_cast<dc*>(As[q]); B.ipiv[t] = ipivs[q]; B.ws[t] = wss[q]; }
    MA_REQUIRE(rpb <= 64, MA_ERR_INVALID, "the batched panel kernel holds <= 64 rows per workgroup (lane = row), got %d", rpb);
    hipLaunchKernelGGL(lu_panel_wave_kernel, dim3(nblk), dim3(64 * nsys), lds, st, B, n, k0, nb, rpb, (unsigned)sys_lds);
  }
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

int lu_launch_panel(c64* A, int n, int k0, int nb, int rpb, int nblk, int ncu, const LuPanelWs& ws, int* ipiv, bool clear_tags, hipStream_t st) {
  return launch_panel_any(0, 1, &A, n, k0, nb, rpb, nblk, ncu, &ws, &ipiv, clear_tags, st);
}
// the register-resident panel kernel: 256 rows per workgroup, nb <= LU_REG_NB columns; ncu = the CUs `st` may use
int lu_launch_panel_reg(c64* A, int n, int k0, int nb, int nblk, int ncu, const LuPanelWs& ws, int* ipiv, int* lists, bool clear_tags, hipStream_t st, c64* lrows, int lcol0,
                        const int* run_if_nonzero) {
  MA_REQUIRE(!lrows || (lcol0 >= 0 && lcol0 + LU_REG_NB <= k0), MA_ERR_INVALID, "left-half columns [%d, %d) not left of the panel at %d", lcol0, lcol0 + LU_REG_NB, k0);
  return launch_panel_any(2, 1, &A, n, k0, nb, 256, nblk, ncu, &ws, &ipiv, clear_tags, st, lists, lrows, lcol0, run_if_nonzero);
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
  const int p = lu_panel_slots_per_cu(lu_panel_reg_lds(), lu_panel_regs(2));
  MA_REQUIRE(p >= 1 && (long long)nblk <= (long long)p * ncu, MA_ERR_UNSUPPORTED, "register panel grid of %d workgroups cannot be co-resident on %d CUs (%d per CU)", nblk, ncu, p);
  return MA_OK;
}
// the same panel of nsys systems (a lock-step batch) in one co-resident grid, a wavefront per system: lu_panel_wave_kernel
int lu_launch_panel_batch(int nsys, c64* const* As, int n, int k0, int nb, int rpb, int nblk, int ncu, const LuPanelWs* wss, int* const* ipivs, bool clear_tags, hipStream_t st) {
  return launch_panel_any(1, nsys, As, n, k0, nb, rpb, nblk, ncu, wss, ipivs, clear_tags, st);
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
int lu_launch_block_row_moves(c64* A, int n, const int* lists, int lstride, int np, const int* k0s, const int* nbs, int x0, int x1, int y0, int y1, c64* B, int nrhs, const unsigned* poison, hipStream_t st) {
  const int ncol = (x1 - x0) + (y1 - y0);
  if (np <= 0 || (ncol <= 0 && nrhs <= 0)) return MA_OK;
  MA_REQUIRE(np <= 8 && nrhs <= 32 && x0 <= x1 && y0 <= y1, MA_ERR_INVALID, "block row moves: bad shape");
  LuBlockPanels P;
  P.np = np;
  int nb_max = 1;
  for (int q = 0; q < 8; ++q) { P.k0[q] = q < np ? k0s[q] : 0; P.nb[q] = q < np ? nbs[q] : 0; if (q < np && nbs[q] > nb_max) nb_max = nbs[q]; }
  for (int q = 0; q < np; ++q) MA_REQUIRE(nbs[q] >= 1 && nbs[q] <= LU_NB_MAX && (k0s[q] - x0) % 32 == 0, MA_ERR_INVALID, "block row moves: panel %d (%d columns from %d)", q, nbs[q], k0s[q]);
  MA_REQUIRE((x1 - x0) % 32 == 0 || y1 == y0, MA_ERR_INVALID, "block row moves: the left range must be whole strips");
  const int grid = (ncol + 31) / 32 + (nrhs > 0 ? 1 : 0);
  if (nb_max <= 64) hipLaunchKernelGGL(lu_block_row_moves_kernel<16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<dc*>(A), n, lists, lstride, P, x0, x1, y0, y1, reinterpret_cast<dc*>(B), nrhs, poison);
  else hipLaunchKernelGGL(lu_block_row_moves_kernel<32>, dim3(grid), dim3(256), 0, st, reinterpret_cast<dc*>(A), n, lists, lstride, P, x0, x1, y0, y1, reinterpret_cast<dc*>(B), nrhs, poison);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

int lu_launch_swaps(c64* A, int n, int k0, int nb, const int* ipiv, int* lists, c64* tmp, int tstride, int x0, int x1, int y0, int y1, c64* B, int nrhs,
                    c64* invd, unsigned* poison, hipStream_t st) {
  int rc = lu_launch_perm(A, n, k0, nb, ipiv, lists, invd, poison, st);
  if (rc) return rc;
  return lu_launch_row_moves(A, n, nb, lists, tmp, tstride, x0, x1, y0, y1, B, nrhs, st);
}

int lu_trsm_configure() {
  MA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(lu_trsm_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * TM_PITCH * 8));

  return MA_OK;
}

// X (nb x ncols, row stride ldx) <- L11^-1 X and the nrhs right-hand sides b_r = B + r*ldb (nb entries each) <- L11^-1 b_r,
// with the inverted diagonal blocks `invd` of lu_launch_swaps
int lu_launch_trsm_mfma(const c64* T, int ldt, int nb, const c64* invd, c64* X, size_t ldx, int ncols, c64* B, size_t ldb, int nrhs, hipStream_t st) {
  if (nb <= 0 || (ncols <= 0 && nrhs <= 0)) return MA_OK;
  MA_REQUIRE(nb <= LU_NB_MAX && nrhs <= 32, MA_ERR_DIM, "trsm: nb %d / nrhs %d beyond the kernel's tiles", nb, nrhs);
  const int nmain = ncols > 0 ? (ncols + 31) / 32 : 0;
  // LDS for the slab of this nb only (34.8 KB at nb = 64): the launch then fits on a CU that already holds two panel
  // workgroups (with the full 128-row 69.6 KB it never did, and queued behind them)
  static const bool wide_only_ = [] { const char* e = getenv("MA_LU_TRSM_WIDE"); return e && atoi(e) != 0; }();
  const size_t lds = (nb <= 64 && !wide_only_) ? 2 * (size_t)32 * TM_PITCH * 8 : 2 * (size_t)std::max(32, ((nb + 15) / 16) * 16) * TM_PITCH * 8;
  static const bool wide_only = [] { const char* e = getenv("MA_LU_TRSM_WIDE"); return e && atoi(e) != 0; }();   // diagnostic: the 8-tile kernel for every width
  auto kern = (nb <= 64 && !wide_only) ? lu_trsm64_kernel : lu_trsm_mfma_kernel;
  hipLaunchKernelGGL(kern, dim3(nmain + (nrhs > 0 ? 1 : 0)), dim3(128), lds, st, reinterpret_cast<const dc*>(T), ldt, nb,
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

// U12 of a whole block (np <= 8 panels of <= 64 columns each, first columns k0s[], widths nbs[]) on the columns X (row 0 = the block's
// first row) and the nrhs right-hand sides: in-block updates included (lu_block_trsm_kernel). invd: the panels' inverted diagonal blocks,
// invd_stride entries apart.
int lu_launch_block_trsm(const c64* A, int n, int np, const int* k0s, const int* nbs, const c64* invd, int invd_stride, c64* X, size_t ldx, int ncols, c64* B, size_t ldb, int nrhs, hipStream_t st) {
  if (np <= 0 || (ncols <= 0 && nrhs <= 0)) return MA_OK;
  MA_REQUIRE(np <= 8 && nrhs <= 32, MA_ERR_DIM, "block trsm: %d panels / %d right-hand sides beyond the kernel's limits", np, nrhs);
  LuBlockPanels P;
  P.np = np;
  for (int q = 0; q < 8; ++q) { P.k0[q] = q < np ? k0s[q] : 0; P.nb[q] = q < np ? nbs[q] : 0; }
  for (int q = 0; q < np; ++q) MA_REQUIRE(nbs[q] >= 1 && nbs[q] <= 64 && (q == 0 || k0s[q] == k0s[q - 1] + nbs[q - 1]), MA_ERR_DIM, "block trsm: panel %d (%d columns from %d)", q, nbs[q], k0s[q]);
  const int nmain = ncols > 0 ? (ncols + 63) / 64 : 0;
  hipLaunchKernelGGL(lu_block_trsm_kernel, dim3(nmain + (nrhs > 0 ? 1 : 0)), dim3(256), 0, st, reinterpret_cast<const dc*>(A), n, P, reinterpret_cast<const dc*>(invd), invd_stride,
                     reinterpret_cast<dc*>(X), ldx, (size_t)1, ncols, nmain, reinterpret_cast<dc*>(B), (size_t)1, ldb, nrhs);
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

ZgemmMode zgemm_mode_from_env() {
  ZgemmMode m;
  if (const char* e = getenv("MA_ZGEMM_DMA")) m.dma = atoi(e);
  if (const char* e = getenv("MA_ZGEMM_TILE_ORDER")) m.tile_order = atoi(e) != 0;
  if (const char* e = getenv("MA_ZGEMM_XCD_TILES")) m.xcd_min_tiles = atoi(e);
  if (const char* e = getenv("MA_ZGEMM_XCD_PERSIST")) m.persist = atoi(e) != 0;
  return m;
}

// `mode`: which kernel family runs the update. A plan resolves the switches ONCE, when it is created, and hands the same mode to
// every launch of its factorisations (the bitwise guarantees between schedules assume one family per factorisation, and getenv
// per launch raced with the tests' setenv under several host threads); NULL = the process-wide mode, read at its first use.
int lu_launch_zgemm_sub(int M, int N, int K, const c64* A, size_t lda, const c64* B, size_t ldb, c64* C, size_t ldc, hipStream_t st, bool use_3m, bool big, const ZgemmMode* mode) {
  if (M <= 0 || N <= 0 || K <= 0) return MA_OK;
  static const ZgemmMode process_mode = zgemm_mode_from_env();
  const ZgemmMode& zm = mode ? *mode : process_mode;
  if (use_3m && K % ZD_BK == 0 && zm.xcd_min_tiles <= 0) {               // an explicit request for the drawn-tile kernel wins
    const int dma_mode = zm.dma;
    if (dma_mode == 1 || dma_mode == 2) {
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
          MA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(zgemm3m_dma_kernel<4, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (128 + 128) * ZD_BK * 16));
          done[dev] = true;
        }
      }
      // big updates: a one-dimensional grid, tiles dealt out XCD by XCD in blocks of 4 x 4 (see the kernel); MA_ZGEMM_TILE_ORDER=0: plain
      const bool xcd_order = zm.tile_order && (long long)((N + 127) / 128) * ((M + 63) / 64) >= 512;
      const int T22 = ((N + 127) / 128) * ((M + 63) / 64);
      const dim3 g22 = xcd_order ? dim3(8 * ((T22 + 7) / 8), 1) : dim3((N + 127) / 128, (M + 63) / 64);
      if (dma_mode == 1 && big) hipLaunchKernelGGL((zgemm3m_dma_kernel<2, 2, true>), g22, dim3(256), 3 * (64 + 128) * ZD_BK * 16, st, M, N, K,
                                                   reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(B), ldb, reinterpret_cast<dc*>(C), ldc, xcd_order ? 1 : 0);
      else if (dma_mode == 1) hipLaunchKernelGGL((zgemm3m_dma_kernel<2, 2, false>), g22, dim3(256), 3 * (64 + 128) * ZD_BK * 16, st, M, N, K,
                                                 reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(B), ldb, reinterpret_cast<dc*>(C), ldc, xcd_order ? 1 : 0);
      else hipLaunchKernelGGL((zgemm3m_dma_kernel<4, 2, false>), dim3((N + 127) / 128, (M + 127) / 128), dim3(512), 3 * (128 + 128) * ZD_BK * 16, st, M, N, K,
                              reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(B), ldb, reinterpret_cast<dc*>(C), ldc, 0);
      MA_HIP(hipGetLastError());
      return MA_OK;
    }
  }
  if (use_3m) {
    dim3 g3((N + Z3_BN - 1) / Z3_BN, (M + Z3_BM - 1) / Z3_BM);
    // large updates draw their tiles XCD by XCD (see the kernel): a ring of counter blocks per device, one block per launch,
    // zeroed at allocation and again by the last workgroup of the launch that used it
    const int xcd_min_tiles = zm.xcd_min_tiles;                             // off unless asked for: it halves the fabric traffic and buys no time
    unsigned* ctr = nullptr;
    unsigned grid_draw = 0;
    if (xcd_min_tiles > 0 && (long long)g3.x * g3.y >= xcd_min_tiles) {
      int dev = 0;
      MA_HIP(hipGetDevice(&dev));
      if (dev >= 0 && dev < 16) {
        static std::mutex mu;
        static unsigned* ring[16] = {};
        static unsigned long long seq[16] = {};
        static int ncu[16] = {};
        std::lock_guard<std::mutex> lock(mu);
        if (!ring[dev]) {
          hipDeviceProp_t prop;
          MA_HIP(hipGetDeviceProperties(&prop, dev));
          ncu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
          MA_HIP(hipMalloc(&ring[dev], sizeof(unsigned) * 16 * 4096));
          MA_HIP(hipMemset(ring[dev], 0, sizeof(unsigned) * 16 * 4096));
          MA_HIP(hipDeviceSynchronize());
        }
        ctr = ring[dev] + 16 * (size_t)(seq[dev]++ % 4096ull);
        grid_draw = (unsigned)std::min<long long>((long long)g3.x * g3.y, 2LL * ncu[dev]);
      }
    }
    const bool one_tile = !zm.persist;
    if (ctr && one_tile) hipLaunchKernelGGL(zgemm3m_sub_drawn_kernel, dim3(g3.x * g3.y), dim3(256), 0, st, M, N, K, reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(B), ldb,
                                reinterpret_cast<dc*>(C), ldc, ctr, 1);
    else if (ctr) hipLaunchKernelGGL(zgemm3m_sub_drawn_kernel, dim3(grid_draw), dim3(256), 0, st, M, N, K, reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(B), ldb,
                                reinterpret_cast<dc*>(C), ldc, ctr, 0);
    else hipLaunchKernelGGL(zgemm3m_sub_kernel, g3, dim3(256), 0, st, M, N, K, reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(B), ldb,
                            reinterpret_cast<dc*>(C), ldc);
    MA_HIP(hipGetLastError());
    return MA_OK;
  }
  dim3 grid((N + ZG_BN - 1) / ZG_BN, (M + ZG_BM - 1) / ZG_BM), block(512);
  hipLaunchKernelGGL(zgemm_sub_kernel, grid, block, 0, st, M, N, K, reinterpret_cast<const dc*>(A), lda, reinterpret_cast<const dc*>(B), ldb,
                     reinterpret_cast<dc*>(C), ldc);
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
  std::lock_guard<std::mutex> lock(mu);
  seen.push_back({key, good});
  *ok = good;
  return MA_OK;
}

int lu_launch_mfma_probe(double* out, int blocks, int iters, hipStream_t st) {
  hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, st, out, iters);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

}  // namespace ma
