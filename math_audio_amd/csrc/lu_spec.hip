// lu_spec.hip — the SPECULATIVE panel: partial pivoting without any exchange between workgroups, verified, with a fallback.
//
// zgetrf's pivot of column c is the largest |re| + |im| of the column below the diagonal after c eliminations (lu_solve,
// math-solvers/src/direct/lu.rs:142-153 -> LAPACK). Finding it is what makes a panel kernel of 40 workgroups exchange 32 times
// per panel (lu_panel_reg_kernel) or run a three-level tournament (lu_calu.hip). But the systems this library exists for --
// Burton-Miller boundary operators (tbem.rs:96-222: identity/2 + a weakly singular and a hypersingular operator) -- are strongly
// diagonal: LAPACK's pivot of every column IS inside the panel's top 32 rows (measured on S10 over the whole 64-frequency list:
// partial pivoting and the tournament return the same bits). So:
//
//   lu_spec_block_kernel   ONE wavefront factors the panel's top 32 x 32 block with partial pivoting among those 32 rows only
//                          (izamax on |re| + |im| in full precision, ties to the lower position: LAPACK's rule) -> L11 \ U11, the
//                          reciprocals and magnitudes of the pivots, the order of the rows (a side buffer; the matrix is not touched);
//   lu_spec_finish_kernel  every other row, a lane each: L = A U11^-1 column by column -- and before each column the CHECK
//                          |a(row, c)| <= |pivot c|: if it holds for every row and column, each pivot was the largest entry of
//                          its whole column, i.e. zgetrf would have chosen the same rows, and what has been written is LAPACK's panel.
//                          The rows' original entries go to a backup on the way;
//   lu_spec_restore_kernel if any row failed (the verdict word): the panel's columns are put back, and the caller's ordinary panel
//                          kernel -- launched behind with `run_if_nonzero = verdict` -- factors the panel. It returns at once otherwise.
//
// Accepted, a 10 000-row half-panel costs two short launches whose workgroups never wait for one another (no residency rule, no
// admission window, no CUs to keep free) instead of 130 us of a co-resident grid. Rejected, the price is one wasted pass.
#include "lu_kernels.hpp"
#include "lu_device.hpp"

namespace ma {

namespace {

constexpr int SP_NB = LU_REG_NB;
constexpr int SP_WIDENED_ATTEMPTS = 2;          // attempts after the first, each with the rows the previous one's check turned up
static_assert(SP_NB == 32, "one wavefront holds the top block: 32 rows in lanes 0..31");

// crecip_fast with selects for its branch (the same operations on the same operands): see lu_calu.hip
__device__ __forceinline__ dc sp_crecip(dc z) {
  const bool sw = !(__builtin_fabs(z.im) < __builtin_fabs(z.re));
  const double p = sw ? z.im : z.re, q = sw ? z.re : z.im;
  const double e = q * rcp_nr(p), g = rcp_nr(__builtin_fma(q, e, p));
  return sw ? dc_make(e * g, -g) : dc_make(g, -e * g);
}

// The device words of one slot's speculative panels (LuSpecWs::ctl), all cleared by the first attempt's block kernel
enum { SP_VERDICT = 0,   // 0: the panel is factored (accepted); != 0: not (yet)
       SP_VCOUNT = 1,    // rows below the top block that failed the first attempt's check (their indices: LuSpecWs::vlist, the first 32)
       SP_STAGE2 = 2,    // 1: the second attempt's block kernel produced a factorisation of the widened candidate set
       SP_ARRIVE = 3, SP_VIOL2 = 4,   // the second attempt's finish: workgroups done, any violation
       SP_NEXT = 5,      // positions below the top block that receive a displaced row (second attempt)
       SP_WORDS = 16 };

// Partial pivoting over the <= 64 candidate rows one wavefront holds (lane = row; `rowid` its index in the matrix, `mypos` the
// position it holds under the interchanges so far: LAPACK's izamax takes the FIRST largest |re| + |im| in position order).
// SECOND = false: the candidates are the panel's top rows, read from the matrix. SECOND = true (the widened attempt): the top rows plus
// the rows that failed the first attempt's check, read from the backup the first attempt's finish kernel left.
// Out (side buffers; the matrix is not written): u11[c][.] = row c of L11 \ U11, rinv / pivmag of the pivots, win[c] = the row that
// became pivot row c; and, replayed by pivot_sequence: the LAPACK-style pivots, the row list for the step kernels, the positions
// below the block that receive a displaced top row -- all tentative until the finish kernel's check has passed.
// a double of lane `wl` (uniform) to every lane through the scalar registers: no LDS round trip, no barrier
__device__ __forceinline__ double bcast_lane(double x, int wl) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), wl), __builtin_amdgcn_readlane(__double2loint(x), wl));
}

template <int NB, bool SECOND>
__global__ __launch_bounds__(64) void lu_spec_block_kernel(const dc* __restrict__ A, int n, int k0, int nbc, const dc* __restrict__ backup, dc* __restrict__ u11, dc* __restrict__ rinv,
                                                           double* __restrict__ pivmag, int* __restrict__ ctl, const int* __restrict__ vlist, int* __restrict__ ext /* [2][32]: position, source row */,
                                                           int* __restrict__ ipiv, int* __restrict__ lists, dc* __restrict__ lrows, int lcol0,
                                                           int* __restrict__ reject_info, unsigned long long* __restrict__ stats, int last_attempt) {
  __shared__ PivotSeqLds s_seq;
  const int lane = threadIdx.x;
  int rowid = -1;
  if constexpr (!SECOND) {
    if (lane < SP_WORDS) __hip_atomic_store(ctl + lane, 0, RLX_AGENT);
    if (lane == 0 && stats) atomicAdd(stats, 1ull);          // half-panels tried
    if (lane < nbc) rowid = k0 + lane;
  } else {
    if (__hip_atomic_load(ctl + SP_VERDICT, RLX_AGENT) == 0) return;          // the first attempt was accepted
    const int cnt = __hip_atomic_load(ctl + SP_VCOUNT, RLX_AGENT);
    if (lane == 0) { __hip_atomic_store(ctl + SP_STAGE2, 0, RLX_AGENT); __hip_atomic_store(ctl + SP_ARRIVE, 0, RLX_AGENT); __hip_atomic_store(ctl + SP_VIOL2, 0, RLX_AGENT); }
    if (cnt > 32) { if (lane == 0 && reject_info) __hip_atomic_store(reject_info, -1, RLX_AGENT); return; }   // too many rows want in: not a job for one wavefront
    if (lane < nbc) rowid = k0 + lane;
    else if (lane >= 32 && lane - 32 < cnt) rowid = vlist[lane - 32];
    if (rowid >= n || (lane >= 32 && rowid < k0 + nbc)) rowid = -1;
  }
  const bool valid = rowid >= 0;
  dc a[NB];
  {
    const dc* src = SECOND ? backup + (size_t)((valid ? rowid : k0) - k0) * NB : A + (size_t)(valid ? rowid : k0) * n + k0;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = (valid && j < nbc) ? src[j] : dc_make(0.0, 0.0); });
  }
  int mypos = rowid;                                         // position in the matrix
  bool done = !valid;
  bool bad = false;                                          // uniform: a column without a usable pivot -- the plan's own kernel decides what that means
  int rank = -1;
  static_for<0, NB>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const bool live = c < nbc;
    // izamax over the rows not yet chosen: the full 64 bits of |re| + |im| (a non-negative double orders as its bit pattern), ties
    // to the lower position
    const double mag = cabs1(a[c]);
    const bool ok = live && !done && mag == mag;
    const u64 bits = ok ? (u64)__double_as_longlong(mag) : 0ull;
    const unsigned hi = (unsigned)(bits >> 32), lo = (unsigned)bits;
    const unsigned mh = wave_umax(hi);
    const bool c1 = ok && hi == mh;
    const unsigned ml = wave_umax(c1 ? lo : 0u);
    const bool c2 = c1 && lo == ml;
    u64 wm = __ballot(c2);
    if (__popcll(wm) > 1) {                                  // equal magnitudes (rare): the lowest position
      const unsigned pm = wave_umin(c2 ? (unsigned)mypos : 0xFFFFFFFFu);
      wm = __ballot(c2 && (unsigned)mypos == pm);
    }
    if (!wm) { if (live) bad = true; wm = __ballot(!done && mypos == k0 + c); }   // nothing to offer (NaNs): the row on the diagonal stands in
    const int wl = wm ? (int)__builtin_ctzll(wm) : 0;
    const bool iam = wm != 0ull && lane == wl;
    // the pivot row reaches the other lanes through the scalar registers (v_readlane with a uniform lane): one wavefront, no LDS, no barrier
    const dc rv = sp_crecip(a[c]);
    const dc piv = dc_make(bcast_lane(a[c].re, wl), bcast_lane(a[c].im, wl));
    const dc ri = dc_make(bcast_lane(rv.re, wl), bcast_lane(rv.im, wl));
    if (lane == 0) s_seq.win[c] = (live && wm) ? __builtin_amdgcn_readlane(rowid, wl) : -1;
    const bool singular = !(piv.re * piv.re + piv.im * piv.im >= 1e-60);
    if (live && (singular || !wm)) bad = true;
    if (live && wm) {
      const int wpos = __builtin_amdgcn_readlane(mypos, wl);
      if (iam) { done = true; mypos = k0 + c; rank = c; }
      else if (!done && mypos == k0 + c) mypos = wpos;       // the row that sat on the diagonal takes the pivot row's place
      if (lane == 0) { rinv[c] = ri; pivmag[c] = cabs1(piv); }
    }
    const bool act = live && wm != 0ull && !done && !singular;
    const dc lf = a[c] * ri;
    const dc l = dc_make(act ? lf.re : 0.0, act ? lf.im : 0.0);
    a[c].re = act ? lf.re : a[c].re; a[c].im = act ? lf.im : a[c].im;
    const double nlr = -l.re, nli = -l.im, li = l.im;
    static_for<c + 1, NB>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const dc u = dc_make(bcast_lane(a[j].re, wl), bcast_lane(a[j].im, wl));
      a[j].re = __builtin_fma(li, u.im, __builtin_fma(nlr, u.re, a[j].re));
      a[j].im = __builtin_fma(nli, u.re, __builtin_fma(nlr, u.im, a[j].im));
    });
  });
  if (bad) {                                                 // no factorisation to offer: the panel stays (or becomes) rejected
    if (lane == 0) {
      if (!SECOND) { __hip_atomic_store(ctl + SP_VERDICT, 1, RLX_AGENT); if (stats) atomicAdd(stats + 1, 1ull); }
      __hip_atomic_store(ctl + SP_VCOUNT, 1000, RLX_AGENT);   // (no further attempt)
      if (reject_info) __hip_atomic_store(reject_info, -1, RLX_AGENT);
    }
    return;
  }
  if (rank >= 0) {
    dc* dst = u11 + (size_t)rank * NB;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; dst[j] = a[j]; });
    // right half of a 64-column panel (lu_plan.hip, pair form): the pivot rows' entries of the LEFT half's columns [lcol0, lcol0 + 32),
    // the block L10 the step after the panel solves with (as lu_panel_reg_kernel leaves them)
    if (lrows) {
      const dc* lsrc = A + (size_t)rowid * n + lcol0;
      dc* ldst = lrows + (size_t)rank * LU_REG_NB;
      static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; ldst[j] = lsrc[j]; });
    }
  }
  __syncthreads();
  pivot_sequence(s_seq, k0, nbc);
  __syncthreads();
  if (lane < nbc) ipiv[k0 + lane] = s_seq.ipiv[lane];
  if (lists) {
    if (lane == 0) lists[0] = s_seq.lm;
    if (lane < s_seq.lm) { lists[1 + lane] = s_seq.ldst[lane]; lists[1 + 2 * LU_NB_MAX + lane] = s_seq.lsrc[lane]; }
  }
  if (lane < s_seq.next) { ext[lane] = s_seq.ext_pos[lane]; ext[32 + lane] = s_seq.ext_src[lane]; }
  if (lane == 0) {
    __hip_atomic_store(ctl + SP_NEXT, s_seq.next, RLX_AGENT);
    if (SECOND) __hip_atomic_store(ctl + SP_STAGE2, 1, RLX_AGENT);
  }
}

struct SpecFinishLds {
  dc U[SP_NB][SP_NB];
  dc rinv[SP_NB];
  double pivmag[SP_NB];
  int ext_pos[SP_NB], ext_src[SP_NB];
};

// One wavefront per workgroup, a lane per position of [k0, n): positions of the top block take their final content (rows of L11 \ U11)
// from the side buffer, the others go through the forward substitution with the check.
// SECOND = false (first attempt): rows are read from the matrix and leave their original entries in the backup; rows that fail the
// check add themselves to the list of the widened attempt. SECOND = true: rows are read from the backup -- a position that receives a
// displaced top row reads THAT row --, and the workgroup that finishes last settles the verdict.
template <int NB, bool SECOND>
__global__ __launch_bounds__(64) void lu_spec_finish_kernel(dc* __restrict__ A, int n, int k0, int nbc, const dc* __restrict__ u11, const dc* __restrict__ rinv,
                                                            const double* __restrict__ pivmag, int* __restrict__ ctl, int* __restrict__ vlist, const int* __restrict__ ext,
                                                            dc* __restrict__ backup, int* __restrict__ reject_info, unsigned long long* __restrict__ stats, int last_attempt) {
  __shared__ __attribute__((aligned(16))) SpecFinishLds S;
  const int lane = threadIdx.x, b = (int)blockIdx.x;
  int next = 0;
  if constexpr (SECOND) {
    if (__hip_atomic_load(ctl + SP_VERDICT, RLX_AGENT) == 0 || __hip_atomic_load(ctl + SP_STAGE2, RLX_AGENT) == 0) return;   // (final before this grid starts: uniform)
    next = min(32, __hip_atomic_load(ctl + SP_NEXT, RLX_AGENT));
    if (lane < next) { S.ext_pos[lane] = ext[lane]; S.ext_src[lane] = ext[32 + lane]; }
  }
  // (first attempt, the block kernel had nothing to offer: the rows still go to the backup -- the restore behind a rejected panel copies it back)
  const bool backup_only = !SECOND && __hip_atomic_load(ctl + SP_VERDICT, RLX_AGENT) != 0;
  for (int idx = lane; idx < NB * NB; idx += 64) {
    const int i = idx / NB, j = idx % NB;
    S.U[i][j] = (i < nbc && j < nbc && j >= i) ? u11[(size_t)i * NB + j] : dc_make(0.0, 0.0);
  }
  if (lane < NB) {
    S.rinv[lane] = lane < nbc ? rinv[lane] : dc_make(0.0, 0.0);
    S.pivmag[lane] = lane < nbc ? pivmag[lane] : 0.0;
  }
  __syncthreads();
  const int row = k0 + 64 * b + lane;
  const bool valid = row < n;
  const bool top = row < k0 + nbc;
  dc a[NB];
  dc* p = A + (size_t)(valid ? row : k0) * n + k0;
  int src_row = row;                                       // the row (of the backup) this position's content comes from
  if constexpr (!SECOND) {
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = (valid && j < nbc) ? p[j] : dc_make(0.0, 0.0); });
    if (valid) {
      dc* bk = backup + (size_t)(row - k0) * NB;
      static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if (j < nbc) bk[j] = a[j]; });
    }
    if (backup_only) return;
  } else {
    for (int e = 0; e < next; ++e) if (S.ext_pos[e] == row) src_row = S.ext_src[e];
    const dc* bk = backup + (size_t)((valid ? src_row : k0) - k0) * NB;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = (valid && j < nbc) ? bk[j] : dc_make(0.0, 0.0); });
  }
  bool viol = false;
  if (top) {
    const dc* src = u11 + (size_t)(row - k0) * NB;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = src[j]; });
  } else {
    static_for<0, NB>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if (c < nbc) {
        viol = viol || !(cabs1(a[c]) <= S.pivmag[c]);      // a larger entry below the block (or a NaN): zgetrf would not have taken this pivot
        const dc l = a[c] * S.rinv[c];
        a[c] = l;
        const double nlr = -l.re, nli = -l.im, li = l.im;
        static_for<c + 1, NB>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          const dc u = S.U[c][j];
          a[j].re = __builtin_fma(li, u.im, __builtin_fma(nlr, u.re, a[j].re));
          a[j].im = __builtin_fma(nli, u.re, __builtin_fma(nlr, u.im, a[j].im));
        });
      }
    });
  }
  if (valid) static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if (j < nbc) p[j] = a[j]; });
  viol = viol && valid;
  // a row that fails the check joins the candidates of the next attempt (a candidate cannot fail: the pivot is the largest of them)
  if (viol) { const int idx = atomicAdd(ctl + SP_VCOUNT, 1); if (idx < 32) vlist[idx] = SECOND ? src_row : row; }
  if constexpr (!SECOND) {
    if (__any(viol) && lane == 0) {
      if (atomicOr(ctl + SP_VERDICT, 1) == 0 && stats) atomicAdd(stats + 1, 1ull);   // the panel's first violation: the first attempt is rejected
    }
    // (whether the widened attempt can take it -- <= 32 rows in the list -- is for its block kernel to see; more than that, and the
    // status word of an optimistic plan is set there)
  } else {
    if (__any(viol) && lane == 0) atomicOr(ctl + SP_VIOL2, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) {
      const int old = atomicAdd(ctl + SP_ARRIVE, 1);
      if (old == (int)gridDim.x - 1) {                       // the last workgroup settles it
        const int v2 = __hip_atomic_load(ctl + SP_VIOL2, RLX_AGENT);
        if (v2 == 0) { __hip_atomic_store(ctl + SP_VERDICT, 0, RLX_AGENT); if (stats) atomicAdd(stats + 2, 1ull); }   // accepted by the widened attempt
        else if (reject_info && (last_attempt || __hip_atomic_load(ctl + SP_VCOUNT, RLX_AGENT) > 32)) __hip_atomic_store(reject_info, -1, RLX_AGENT);
      }
    }
  }
}

__global__ __launch_bounds__(256) void lu_spec_restore_kernel(dc* __restrict__ A, int n, int k0, int nbc, const dc* __restrict__ backup, const int* __restrict__ verdict) {
  if (__hip_atomic_load(verdict, RLX_AGENT) == 0) return;
  const long long total = (long long)(n - k0) * SP_NB;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int r = (int)(idx / SP_NB), j = (int)(idx % SP_NB);
    if (j < nbc) A[(size_t)(k0 + r) * n + k0 + j] = backup[idx];
  }
}

}  // namespace

// The two attempts (block + finish each; the second pair returns at once where the first was accepted), then -- optimistic = false --
// the restore of a panel both attempts had to give up, behind which the caller launches its own panel kernel gated by ws.ctl
// (run_if_nonzero). optimistic = true: no restore, nothing behind: a panel given up leaves -1 in *reject_info (the system's status
// word) and the CALLER solves that system again in the verified mode.
int lu_launch_panel_spec(c64* A, int n, int k0, int nb, const LuSpecWs& ws, int* ipiv, int* lists, hipStream_t st, c64* lrows, int lcol0, bool optimistic, int* reject_info) {
  MA_REQUIRE(nb >= 1 && nb <= LU_REG_NB && k0 >= 0 && k0 + nb <= n, MA_ERR_INVALID, "panel [%d, %d) outside 0..%d", k0, k0 + nb, n);
  MA_REQUIRE(!lrows || (lcol0 >= 0 && lcol0 + LU_REG_NB <= k0), MA_ERR_INVALID, "left-half columns [%d, %d) not left of the panel at %d", lcol0, lcol0 + LU_REG_NB, k0);
  MA_REQUIRE(ws.u11 && ws.rinv && ws.pivmag && ws.ctl && ws.vlist && ws.ext && ws.backup && ws.rows >= n - k0, MA_ERR_INVALID, "speculative-panel workspace missing or too small");
  MA_REQUIRE(!optimistic || reject_info, MA_ERR_INVALID, "the optimistic form needs the system's status word");
  int* rj = optimistic ? reject_info : nullptr;
  dc* dA = reinterpret_cast<dc*>(A); dc* u11 = reinterpret_cast<dc*>(ws.u11); dc* ri = reinterpret_cast<dc*>(ws.rinv); dc* bk = reinterpret_cast<dc*>(ws.backup);
  const dim3 fgrid((n - k0 + 63) / 64);
  hipLaunchKernelGGL((lu_spec_block_kernel<LU_REG_NB, false>), dim3(1), dim3(64), 0, st, dA, n, k0, nb, bk, u11, ri, ws.pivmag, ws.ctl, ws.vlist, ws.ext, ipiv, lists, reinterpret_cast<dc*>(lrows), lcol0, rj, ws.stats, 0);
  hipLaunchKernelGGL((lu_spec_finish_kernel<LU_REG_NB, false>), fgrid, dim3(64), 0, st, dA, n, k0, nb, u11, ri, ws.pivmag, ws.ctl, ws.vlist, ws.ext, bk, rj, ws.stats, 0);
  for (int t = 0; t < SP_WIDENED_ATTEMPTS; ++t) {           // each returns at once where the panel has been accepted
    const int last = t == SP_WIDENED_ATTEMPTS - 1;
    hipLaunchKernelGGL((lu_spec_block_kernel<LU_REG_NB, true>), dim3(1), dim3(64), 0, st, dA, n, k0, nb, bk, u11, ri, ws.pivmag, ws.ctl, ws.vlist, ws.ext, ipiv, lists, reinterpret_cast<dc*>(lrows), lcol0, rj, ws.stats, last);
    hipLaunchKernelGGL((lu_spec_finish_kernel<LU_REG_NB, true>), fgrid, dim3(64), 0, st, dA, n, k0, nb, u11, ri, ws.pivmag, ws.ctl, ws.vlist, ws.ext, bk, rj, ws.stats, last);
  }
  MA_HIP(hipGetLastError());
  if (optimistic) return MA_OK;
  const long long total = (long long)(n - k0) * LU_REG_NB;
  int gx = (int)std::min<long long>((total + 255) / 256, 1024);
  hipLaunchKernelGGL(lu_spec_restore_kernel, dim3(gx), dim3(256), 0, st, dA, n, k0, nb, reinterpret_cast<const dc*>(ws.backup), ws.ctl);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

}  // namespace ma
