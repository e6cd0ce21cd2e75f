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
static_assert(SP_NB == 32, "one wavefront holds the top block: 32 rows in lanes 0..31");

// crecip_fast with selects for its branch (the same operations on the same operands): see lu_calu.hip
__device__ __forceinline__ dc sp_crecip(dc z) {
  const bool sw = !(__builtin_fabs(z.im) < __builtin_fabs(z.re));
  const double p = sw ? z.im : z.re, q = sw ? z.re : z.im;
  const double e = q * rcp_nr(p), g = rcp_nr(__builtin_fma(q, e, p));
  return sw ? dc_make(e * g, -g) : dc_make(g, -e * g);
}

template <int NB>
__global__ __launch_bounds__(64) void lu_spec_block_kernel(const dc* __restrict__ A, int n, int k0, int nbc, dc* __restrict__ u11, dc* __restrict__ rinv,
                                                           double* __restrict__ pivmag, int* __restrict__ order, int* __restrict__ verdict) {
  __shared__ __attribute__((aligned(16))) dc s_row[2][NB];
  __shared__ __attribute__((aligned(16))) dc s_ri[2];
  const int lane = threadIdx.x;
  const bool valid = lane < nbc;
  dc a[NB];
  {
    const dc* src = A + (size_t)(k0 + (valid ? lane : 0)) * n + k0;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = (valid && j < nbc) ? src[j] : dc_make(0.0, 0.0); });
  }
  if (lane == 0) __hip_atomic_store(verdict, 0, RLX_AGENT);
  int mypos = lane;                                          // the position (0..31) this lane's row holds under the interchanges so far
  bool done = !valid;
  bool bad = false;                                          // uniform: a column without a usable pivot -- the ordinary kernel decides what that means
  static_for<0, NB>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr int buf = c & 1;
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
    const unsigned pm = wave_umin(c2 ? (unsigned)mypos : 0xFFFFFFFFu);
    u64 wm = __ballot(c2 && (unsigned)mypos == pm);
    if (!wm) { if (live) bad = true; wm = __ballot(!done && mypos == c); }   // nothing to offer (NaNs): the row on the diagonal stands in
    const int wl = wm ? (int)__builtin_ctzll(wm) : 0;
    const bool iam = wm != 0ull && lane == wl;
    const dc rv = sp_crecip(a[c]);
    if (iam) {
      static_for<c, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; s_row[buf][j] = a[j]; });
      s_ri[buf] = rv;
    }
    __syncthreads();
    const dc piv = s_row[buf][c], ri = s_ri[buf];
    const bool singular = !(piv.re * piv.re + piv.im * piv.im >= 1e-60);
    if (live && (singular || !wm)) bad = true;
    if (live && wm) {
      const int wpos = __builtin_amdgcn_readlane(mypos, wl);
      if (iam) { done = true; mypos = c; }
      else if (!done && mypos == c) mypos = wpos;            // the row that sat on the diagonal takes the pivot row's place
      if (lane == 0) { rinv[c] = ri; pivmag[c] = cabs1(piv); }
    }
    const bool act = live && wm != 0ull && !done && !singular;
    const dc lf = a[c] * ri;
    const dc l = dc_make(act ? lf.re : 0.0, act ? lf.im : 0.0);
    a[c].re = act ? lf.re : a[c].re; a[c].im = act ? lf.im : a[c].im;
    const double nlr = -l.re, nli = -l.im, li = l.im;
    static_for<c + 1, NB>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const dc u = s_row[buf][j];
      a[j].re = __builtin_fma(li, u.im, __builtin_fma(nlr, u.re, a[j].re));
      a[j].im = __builtin_fma(nli, u.re, __builtin_fma(nlr, u.im, a[j].im));
    });
  });
  if (valid) {
    dc* dst = u11 + (size_t)mypos * NB;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; dst[j] = a[j]; });
    order[mypos] = lane;
  }
  if (bad && lane == 0) __hip_atomic_store(verdict, 1, RLX_AGENT);
}

struct SpecFinishLds {
  dc U[SP_NB][SP_NB];
  dc rinv[SP_NB];
  double pivmag[SP_NB];
  PivotSeqLds seq;
};

// One wavefront per workgroup, a lane per row of [k0, n): rows of the top block take their final content from the side buffer, the
// others go through the forward substitution with the check. Workgroup 0 also writes the pivots, the row list and (right half of a
// pair) the pivot rows' left-half entries.
template <int NB>
__global__ __launch_bounds__(64) void lu_spec_finish_kernel(dc* __restrict__ A, int n, int k0, int nbc, const dc* __restrict__ u11, const dc* __restrict__ rinv,
                                                            const double* __restrict__ pivmag, const int* __restrict__ order, int* __restrict__ verdict,
                                                            dc* __restrict__ backup, int* __restrict__ ipiv, int* __restrict__ lists, dc* __restrict__ lrows, int lcol0) {
  __shared__ __attribute__((aligned(16))) SpecFinishLds S;
  const int lane = threadIdx.x, b = (int)blockIdx.x;
  for (int idx = lane; idx < NB * NB; idx += 64) {
    const int i = idx / NB, j = idx % NB;
    S.U[i][j] = (i < nbc && j < nbc && j >= i) ? u11[(size_t)i * NB + j] : dc_make(0.0, 0.0);
  }
  if (lane < NB) {
    S.rinv[lane] = lane < nbc ? rinv[lane] : dc_make(0.0, 0.0);
    S.pivmag[lane] = lane < nbc ? pivmag[lane] : 0.0;
    if (b == 0) S.seq.win[lane] = lane < nbc ? k0 + order[lane] : -1;
  }
  __syncthreads();
  const int row = k0 + 64 * b + lane;
  const bool valid = row < n;
  dc a[NB];
  dc* p = A + (size_t)(valid ? row : k0) * n + k0;
  static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = (valid && j < nbc) ? p[j] : dc_make(0.0, 0.0); });
  if (valid) {
    dc* bk = backup + (size_t)(row - k0) * NB;
    static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; if (j < nbc) bk[j] = a[j]; });
  }
  if (b == 0) {
    pivot_sequence(S.seq, k0, nbc);
    __syncthreads();
    if (lane < nbc) ipiv[k0 + lane] = S.seq.ipiv[lane];
    if (lists) {
      if (lane == 0) lists[0] = S.seq.lm;
      if (lane < S.seq.lm) { lists[1 + lane] = S.seq.ldst[lane]; lists[1 + 2 * LU_NB_MAX + lane] = S.seq.lsrc[lane]; }
    }
    if (lrows && lane < nbc) {                             // as lu_panel_reg_kernel leaves them: row c = the pivot row of column c, its entries of the left half's columns
      const dc* lsrc = A + (size_t)S.seq.win[lane] * n + lcol0;
      dc* ldst = lrows + (size_t)lane * LU_REG_NB;
      static_for<0, NB>([&](auto jc) { constexpr int j = decltype(jc)::value; ldst[j] = lsrc[j]; });
    }
  }
  const bool top = row < k0 + nbc;
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
  if (__any(valid && viol) && lane == 0) atomicOr(verdict, 1);
}

__global__ __launch_bounds__(256) void lu_spec_restore_kernel(dc* __restrict__ A, int n, int k0, int nbc, const dc* __restrict__ backup, const int* __restrict__ verdict,
                                                              unsigned long long* __restrict__ stats /* [0] accepted, [1] rejected panels */) {
  const bool rejected = __hip_atomic_load(verdict, RLX_AGENT) != 0;
  if (stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(stats + (rejected ? 1 : 0), 1ull);
  if (!rejected) return;
  const long long total = (long long)(n - k0) * SP_NB;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int r = (int)(idx / SP_NB), j = (int)(idx % SP_NB);
    if (j < nbc) A[(size_t)(k0 + r) * n + k0 + j] = backup[idx];
  }
}

}  // namespace

int lu_launch_panel_spec(c64* A, int n, int k0, int nb, const LuSpecWs& ws, int* ipiv, int* lists, hipStream_t st, c64* lrows, int lcol0) {
  MA_REQUIRE(nb >= 1 && nb <= LU_REG_NB && k0 >= 0 && k0 + nb <= n, MA_ERR_INVALID, "panel [%d, %d) outside 0..%d", k0, k0 + nb, n);
  MA_REQUIRE(!lrows || (lcol0 >= 0 && lcol0 + LU_REG_NB <= k0), MA_ERR_INVALID, "left-half columns [%d, %d) not left of the panel at %d", lcol0, lcol0 + LU_REG_NB, k0);
  MA_REQUIRE(ws.u11 && ws.rinv && ws.pivmag && ws.order && ws.verdict && ws.backup && ws.rows >= n - k0, MA_ERR_INVALID, "speculative-panel workspace missing or too small");
  hipLaunchKernelGGL(lu_spec_block_kernel<LU_REG_NB>, dim3(1), dim3(64), 0, st, reinterpret_cast<const dc*>(A), n, k0, nb, reinterpret_cast<dc*>(ws.u11), reinterpret_cast<dc*>(ws.rinv),
                     ws.pivmag, ws.order, ws.verdict);
  MA_HIP(hipGetLastError());
  hipLaunchKernelGGL(lu_spec_finish_kernel<LU_REG_NB>, dim3((n - k0 + 63) / 64), dim3(64), 0, st, reinterpret_cast<dc*>(A), n, k0, nb, reinterpret_cast<const dc*>(ws.u11),
                     reinterpret_cast<const dc*>(ws.rinv), ws.pivmag, ws.order, ws.verdict, reinterpret_cast<dc*>(ws.backup), ipiv, lists, reinterpret_cast<dc*>(lrows), lcol0);
  MA_HIP(hipGetLastError());
  const long long total = (long long)(n - k0) * LU_REG_NB;
  int gx = (int)std::min<long long>((total + 255) / 256, 1024);
  hipLaunchKernelGGL(lu_spec_restore_kernel, dim3(gx), dim3(256), 0, st, reinterpret_cast<dc*>(A), n, k0, nb, reinterpret_cast<const dc*>(ws.backup), ws.verdict, ws.stats);
  MA_HIP(hipGetLastError());
  return MA_OK;
}

}  // namespace ma
