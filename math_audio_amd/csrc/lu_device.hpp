// lu_device.hpp — device helpers shared by the dense-LU kernel files (lu_kernels.hip, lu_calu.hip).
#pragma once
#include "ma_device_math.hpp"
#include <type_traits>

namespace ma {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#ifndef LU_GROUPS
#define LU_GROUPS 8              // candidate-gather groups of the panel kernel (workgroup b -> group b % 8 = its XCD)
#endif
#ifndef LU_POLL_SLEEP
#define LU_POLL_SLEEP 1          // x 64 clocks between two sweeps of the group granules
#endif
#ifndef LU_GRANULE_STRIDE
#define LU_GRANULE_STRIDE 16
#endif

__device__ __forceinline__ void st_sc1(u64* p, double v) { __hip_atomic_store(p, (u64)__double_as_longlong(v), RLX_AGENT); }
__device__ __forceinline__ double ld_sc1(const u64* p) { return __longlong_as_double((long long)__hip_atomic_load(p, RLX_AGENT)); }
__device__ __forceinline__ double cabs1(dc z) { return __builtin_fabs(z.re) + __builtin_fabs(z.im); }

// reciprocal of a complex number the way LAPACK forms ONE / A(j,j) (Smith's division)
__device__ __forceinline__ dc crecip(dc z) {
  if (__builtin_fabs(z.im) < __builtin_fabs(z.re)) {
    double e = z.im / z.re, f = z.re + z.im * e;
    return dc_make(1.0 / f, -e / f);
  }
  double e = z.re / z.im, f = z.im + z.re * e;
  return dc_make(e / f, -1.0 / f);
}

// the same reciprocal with v_rcp_f64 + two Newton steps in place of the three IEEE divisions (about 600 cycles on the critical
// path of every column of lu_panel_reg_kernel): within 2 ulp of crecip
__device__ __forceinline__ double rcp_nr(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ dc crecip_fast(dc z) {
  if (__builtin_fabs(z.im) < __builtin_fabs(z.re)) {
    const double e = z.im * rcp_nr(z.re), g = rcp_nr(__builtin_fma(z.im, e, z.re));
    return dc_make(g, -e * g);
  }
  const double e = z.re * rcp_nr(z.im), g = rcp_nr(__builtin_fma(z.re, e, z.im));
  return dc_make(e * g, -g);
}

// maximum of an unsigned value over the wavefront (DPP shifts inside the rows of 16 lanes, then the two row broadcasts);
// lanes that receive nothing contribute 0
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8 -> lane 15 of a row holds the row's maximum
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ unsigned wave_umin(unsigned v) { return ~wave_umax(~v); }

// compile-time loop: f(integral_constant<int, I>) for I = I0 .. N-1. Every index into the row registers is a constant when the
// code is generated (a runtime-indexed array of 4 NB registers would live in scratch memory: the unroller alone left it there)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}


// LAPACK-style pivots and the (destination, source) row list of a panel whose pivot rows are known: win[c] = the row (its index in
// the matrix at the panel's start) that becomes pivot row c, c = 0..nbc-1 <= 32. One wavefront (all 64 lanes) replays the
// interchange sequence "position k0 + c <-> wherever row win[c] is now": lanes 0..31 track what the panel's top positions hold,
// lanes 32.. the positions below that a swap has touched. Out: ipiv[c] (position swapped with k0 + c), the moved rows as
// (ldst[i], lsrc[i]), i < lm ("position ldst holds what row lsrc held"), and the touched positions below the top block as
// (ext_pos[e], ext_src[e]), e < next: ext_src are top-block rows that did not become pivot rows.
struct PivotSeqLds {
  int win[32];
  int ipiv[32], ext_pos[32], ext_src[32], next;
  int ldst[64], lsrc[64], lm;
};
__device__ __forceinline__ void pivot_sequence(PivotSeqLds& S, int k0, int nbc) {
  const int lane = threadIdx.x & 63;
  int pos = lane < 32 ? k0 + lane : -1, content = pos;
  bool act = lane < nbc;
  int next = 0;
  for (int c = 0; c < nbc; ++c) {
    const int gc = k0 + c, r = S.win[c];
    if (r < 0) { if (lane == 0) S.ipiv[c] = gc; continue; }
    const u64 hm = __ballot(act && content == r);
    const int hl = hm ? (int)__builtin_ctzll(hm) : -1;
    const int P = hl >= 0 ? __builtin_amdgcn_readlane(pos, hl) : r;       // where row r is now
    if (lane == 0) S.ipiv[c] = P;
    const int old = __builtin_amdgcn_readlane(content, c);
    if (P != gc) {
      if (hl >= 0) { if (lane == hl) content = old; }
      else { if (lane == 32 + next) { pos = P; content = old; act = true; } ++next; }
      if (lane == c) content = r;
    }
  }
  const bool keep = act && content != pos;
  const u64 km = __ballot(keep);
  if (keep) { const int o = __popcll(km & (((u64)1 << lane) - 1)); S.ldst[o] = pos; S.lsrc[o] = content; }
  if (lane >= 32 && lane < 32 + next) { S.ext_pos[lane - 32] = pos; S.ext_src[lane - 32] = content; }
  if (lane == 0) { S.lm = __popcll(km); S.next = next; }
}

}  // namespace ma
