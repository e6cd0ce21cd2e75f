// ma_device_math.hpp — f64 device primitives for the gfx950 kernels (64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>

namespace ma {

struct dc { double re, im; };   // complex128 in registers

__device__ __forceinline__ dc dc_make(double re, double im) { dc z; z.re = re; z.im = im; return z; }
__device__ __forceinline__ dc operator+(dc a, dc b) { return dc_make(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ dc operator-(dc a, dc b) { return dc_make(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ dc operator*(dc a, dc b) { return dc_make(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ __forceinline__ dc operator*(dc a, double s) { return dc_make(a.re * s, a.im * s); }
__device__ __forceinline__ dc dc_neg(dc a) { return dc_make(-a.re, -a.im); }

// sqrt and reciprocal sqrt of a positive, normal double: v_rsq_f64 seed + two coupled
// Goldschmidt steps (g -> sqrt(x), h -> 0.5/sqrt(x)); both results within ~1 ulp.
__device__ __forceinline__ void sqrt_rsqrt(double x, double& s, double& rs) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = 0.5 * y;
  double r = __builtin_fma(-g, h, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-g, h, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  // one residual correction on g for the last bit
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  s = g;
  rs = h + h;
}

// sin and cos of a moderate argument (|x| < 2^30): two-FMA Cody–Waite reduction by pi/2
// (exact first step because of the cancellation) and the classic degree-13/14
// minimax kernels on [-pi/4, pi/4] (published fdlibm k_sin/k_cos coefficients).
// sincos_bounded: the caller guarantees |x| < 2^30 (the BEM kernels: k r with r inside the mesh's bounding box, checked per
// frequency on the host) -- without the library fall-back in the instruction stream the assembly runs 3 % and the streamed
// operator 11 % faster. sincos_fast: larger or non-finite arguments take the library path.
__device__ __forceinline__ void sincos_bounded(double x, double& sn, double& cs) {
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632679489655800e+00;
  const double pio2_lo = 6.12323399573676603587e-17;
  double fn = __builtin_rint(x * two_over_pi);
  double r = __builtin_fma(-fn, pio2_hi, x);
  r = __builtin_fma(-fn, pio2_lo, r);
  int q = (int)fn;
  double z = r * r;
  // sin kernel
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double ps = __builtin_fma(z, S6, S5);
  ps = __builtin_fma(z, ps, S4);
  ps = __builtin_fma(z, ps, S3);
  ps = __builtin_fma(z, ps, S2);
  ps = __builtin_fma(z, ps, S1);
  double sr = __builtin_fma(r * z, ps, r);
  // cos kernel
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double pc = __builtin_fma(z, C6, C5);
  pc = __builtin_fma(z, pc, C4);
  pc = __builtin_fma(z, pc, C3);
  pc = __builtin_fma(z, pc, C2);
  pc = __builtin_fma(z, pc, C1);
  double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
  // quadrant
  double s0 = (q & 1) ? cr : sr;
  double c0 = (q & 1) ? sr : cr;
  sn = (q & 2) ? -s0 : s0;
  cs = ((q + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ void sincos_fast(double x, double& sn, double& cs) {
  if (!(__builtin_fabs(x) < 1073741824.0)) {   // wave-uniformly false for acoustic kr
    sincos(x, &sn, &cs);
    return;
  }
  sincos_bounded(x, sn, cs);
}

// butterfly sum over the 64 lanes of a wavefront; every lane ends with the total
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// sum of a double over the wavefront, valid in lane 63 (DPP shifts inside the rows of 16 lanes, then the two row broadcasts;
// lanes that receive nothing add 0)
__device__ __forceinline__ double wave_sum_lane63(double v) {
#define MA_DPP_ADD(ctrl, rmask)                                                                                         \
  {                                                                                                                     \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, false);                          \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, false);                          \
    v += __hiloint2double(hi, lo);                                                                                      \
  }
  MA_DPP_ADD(0x111, 0xf) MA_DPP_ADD(0x112, 0xf) MA_DPP_ADD(0x114, 0xf) MA_DPP_ADD(0x118, 0xf)   // row_shr 1, 2, 4, 8: lane 15 of a row = row sum
  MA_DPP_ADD(0x142, 0xa) MA_DPP_ADD(0x143, 0xc)                                                 // row_bcast 15 / 31: lane 63 = total
#undef MA_DPP_ADD
  return v;
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
  unsigned lane = threadIdx.x & 63u;
  return lane == 0 ? 0ull : (~0ull >> (64u - lane));
}

}  // namespace ma
