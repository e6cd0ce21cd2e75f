// Round 5 probe: what the pieces of one column of the tournament's elimination cost on one wavefront per SIMD (shader clocks, s_memtime
// around each piece, 256 threads, one workgroup): single-lane and full-wave LDS stores, broadcast LDS loads, the DPP reduction, the
// reciprocal, the barrier, 31 complex FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../math_audio_amd/csrc/lu_device.hpp"
using namespace ma;
#define T0() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define T1(i) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc[i] += t1 - t0
#define KEEP1(j) asm volatile("" : "+v"(a[j].re), "+v"(a[j].im));
#define KEEP8(j) KEEP1(j) KEEP1(j + 1) KEEP1(j + 2) KEEP1(j + 3) KEEP1(j + 4) KEEP1(j + 5) KEEP1(j + 6) KEEP1(j + 7)
__global__ __launch_bounds__(256) void probe(unsigned long long* out, double* sink, int sel_lane) {
  __shared__ __attribute__((aligned(16))) dc buf[4][32];
  __shared__ __attribute__((aligned(16))) unsigned keys[4];
  __shared__ __attribute__((aligned(16))) dc dump[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  dc a[32];
  for (int j = 0; j < 32; ++j) a[j] = dc_make(1.0 + tid + j, 0.5 * j - tid);
  unsigned long long acc[12] = {0}, t0, t1;
  for (int it = 0; it < 64; ++it) {
    T0(); T1(0);                                                          // [0] the stamp pair itself
    T0(); if (lane == sel_lane) { static_for<0, 32>([&](auto jc) { constexpr int j = decltype(jc)::value; buf[wave][j] = a[j]; }); } T1(1);     // [1] 32 single-lane b128 stores
    T0(); static_for<0, 8>([&](auto jc) { constexpr int j = decltype(jc)::value; reinterpret_cast<dc*>(buf)[(j * 256 + tid) % 128] = a[j]; }); T1(2);   // [2] 8 full-wave b128 stores
    T0(); static_for<0, 32>([&](auto jc) { constexpr int j = decltype(jc)::value; const dc u = buf[wave][j]; a[j].re += u.re; a[j].im += u.im; }); KEEP8(0) KEEP8(8) KEEP8(16) KEEP8(24) T1(3);   // [3] 32 broadcast loads + 2 adds each
    T0(); unsigned k = (unsigned)__double_as_longlong(a[5].re) >> 7; unsigned m = wave_umax(k | lane); asm volatile("" : "+s"(m)); T1(4); a[0].re += m;   // [4] one DPP reduction
    T0(); dc z = a[9]; const bool sw = !(__builtin_fabs(z.im) < __builtin_fabs(z.re)); const double p = sw ? z.im : z.re, q = sw ? z.re : z.im;
          const double e = q * rcp_nr(p), g = rcp_nr(__builtin_fma(q, e, p)); dc rv = sw ? dc_make(e * g, -g) : dc_make(g, -e * g); asm volatile("" : "+v"(rv.re), "+v"(rv.im)); T1(5); a[1].re += rv.re;   // [5] reciprocal
    T0(); __syncthreads(); T1(6);                                         // [6] barrier
    T0(); { const dc l = a[2]; static_for<0, 31>([&](auto jc) { constexpr int j = decltype(jc)::value; const dc u = a[(j + 7) & 31];
            a[j].re = __builtin_fma(l.im, u.im, __builtin_fma(-l.re, u.re, a[j].re)); a[j].im = __builtin_fma(-l.im, u.re, __builtin_fma(-l.re, u.im, a[j].im)); }); }
          KEEP8(0) KEEP8(8) KEEP8(16) KEEP8(24) T1(7);       // [7] 31 complex FMAs on registers
    T0(); if (lane == sel_lane) { static_for<0, 32>([&](auto jc) { constexpr int j = decltype(jc)::value; reinterpret_cast<double*>(buf[wave])[2 * j] = a[j].re; reinterpret_cast<double*>(buf[wave])[2 * j + 1] = a[j].im; }); } T1(8);   // [8] 64 single-lane b64 stores
    T0(); { dc* base = (lane == sel_lane) ? &buf[wave][0] : &dump[wave][lane]; const int stride = (lane == sel_lane) ? 1 : 0;
            static_for<0, 32>([&](auto jc) { constexpr int j = decltype(jc)::value; base[j * stride] = a[j]; }); } T1(10);   // [10] 32 full-wave b128 stores, one lane's address real, the others to a per-lane dump slot
    T0(); if (lane == sel_lane) { static_for<0, 32>([&](auto jc) { constexpr int j = decltype(jc)::value; reinterpret_cast<float*>(buf[wave])[j] = (float)a[j].re; }); } T1(11);   // [11] 32 single-lane b32 stores
    T0(); { const uint4 k4 = *reinterpret_cast<const uint4*>(keys); unsigned s = __builtin_amdgcn_readfirstlane(k4.x) | __builtin_amdgcn_readfirstlane(k4.w); asm volatile("" : "+s"(s)); a[3].re += s; } T1(9);   // [9] one b128 load + readfirstlane
  }
  double s = 0; for (int j = 0; j < 32; ++j) s += a[j].re + a[j].im;
  sink[blockIdx.x * 256 + tid] = s;
  if (tid == 0) for (int i = 0; i < 12; ++i) out[i] = acc[i] / 64;
}
int main() {
  unsigned long long* d; double* sink; hipMalloc(&d, 96); hipMalloc(&sink, 8 * 256);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d, sink, 17); hipDeviceSynchronize(); }
  unsigned long long h[12]; hipMemcpy(h, d, 96, hipMemcpyDeviceToHost);
  const char* nm[12] = {"stamp pair", "32 single-lane b128 stores", "8 full-wave b128 stores", "32 broadcast b128 loads", "DPP max reduction", "complex reciprocal", "barrier", "31 complex FMAs", "64 single-lane b64 stores", "b128 load + readfirstlane", "32 full-wave b128 stores (dump)", "32 single-lane b32 stores"};
  for (int i = 0; i < 12; ++i) printf("%-32s %6llu clocks (minus stamp: %lld)\n", nm[i], h[i], (long long)h[i] - (long long)h[0]);
  return 0;
}
