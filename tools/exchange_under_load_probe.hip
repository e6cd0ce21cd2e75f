// Diagnostic: latency of the panel kernel's exchange (two-level flag round over 227 co-resident workgroups, 8 groups) while
// a second stream keeps the chip busy: nothing / a streaming copy (HBM + vector-memory path) / matrix-core issue only /
// LDS traffic only. Variants of how the pollers read: agent-scope vector loads (what lu_panel_kernel does), or scalar loads
// (s_load ... glc: the scalar data path, not the CU's vector memory pipeline) on uncached memory.
// build: hipcc -O2 --offload-arch=gfx950 tools/exchange_under_load_probe.hip -o tools/exchange_under_load_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
typedef unsigned long long u64;
typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned sload_glc(const unsigned* p) {
  unsigned v;
  asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

// POLL 0: agent-scope vector loads; POLL 1: scalar loads with glc (memory must be uncached for cross-XCD visibility)
template <int POLL>
__global__ __launch_bounds__(64) void round_kernel(unsigned* flags, unsigned* xflags, int ngroups, int stride, int rounds, u64* out_ticks, unsigned* out_fail) {
  const int b = blockIdx.x, lane = threadIdx.x, G = gridDim.x;
  __builtin_amdgcn_s_setprio(3);
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  const int grp = b % ngroups, mem = b / ngroups, per = (G + ngroups - 1 - grp) / ngroups;
  for (int r = 1; r <= rounds && !fail; ++r) {
    if (lane == 0) __hip_atomic_store(flags + (size_t)(grp * 64 + mem) * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (mem == 0) {
      for (;;) {
        bool ok = true;
        for (int t = lane; t < per; t += 64) ok = ok && (__hip_atomic_load(flags + (size_t)(grp * 64 + t) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= (unsigned)r);
        if (__all(ok)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
      }
      if (lane == 0) __hip_atomic_store(xflags + (size_t)grp * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (;;) {
      bool ok = true;
      if (POLL == 0) {
        if (lane < ngroups) ok = __hip_atomic_load(xflags + (size_t)lane * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= (unsigned)r;
        ok = __all(ok);
      } else {
        for (int q = 0; q < ngroups; ++q) ok = ok && (sload_glc(xflags + (size_t)q * stride) >= (unsigned)r);
      }
      if (ok) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) out_ticks[0] = t1 - t0;
  if (lane == 0 && fail) out_fail[0] = 1;
}

// two-level round with PIPELINED polls: a new poll goes out every SP x 64 clocks whether or not the previous one has
// returned (up to 3 in flight), so the arrival of the awaited flag is seen one round trip + SP/2 later on average instead of
// one round trip + half a round trip
template <int SP>
__global__ __launch_bounds__(64) void round_pipe_kernel(unsigned* flags, unsigned* xflags, int ngroups, int stride, int rounds, u64* out_ticks, unsigned* out_fail) {
  const int b = blockIdx.x, lane = threadIdx.x, G = gridDim.x;
  __builtin_amdgcn_s_setprio(3);
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  const int grp = b % ngroups, mem = b / ngroups, per = (G + ngroups - 1 - grp) / ngroups;
  const unsigned* gp = flags + (size_t)(grp * 64 + (lane < per ? lane : 0)) * stride;
  const unsigned* xp = xflags + (size_t)(lane < ngroups ? lane : 0) * stride;
  for (int r = 1; r <= rounds && !fail; ++r) {
    if (lane == 0) __hip_atomic_store(flags + (size_t)(grp * 64 + mem) * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (mem == 0) {
      unsigned a = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_s_sleep(SP);
      unsigned c = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_s_sleep(SP);
      for (;;) {
        unsigned d = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__all(a >= (unsigned)r)) break;
        a = c; c = d;
        __builtin_amdgcn_s_sleep(SP);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
      }
      if (lane == 0) __hip_atomic_store(xflags + (size_t)grp * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    {
      unsigned a = __hip_atomic_load(xp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_s_sleep(SP);
      unsigned c = __hip_atomic_load(xp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __builtin_amdgcn_s_sleep(SP);
      for (;;) {
        unsigned d = __hip_atomic_load(xp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (__all(a >= (unsigned)r)) break;
        a = c; c = d;
        __builtin_amdgcn_s_sleep(SP);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
      }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) out_ticks[0] = t1 - t0;
  if (lane == 0 && fail) out_fail[0] = 1;
}

// two-level round on the SCALAR memory path only (uncached memory): flags of a group's members 4 B apart in one 256-B block,
// the 8 group flags in one 32-B block; stores s_store_dword glc, loads s_load_dwordx16 / x8 glc
typedef unsigned u8v __attribute__((ext_vector_type(8)));
typedef unsigned u16v __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void sstore(unsigned* p, unsigned v) {
  asm volatile("s_store_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" :: "s"(v), "s"(p) : "memory");
}
__global__ __launch_bounds__(64) void round_scalar_kernel(unsigned* flags, unsigned* xflags, int ngroups, int rounds, u64* out_ticks, unsigned* out_fail) {
  const int b = blockIdx.x, G = gridDim.x;
  __builtin_amdgcn_s_setprio(3);
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  const int grp = b % ngroups, mem = b / ngroups, per = (G + ngroups - 1 - grp) / ngroups;
  unsigned* mine = flags + grp * 64 + mem;
  const unsigned* gbase = flags + grp * 64;
  for (int r = 1; r <= rounds && !fail; ++r) {
    sstore(mine, (unsigned)r);
    if (mem == 0) {
      for (;;) {
        u16v a, c;
        asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(c) : "s"(gbase) : "memory");
        bool ok = true;
        for (int t = 0; t < 16; ++t) { if (t < per) ok = ok && a[t] >= (unsigned)r; if (16 + t < per) ok = ok && c[t] >= (unsigned)r; }
        if (ok) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
      }
      sstore(xflags + grp, (unsigned)r);
    }
    for (;;) {
      u8v a;
      asm volatile("s_load_dwordx8 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(a) : "s"(xflags) : "memory");
      bool ok = true;
      for (int t = 0; t < 8; ++t) if (t < ngroups) ok = ok && a[t] >= (unsigned)r;
      if (ok) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && b == 0) out_ticks[0] = t1 - t0;
  if (threadIdx.x == 0 && fail) out_fail[0] = 1;
}
// the same layout with vector accesses (agent-scope atomics), for comparison
__global__ __launch_bounds__(64) void round_packed_kernel(unsigned* flags, unsigned* xflags, int ngroups, int rounds, u64* out_ticks, unsigned* out_fail) {
  const int b = blockIdx.x, lane = threadIdx.x, G = gridDim.x;
  __builtin_amdgcn_s_setprio(3);
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  const int grp = b % ngroups, mem = b / ngroups, per = (G + ngroups - 1 - grp) / ngroups;
  for (int r = 1; r <= rounds && !fail; ++r) {
    if (lane == 0) __hip_atomic_store(flags + grp * 64 + mem, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (mem == 0) {
      for (;;) {
        bool ok = lane < per ? __hip_atomic_load(flags + grp * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= (unsigned)r : true;
        if (__all(ok)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
      }
      if (lane == 0) __hip_atomic_store(xflags + grp, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (;;) {
      bool ok = lane < ngroups ? __hip_atomic_load(xflags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= (unsigned)r : true;
      if (__all(ok)) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) out_ticks[0] = t1 - t0;
  if (lane == 0 && fail) out_fail[0] = 1;
}

// flat round: everyone polls everyone's flag
__global__ __launch_bounds__(64) void flat_kernel(unsigned* flags, int stride, int rounds, u64* out_ticks, unsigned* out_fail) {
  const int b = blockIdx.x, lane = threadIdx.x, G = gridDim.x;
  __builtin_amdgcn_s_setprio(3);
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  for (int r = 1; r <= rounds && !fail; ++r) {
    if (lane == 0) __hip_atomic_store(flags + (size_t)b * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (;;) {
      bool ok = true;
      for (int t = lane; t < G; t += 64) ok = ok && (__hip_atomic_load(flags + (size_t)t * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= (unsigned)r);
      if (__all(ok)) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) out_ticks[0] = t1 - t0;
  if (lane == 0 && fail) out_fail[0] = 1;
}

// backgrounds ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_kernel(const double2* __restrict__ src, double2* __restrict__ dst, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void mfma_kernel(double* out, int iters) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ __launch_bounds__(256) void lds_kernel(double* out, int iters) {
  __shared__ double2 buf[2048];                       // 32 KB, like one trailing-update workgroup
  for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = make_double2(i, -i);
  __syncthreads();
  double2 acc = make_double2(0, 0);
  int idx = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    double2 v = buf[idx]; acc.x += v.x; acc.y += v.y;
    buf[(idx + 256) & 2047] = acc;
    idx = (idx + 263) & 2047;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
}
// the same three together: global -> LDS -> matrix cores, like the update kernel
__global__ __launch_bounds__(256) void gemm_like_kernel(const double2* __restrict__ src, double* out, size_t n, int iters) {
  __shared__ double2 buf[2048];
  d4 a0 = {0, 0, 0, 0}, a1 = a0;
  size_t i0 = (blockIdx.x * (size_t)2048) % n;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = src[(i0 + i) % n];
    __syncthreads();
    for (int k = 0; k < 16; ++k) {
      const double2 v = buf[(threadIdx.x + 64 * k) & 2047];
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v.x, v.y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v.y, v.x, a1, 0, 0, 0);
    }
    __syncthreads();
    i0 = (i0 + (size_t)gridDim.x * 2048) % n;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1];
}

int main() {
  const int G = 227, NG = 8, rounds = 20000;
  unsigned *fl_c, *xf_c, *fl_u, *xf_u, *d_fail; u64* ticks;
  const size_t fbytes = 64 * 1024 * 4;
  CK(hipMalloc(&fl_c, fbytes)); CK(hipMalloc(&xf_c, fbytes));
  CK(hipExtMallocWithFlags((void**)&fl_u, fbytes, hipDeviceMallocUncached)); CK(hipExtMallocWithFlags((void**)&xf_u, fbytes, hipDeviceMallocUncached));
  CK(hipMalloc(&ticks, 8)); CK(hipMalloc(&d_fail, 4));
  const size_t n = (size_t)1 << 27;                   // 2 GiB of double2 per buffer
  double2 *src, *dst; double* outb;
  CK(hipMalloc(&src, n * 16)); CK(hipMalloc(&dst, n * 16)); CK(hipMalloc(&outb, 8 * 256 * 1024));
  CK(hipMemset(src, 0, n * 16));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  const char* bgname[] = {"idle chip", "streaming copy", "matrix cores only", "LDS only", "global->LDS->MFMA", "update kernel K=256", "update kernel K=64", "update kernel K=256, 1 wg/CU"};
  // the library's own trailing-update kernel as background, through the C-ABI
  typedef int (*zg_t)(int, int, int, const void*, const void*, void*, int, void*);
  void* lib = dlopen("math_audio_amd/lib/libmathaudio_hip.so", RTLD_NOW);
  zg_t zg = lib ? (zg_t)dlsym(lib, "ma_diag_zgemm_dev") : nullptr;
  if (!zg) printf("libmathaudio_hip.so not found: update-kernel backgrounds skipped\n");
  const int ZM = 8192;
  void *zA, *zB, *zC;
  CK(hipMalloc(&zA, (size_t)ZM * 256 * 16)); CK(hipMalloc(&zB, (size_t)ZM * 256 * 16)); CK(hipMalloc(&zC, (size_t)ZM * ZM * 16));
  CK(hipMemset(zA, 0, (size_t)ZM * 256 * 16)); CK(hipMemset(zB, 0, (size_t)ZM * 256 * 16)); CK(hipMemset(zC, 0, (size_t)ZM * ZM * 16));
  for (int bg = 0; bg < 8; ++bg) {
    if (bg >= 5 && !zg) continue;
    for (int var = 0; var < 12; ++var) {
      if (var != 1 && var < 9) continue;
      // var 0: cached memory, vector polls, flags 4 B apart ... var 1: one 128-B line per flag; var 2: uncached memory, vector polls; var 3: uncached, scalar polls
      unsigned* fl = (var == 2 || var == 3) ? fl_u : fl_c; unsigned* xf = (var == 2 || var == 3) ? xf_u : xf_c;
      const int stride = var == 0 ? 2 : 32;
      CK(hipMemset(fl_c, 0, fbytes)); CK(hipMemset(xf_c, 0, fbytes)); CK(hipMemset(fl_u, 0, fbytes)); CK(hipMemset(xf_u, 0, fbytes)); CK(hipMemset(ticks, 0, 8)); CK(hipMemset(d_fail, 0, 4));
      CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, sb));
      if (bg == 1) hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, sb, src, dst, n, 300);
      if (bg == 2) hipLaunchKernelGGL(mfma_kernel, dim3(512), dim3(256), 0, sb, outb, 6000000);
      if (bg == 3) hipLaunchKernelGGL(lds_kernel, dim3(512), dim3(256), 0, sb, outb, 12000000);
      if (bg == 4) hipLaunchKernelGGL(gemm_like_kernel, dim3(512), dim3(256), 0, sb, src, outb, n, 300000);
      if (bg == 5) zg(ZM, ZM, 256, zA, zB, zC, 250, sb);
      if (bg == 6) zg(ZM, ZM, 64, zA, zB, zC, 600, sb);
      if (bg == 7) zg(ZM, 1024, 256, zA, zB, zC, 1500, sb);        // 2048 tiles... a narrow update: fewer workgroups in flight per CU at the tail
      CK(hipEventRecord(e1, sb));
      if (var == 9) hipLaunchKernelGGL(round_scalar_kernel, dim3(G), dim3(64), 0, sa, fl_u, xf_u, NG, rounds, ticks, d_fail);
      else if (var == 10) hipLaunchKernelGGL(round_packed_kernel, dim3(G), dim3(64), 0, sa, fl_u, xf_u, NG, rounds, ticks, d_fail);
      else if (var == 11) hipLaunchKernelGGL(round_packed_kernel, dim3(G), dim3(64), 0, sa, fl_c, xf_c, NG, rounds, ticks, d_fail);
      else if (var == 6) hipLaunchKernelGGL(round_pipe_kernel<4>, dim3(G), dim3(64), 0, sa, fl, xf, NG, 32, rounds, ticks, d_fail);
      else if (var == 7) hipLaunchKernelGGL(round_pipe_kernel<8>, dim3(G), dim3(64), 0, sa, fl, xf, NG, 32, rounds, ticks, d_fail);
      else if (var == 8) hipLaunchKernelGGL(round_pipe_kernel<16>, dim3(G), dim3(64), 0, sa, fl, xf, NG, 32, rounds, ticks, d_fail);
      else if (var >= 4) hipLaunchKernelGGL(flat_kernel, dim3(G), dim3(64), 0, sa, fl, var == 4 ? 2 : 32, rounds, ticks, d_fail);
      else if (var == 3) hipLaunchKernelGGL(round_kernel<1>, dim3(G), dim3(64), 0, sa, fl, xf, NG, stride, rounds, ticks, d_fail);
      else hipLaunchKernelGGL(round_kernel<0>, dim3(G), dim3(64), 0, sa, fl, xf, NG, stride, rounds, ticks, d_fail);
      CK(hipStreamSynchronize(sa));
      const bool still = hipEventQuery(e1) == hipErrorNotReady;
      CK(hipDeviceSynchronize());
      float bgms = 0; CK(hipEventElapsedTime(&bgms, e0, e1));
      u64 t; unsigned f; CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, d_fail, 4, hipMemcpyDeviceToHost));
      const char* vn[] = {"cached, vector polls, 8-B spacing", "cached, vector polls, 128-B spacing", "uncached, vector polls", "uncached, scalar polls", "FLAT, 8-B spacing", "FLAT, 128-B spacing", "two-level, pipelined polls 4x64 clk", "two-level, pipelined polls 8x64 clk", "two-level, pipelined polls 16x64 clk", "SCALAR path, packed flags, uncached", "vector path, packed flags, uncached", "vector path, packed flags, cached"};
      printf("%-20s | %-36s | %7.3f us per round%s | background %7.1f ms%s\n", bgname[bg], vn[var], t / 100.0 / rounds, f ? " (TIMED OUT)" : "", bgms,
             bg == 0 ? "" : (still ? " (outlasted the rounds)" : " (ENDED EARLY)"));
      fflush(stdout);
    }
  }
  return 0;
}
