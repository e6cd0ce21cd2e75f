// Diagnostic: (1) which XCD each workgroup of a launch lands on (XCC_ID), (2) latency of an all-to-all flag round
// among co-resident workgroups: all 8 XCDs with agent-scope (sc1) accesses vs the workgroups of ONE XCD with
// workgroup-scope (sc0: served by the XCD's shared L2) accesses.
// build: hipcc -O2 --offload-arch=gfx950 tools/xcd_exchange_probe.hip -o tools/xcd_exchange_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;

// sc0 accesses: miss the per-CU vector cache, served by the XCD's L2 (coherent among the CUs of one XCD only)
__device__ __forceinline__ unsigned ld_sc0(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_sc0(unsigned* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

// MODE 0: every workgroup takes part, agent scope. MODE 1: only workgroups on XCD `want` (first `cap` claimers), sc0
// accesses (workgroup scope: may be served by the CU's own vector cache, i.e. NOT coherent between CUs). MODE 2: one XCD, agent scope.
template <int MODE>
__global__ __launch_bounds__(64) void round_kernel(unsigned* flags, unsigned* claim, int want, int cap, int rounds, u64* out_ticks, int* out_xcc, int* out_np) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const unsigned x = xcc_id();
  if (lane == 0) out_xcc[b] = (int)x;
  int me = b, np = gridDim.x;
  if (MODE >= 1) {
    if ((int)x != want) return;
    __shared__ int slot;
    if (lane == 0) slot = (int)atomicAdd(claim, 1u);
    __syncthreads();
    me = slot; np = cap;
    if (me >= cap) return;
  }
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  for (int r = 1; r <= rounds && !fail; ++r) {
    if (lane == 0) {
      if (MODE != 1) __hip_atomic_store(flags + me, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else st_sc0(flags + me, (unsigned)r);
    }
    for (;;) {
      bool ok = true;
      for (int t = lane; t < np; t += 64) {
        unsigned v;
        if (MODE != 1) v = __hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else v = ld_sc0(flags + t);
        ok = ok && (v >= (unsigned)r);
      }
      if (__all(ok)) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { fail = true; break; }   // 2 s: never hang
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && me == 0) { out_ticks[0] = t1 - t0; out_np[0] = fail ? -np : np; }
}

// MODE 3: all XCDs, agent scope, but only workgroups b < cap take part (b mod 8 spreads them over the XCDs).
// MODE 4: two-level round over all workgroups: every workgroup publishes; the leader of each XCD (its first claimer) waits
//         for its XCD's workgroups and publishes the XCD flag; every workgroup polls only the 8 XCD flags.
template <int MODE>
__global__ __launch_bounds__(64) void round2_kernel(unsigned* flags, unsigned* xflags, unsigned* claim, int cap, int rounds, u64* out_ticks, int* out_np) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const unsigned x = xcc_id();
  if (MODE == 3 && b >= cap) return;
  __shared__ int slot;
  if (MODE == 4) { if (lane == 0) slot = (int)atomicAdd(claim + x, 1u); __syncthreads(); }
  const int per = gridDim.x / 8;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  for (int r = 1; r <= rounds && !fail; ++r) {
    if (MODE == 3) {
      if (lane == 0) __hip_atomic_store(flags + b, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        bool ok = true;
        for (int t = lane; t < cap; t += 64) ok = ok && (__hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r);
        if (__all(ok)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { fail = true; break; }
      }
    } else {
      if (lane == 0) __hip_atomic_store(flags + x * 64 + slot, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slot == 0) {       // XCD leader: gather the XCD's workgroups, then publish the XCD flag
        for (;;) {
          bool ok = true;
          for (int t = lane; t < per; t += 64) ok = ok && (__hip_atomic_load(flags + x * 64 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r);
          if (__all(ok)) break;
          if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { fail = true; break; }
        }
        if (lane == 0) __hip_atomic_store(xflags + x * 16, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (;;) {
        bool ok = true;
        if (lane < 8) ok = __hip_atomic_load(xflags + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r;
        if (__all(ok)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { fail = true; break; }
      }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) { out_ticks[0] = t1 - t0; out_np[0] = fail ? -1 : (MODE == 3 ? cap : (int)gridDim.x); }
}

// MODE 5: generalised two-level round: group = b % ngroups (ngroups = 8: the XCDs), member = b / ngroups; flags `stride`
// words apart; every workgroup polls the ngroups group flags. ngroups = 0: flat round over all workgroups with that stride.
__global__ __launch_bounds__(64) void round3_kernel(unsigned* flags, unsigned* xflags, int ngroups, int stride, int rounds, u64* out_ticks) {
  const int b = blockIdx.x, lane = threadIdx.x, G = gridDim.x;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  const int grp = ngroups ? b % ngroups : 0, mem = ngroups ? b / ngroups : b, per = ngroups ? G / ngroups : G;
  for (int r = 1; r <= rounds && !fail; ++r) {
    if (lane == 0) __hip_atomic_store(flags + (size_t)(grp * per + mem) * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mem == 0 || ngroups == 0) {
      for (;;) {
        bool ok = true;
        for (int t = lane; t < per; t += 64) ok = ok && (__hip_atomic_load(flags + (size_t)(grp * per + t) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r);
        if (__all(ok)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { fail = true; break; }
      }
      if (ngroups && lane == 0) __hip_atomic_store(xflags + (size_t)grp * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ngroups) for (;;) {
      bool ok = true;
      if (lane < ngroups) ok = __hip_atomic_load(xflags + (size_t)lane * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r;
      if (__all(ok)) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { fail = true; break; }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) out_ticks[0] = fail ? 0 : t1 - t0;
}

// MODE 6: flat round over all workgroups (flags 8 B apart, like the panel kernel's granules) with a delay before the
// first poll and a sleep between polls (units of 64 clocks): do fewer, later polls shorten the round?
template <int D0, int DS>
__global__ __launch_bounds__(64) void round4_kernel(unsigned long long* flags, int rounds, u64* out_ticks) {
  const int b = blockIdx.x, lane = threadIdx.x, G = gridDim.x;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  for (int r = 1; r <= rounds && !fail; ++r) {
    if (lane == 0) __hip_atomic_store(flags + b, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (D0 > 0) __builtin_amdgcn_s_sleep(D0);
    for (;;) {
      bool ok = true;
      for (int t = lane; t < G; t += 64) ok = ok && (__hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)r);
      if (__all(ok)) break;
      if (DS > 0) __builtin_amdgcn_s_sleep(DS);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { fail = true; break; }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) out_ticks[0] = fail ? 0 : t1 - t0;
}
template <int D0, int DS>
static void run4(unsigned long long* f, u64* ticks, int rounds) {
  hipMemset(f, 0, 4096 * 8); hipMemset(ticks, 0, 8);
  hipLaunchKernelGGL((round4_kernel<D0, DS>), dim3(227), dim3(64), 0, 0, f, rounds, ticks);
  hipDeviceSynchronize();
  u64 t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  printf("flat, 227 workgroups, first poll after %3d x 64 clk, %3d x 64 clk between polls: %.3f us per round\n", D0, DS, t / 100.0 / rounds);
}

// MODE 7: one flat round that carries REC 8-byte flags per participant (the candidates of REC systems packed side by
// side, 8*REC bytes per workgroup): what a panel kernel batched over REC systems would exchange per column
template <int REC>
__global__ __launch_bounds__(64) void round5_kernel(unsigned long long* flags, int rounds, u64* out_ticks) {
  const int b = blockIdx.x, lane = threadIdx.x, G = gridDim.x;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  for (int r = 1; r <= rounds && !fail; ++r) {
    if (lane < REC) __hip_atomic_store(flags + (size_t)b * REC + lane, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_sleep(40);
    for (;;) {
      bool ok = true;
      for (int t = lane; t < G * REC; t += 64) ok = ok && (__hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned long long)r);
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(1);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { fail = true; break; }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) out_ticks[0] = fail ? 0 : t1 - t0;
}
template <int REC>
static void run5(unsigned long long* f, u64* ticks, int rounds) {
  hipMemset(f, 0, 4096 * 8); hipMemset(ticks, 0, 8);
  hipLaunchKernelGGL((round5_kernel<REC>), dim3(227), dim3(64), 0, 0, f, rounds, ticks);
  hipDeviceSynchronize();
  u64 t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
  printf("one flat round of 227 workgroups carrying %d flags each: %.3f us per round\n", REC, t / 100.0 / rounds);
}

int main() {
  unsigned *flags, *claim; u64* ticks; int *xcc, *np;
  const int G = 256;
  hipMalloc(&flags, 4096 * 4); hipMalloc(&claim, 64); hipMalloc(&ticks, 8); hipMalloc(&xcc, 4096 * 4); hipMalloc(&np, 4);
  std::vector<int> hx(4096);
  const int rounds = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(flags, 0, 4096 * 4); hipMemset(ticks, 0, 8);
    hipLaunchKernelGGL(round_kernel<0>, dim3(G), dim3(64), 0, 0, flags, claim, 0, 0, rounds, ticks, xcc, np);
    hipDeviceSynchronize();
    u64 t; int n; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy(&n, np, 4, hipMemcpyDeviceToHost);
    printf("all XCDs, agent scope (sc1): %d workgroups, %.3f us per round (%s)\n", n, t / 100.0 / rounds, hipGetErrorString(hipGetLastError()));
  }
  hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost);
  printf("XCC_ID of workgroups 0..31:");
  for (int i = 0; i < 32; ++i) printf(" %d", hx[i]);
  int cnt[16] = {0}; bool rr = true;
  for (int i = 0; i < G; ++i) { cnt[hx[i] & 15]++; if (hx[i] != hx[i % 8]) rr = false; }
  printf("\nper-XCD counts:"); for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
  printf("   workgroup b -> XCD is %s\n", rr ? "periodic in b mod 8" : "NOT periodic in b mod 8");
  for (int grid : {256, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(flags, 0, 4096 * 4); hipMemset(ticks, 0, 8); hipMemset(claim, 0, 4);
      const int cap2 = grid / 8;
      hipLaunchKernelGGL(round_kernel<2>, dim3(grid), dim3(64), 0, 0, flags, claim, 0, cap2, rounds, ticks, xcc, np);
      hipDeviceSynchronize();
      u64 t2; int n2; hipMemcpy(&t2, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy(&n2, np, 4, hipMemcpyDeviceToHost);
      printf("XCD 0 only, agent scope (sc1), grid %d: %d workgroups, %.3f us per round (%s)\n", grid, n2, t2 / 100.0 / rounds, hipGetErrorString(hipGetLastError()));
    }
    for (int rep = 0; rep < 1; ++rep) {
      hipMemset(flags, 0, 4096 * 4); hipMemset(ticks, 0, 8); hipMemset(claim, 0, 4);
      const int cap = grid / 8;
      hipLaunchKernelGGL(round_kernel<1>, dim3(grid), dim3(64), 0, 0, flags, claim, 0, cap, rounds, ticks, xcc, np);
      hipDeviceSynchronize();
      u64 t; int n; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy(&n, np, 4, hipMemcpyDeviceToHost);
      printf("XCD 0 only, workgroup scope (sc0), grid %d: %d workgroups, %.3f us per round (%s)\n", grid, n, t / 100.0 / rounds, hipGetErrorString(hipGetLastError()));
    }
  }
  unsigned* xflags; hipMalloc(&xflags, 4096);
  for (int cap : {8, 32, 64, 128}) {
    hipMemset(flags, 0, 4096 * 4); hipMemset(ticks, 0, 8);
    hipLaunchKernelGGL(round2_kernel<3>, dim3(256), dim3(64), 0, 0, flags, xflags, claim, cap, rounds, ticks, np);
    hipDeviceSynchronize();
    u64 t; int n; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy(&n, np, 4, hipMemcpyDeviceToHost);
    printf("all XCDs, agent scope, %d workgroups (spread over the XCDs): %.3f us per round\n", n, t / 100.0 / rounds);
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(flags, 0, 4096 * 4); hipMemset(xflags, 0, 4096); hipMemset(ticks, 0, 8); hipMemset(claim, 0, 64);
    hipLaunchKernelGGL(round2_kernel<4>, dim3(256), dim3(64), 0, 0, flags, xflags, claim, 0, rounds, ticks, np);
    hipDeviceSynchronize();
    u64 t; int n; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost); hipMemcpy(&n, np, 4, hipMemcpyDeviceToHost);
    printf("two-level (XCD leader gathers 32, everyone polls 8 XCD flags), %d workgroups: %.3f us per round\n", n, t / 100.0 / rounds);
  }
  unsigned *f2, *x2; hipMalloc(&f2, 256 * 64 * 4); hipMalloc(&x2, 64 * 64 * 4);
  for (int ng : {0, 8, 16, 32})
    for (int stride : {1, 2, 32}) {
      hipMemset(f2, 0, 256 * 64 * 4); hipMemset(x2, 0, 64 * 64 * 4); hipMemset(ticks, 0, 8);
      hipLaunchKernelGGL(round3_kernel, dim3(256), dim3(64), 0, 0, f2, x2, ng, stride, rounds, ticks);
      hipDeviceSynchronize();
      u64 t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      printf("256 workgroups, %2d groups, flags %3d B apart: %.3f us per round\n", ng, stride * 4, t / 100.0 / rounds);
    }
  unsigned long long* f8; hipMalloc(&f8, 4096 * 8);
  run5<1>(f8, ticks, rounds); run5<2>(f8, ticks, rounds); run5<3>(f8, ticks, rounds); run5<4>(f8, ticks, rounds);
  run4<0, 0>(f8, ticks, rounds); run4<0, 1>(f8, ticks, rounds); run4<0, 8>(f8, ticks, rounds); run4<0, 32>(f8, ticks, rounds);
  run4<20, 1>(f8, ticks, rounds); run4<40, 1>(f8, ticks, rounds); run4<60, 1>(f8, ticks, rounds); run4<40, 8>(f8, ticks, rounds);
  run4<60, 16>(f8, ticks, rounds); run4<80, 8>(f8, ticks, rounds);
  {  // two (three) independent flat rounds at the same time on separate streams and flag arrays: do they slow each other?
    hipStream_t sa[3]; unsigned long long* fl[3]; u64* tk[3];
    for (int i = 0; i < 3; ++i) { hipStreamCreateWithFlags(&sa[i], hipStreamNonBlocking); hipMalloc(&fl[i], 4096 * 8); hipMalloc(&tk[i], 8); }
    for (int nk = 1; nk <= 3; ++nk) {
      for (int i = 0; i < nk; ++i) { hipMemset(fl[i], 0, 4096 * 8); hipMemset(tk[i], 0, 8); }
      hipDeviceSynchronize();
      for (int i = 0; i < nk; ++i) hipLaunchKernelGGL((round4_kernel<40, 1>), dim3(227), dim3(64), 0, sa[i], fl[i], rounds, tk[i]);
      hipDeviceSynchronize();
      printf("%d concurrent flat rounds of 227 workgroups:", nk);
      for (int i = 0; i < nk; ++i) { u64 t; hipMemcpy(&t, tk[i], 8, hipMemcpyDeviceToHost); printf(" %.3f us", t / 100.0 / rounds); }
      printf(" per round\n");
    }
    unsigned *f3[3], *x3[3];
    for (int i = 0; i < 3; ++i) { hipMalloc(&f3[i], 256 * 64 * 4); hipMalloc(&x3[i], 64 * 64 * 4); }
    for (int nk = 1; nk <= 3; ++nk) {
      for (int i = 0; i < nk; ++i) { hipMemset(f3[i], 0, 256 * 64 * 4); hipMemset(x3[i], 0, 64 * 64 * 4); hipMemset(tk[i], 0, 8); }
      hipDeviceSynchronize();
      for (int i = 0; i < nk; ++i) hipLaunchKernelGGL(round3_kernel, dim3(224), dim3(64), 0, sa[i], f3[i], x3[i], 8, 32, rounds, tk[i]);
      hipDeviceSynchronize();
      printf("%d concurrent two-level rounds (8 groups, flags 128 B apart) of 224 workgroups:", nk);
      for (int i = 0; i < nk; ++i) { u64 t; hipMemcpy(&t, tk[i], 8, hipMemcpyDeviceToHost); printf(" %.3f us", t / 100.0 / rounds); }
      printf(" per round\n");
    }
  }
  return 0;
}
