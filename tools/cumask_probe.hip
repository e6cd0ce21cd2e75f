// Diagnostic (round 3): do CU-masked streams (hipExtStreamCreateWithCUMask) give the latency-bound panel exchange CUs of its own?
//  1. census: which (XCC, SE, CU) the workgroups of a kernel land on under a mask (how the mask's bits map to the chip);
//  2. the two-level / flat flag round of the panel kernel on a masked stream, alone and beside the library's trailing-update
//     kernel on (a) an unmasked stream (today's schedule) and (b) the complementary mask (disjoint CUs);
//  3. what the update kernel loses when it runs on 256 - P CUs.
// build: hipcc -O2 --offload-arch=gfx950 tools/cumask_probe.hip -o tools/cumask_probe.bin
// run from the repository root (it loads math_audio_amd/lib/libmathaudio_hip.so for the update kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <set>
#include <map>
#include <vector>
typedef unsigned long long u64;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void census_kernel(unsigned* out, int spin_ticks) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (u64)spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

// two-level round (ngroups group leaders gather their members, everyone polls the group flags); ngroups == 0: flat round
__global__ __launch_bounds__(64) void round_kernel(unsigned* flags, unsigned* xflags, int ngroups, int stride, int rounds, u64* out_ticks, unsigned* out_fail) {
  const int b = blockIdx.x, lane = threadIdx.x, G = gridDim.x;
  __builtin_amdgcn_s_setprio(3);
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  if (ngroups == 0) {
    for (int r = 1; r <= rounds && !fail; ++r) {
      if (lane == 0) __hip_atomic_store(flags + (size_t)b * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        bool ok = true;
        for (int t = lane; t < G; t += 64) ok = ok && (__hip_atomic_load(flags + (size_t)t * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r);
        if (__all(ok)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
      }
    }
  } else {
    const int grp = b % ngroups, mem = b / ngroups, per = (G + ngroups - 1 - grp) / ngroups;
    for (int r = 1; r <= rounds && !fail; ++r) {
      if (lane == 0) __hip_atomic_store(flags + (size_t)(grp * 64 + mem) * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (mem == 0) {
        for (;;) {
          bool ok = true;
          for (int t = lane; t < per; t += 64) ok = ok && (__hip_atomic_load(flags + (size_t)(grp * 64 + t) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r);
          if (__all(ok)) break;
          if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
        }
        if (lane == 0) __hip_atomic_store(xflags + (size_t)grp * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (;;) {
        bool ok = true;
        if (lane < ngroups) ok = __hip_atomic_load(xflags + (size_t)lane * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r;
        if (__all(ok)) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
      }
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) out_ticks[0] = t1 - t0;
  if (lane == 0 && fail) out_fail[0] = 1;
}

// a round that also moves a row: every workgroup publishes 1 KB (64 lanes x 16 B, sc1) + its flag, then polls all flags
// (flat, G <= 64) and fetches the 1 KB of workgroup (r % G) -- the panel kernel's column step without its arithmetic
__global__ __launch_bounds__(64) void row_round_kernel(unsigned* flags, double2* rows, int stride, int rounds, u64* out_ticks, unsigned* out_fail, double* sink) {
  const int b = blockIdx.x, lane = threadIdx.x, G = gridDim.x;
  __builtin_amdgcn_s_setprio(3);
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  bool fail = false;
  double acc = 0.0;
  for (int r = 1; r <= rounds && !fail; ++r) {
    double2 v = make_double2((double)r + acc * 1e-30, (double)lane);
    u64* dst = reinterpret_cast<u64*>(rows + ((size_t)(r & 1) * G + b) * 64 + lane);
    __hip_atomic_store(dst, (u64)__double_as_longlong(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 1, (u64)__double_as_longlong(v.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(flags + (size_t)b * stride, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
      bool ok = true;
      if (lane < G) ok = __hip_atomic_load(flags + (size_t)lane * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)r;
      if (__all(ok)) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { fail = true; break; }
    }
    const u64* src = reinterpret_cast<const u64*>(rows + ((size_t)(r & 1) * G + (r % G)) * 64 + lane);
    acc += __longlong_as_double((long long)__hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (lane == 0 && b == 0) out_ticks[0] = t1 - t0;
  if (lane == 0 && fail) out_fail[0] = 1;
  if (acc == 12345.678) sink[0] = acc;
}

static void make_mask(std::vector<uint32_t>& m, int lo, int hi) {   // bits [lo, hi) set
  m.assign(8, 0u);
  for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32);
}

static void census(hipStream_t st, unsigned* d_out, const char* name) {
  const int G = 2048;
  CK(hipMemsetAsync(d_out, 0xff, 8 * G, st));
  hipLaunchKernelGGL(census_kernel, dim3(G), dim3(64), 0, st, d_out, 2000);   // 20 us each
  CK(hipStreamSynchronize(st));
  std::vector<unsigned> h(2 * G);
  CK(hipMemcpy(h.data(), d_out, 8 * G, hipMemcpyDeviceToHost));
  std::set<unsigned> cus; std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int i = 0; i < G; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    cus.insert(key); per_xcc[xcc].insert(key & 0xfff);
  }
  printf("census %-28s: %3zu distinct CUs;", name, cus.size());
  for (auto& kv : per_xcc) printf(" xcc%u:%zu", kv.first, kv.second.size());
  printf("\n");
  if (cus.size() <= 72) {
    printf("   (xcc.se.cu):");
    for (unsigned k : cus) printf(" %u.%u.%u", k >> 12, (k >> 8) & 0xf, k & 0xf);
    printf("\n");
  }
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 64;        // CUs of the panel set = mask bits [0, P)
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  printf("device: %s, %d CUs; panel set = mask bits [0, %d)\n", prop.name, prop.multiProcessorCount, P);
  std::vector<uint32_t> mA, mB, mAll;
  make_mask(mA, 0, P); make_mask(mB, P, 256); make_mask(mAll, 0, 256);
  hipStream_t sA, sB, sU, sU2;
  hipError_t e = hipExtStreamCreateWithCUMask(&sA, 8, mA.data());
  if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e)); return 0; }
  CK(hipExtStreamCreateWithCUMask(&sB, 8, mB.data()));
  CK(hipStreamCreateWithFlags(&sU, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sU2, hipStreamNonBlocking));
  unsigned* d_cen; CK(hipMalloc(&d_cen, 8 * 4096));
  census(sU, d_cen, "unmasked");
  census(sA, d_cen, "mask A = bits [0,P)");
  census(sB, d_cen, "mask B = bits [P,256)");
  { std::vector<uint32_t> m; make_mask(m, 0, 8); hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m.data())); census(s, d_cen, "bits [0,8)"); CK(hipStreamDestroy(s)); }
  { std::vector<uint32_t> m; make_mask(m, 8, 16); hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m.data())); census(s, d_cen, "bits [8,16)"); CK(hipStreamDestroy(s)); }
  { std::vector<uint32_t> m; make_mask(m, 0, 32); hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m.data())); census(s, d_cen, "bits [0,32)"); CK(hipStreamDestroy(s)); }

  unsigned *fl, *xf, *d_fail; u64* ticks; double2* rows; double* sink;
  const size_t fbytes = 64 * 1024 * 4;
  CK(hipMalloc(&fl, fbytes)); CK(hipMalloc(&xf, fbytes)); CK(hipMalloc(&ticks, 8)); CK(hipMalloc(&d_fail, 4));
  CK(hipMalloc(&rows, 2 * 256 * 64 * 16)); CK(hipMalloc(&sink, 8));
  typedef int (*zg_t)(int, int, int, const void*, const void*, void*, int, void*);
  void* lib = dlopen("math_audio_amd/lib/libmathaudio_hip.so", RTLD_NOW);
  zg_t zg = lib ? (zg_t)dlsym(lib, "ma_diag_zgemm_dev") : nullptr;
  if (!zg) { printf("libmathaudio_hip.so not found\n"); return 0; }
  const int ZM = 8192;
  void *zA, *zB, *zC;
  CK(hipMalloc(&zA, (size_t)ZM * 256 * 16)); CK(hipMalloc(&zB, (size_t)ZM * 256 * 16)); CK(hipMalloc(&zC, (size_t)ZM * ZM * 16));
  CK(hipMemset(zA, 0, (size_t)ZM * 256 * 16)); CK(hipMemset(zB, 0, (size_t)ZM * 256 * 16)); CK(hipMemset(zC, 0, (size_t)ZM * ZM * 16));

  // 3. the update kernel alone: all CUs, 256 - P CUs, P CUs
  for (int w = 0; w < 3; ++w) {
    hipStream_t s = w == 0 ? sU : (w == 1 ? sB : sA);
    zg(ZM, ZM, 256, zA, zB, zC, 2, s); CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = w == 2 ? 6 : 20;
    CK(hipEventRecord(e0, s)); zg(ZM, ZM, 256, zA, zB, zC, reps, s); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("update kernel %dx%dx256 alone on %-22s: %7.3f ms per launch = %6.1f TFLOP/s\n", ZM, ZM, w == 0 ? "an unmasked stream" : (w == 1 ? "mask B (256-P CUs)" : "mask A (P CUs)"), ms / reps,
           8.0 * ZM * (double)ZM * 256 / (ms / reps * 1e-3) / 1e12);
    fflush(stdout);
  }

  // 2. exchange rounds. stream of the round x background {none, unmasked update, update on mask B}
  struct Shape { const char* name; int G, NG, kind; };
  const Shape shapes[] = {{"two-level 227 wg / 8 groups", 227, 8, 0}, {"two-level 64 wg / 8 groups", 64, 8, 0}, {"flat 32 wg", 32, 0, 0}, {"flat 64 wg", 64, 0, 0},
                          {"flat 40 wg + 1 KB row", 40, 0, 1}, {"flat 64 wg + 1 KB row", 64, 0, 1}};
  const int rounds = 20000;
  for (const Shape& sh : shapes) {
    for (int cfg = 0; cfg < 5; ++cfg) {
      // cfg 0: round unmasked, idle | 1: round on A, idle | 2: round unmasked beside unmasked update (today) | 3: round on A beside update on B | 4: round on A beside UNMASKED update
      hipStream_t sr = (cfg == 0 || cfg == 2) ? sU : sA;
      hipStream_t sbg = (cfg == 2 || cfg == 4) ? sU2 : sB;
      const bool bg = cfg >= 2;
      CK(hipMemset(fl, 0, fbytes)); CK(hipMemset(xf, 0, fbytes)); CK(hipMemset(ticks, 0, 8)); CK(hipMemset(d_fail, 0, 4)); CK(hipMemset(rows, 0, 2 * 256 * 64 * 16));
      CK(hipDeviceSynchronize());
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, sbg));
      const int bgreps = 400;
      if (bg) zg(ZM, ZM, 256, zA, zB, zC, bgreps, sbg);
      CK(hipEventRecord(e1, sbg));
      if (sh.kind == 0) hipLaunchKernelGGL(round_kernel, dim3(sh.G), dim3(64), 0, sr, fl, xf, sh.NG, 32, rounds, ticks, d_fail);
      else hipLaunchKernelGGL(row_round_kernel, dim3(sh.G), dim3(64), 0, sr, fl, rows, 32, rounds, ticks, d_fail, sink);
      CK(hipStreamSynchronize(sr));
      const bool still = hipEventQuery(e1) == hipErrorNotReady;
      CK(hipDeviceSynchronize());
      float bgms = 0; CK(hipEventElapsedTime(&bgms, e0, e1));
      u64 t; unsigned f; CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&f, d_fail, 4, hipMemcpyDeviceToHost));
      const char* cn[] = {"round unmasked, idle chip", "round on mask A, idle chip", "round unmasked | update unmasked", "round on mask A | update on mask B", "round on mask A | update unmasked"};
      printf("%-28s | %-36s | %7.3f us per round%s", sh.name, cn[cfg], t / 100.0 / rounds, f ? " (TIMED OUT)" : "");
      if (bg) printf(" | update %6.3f ms per launch%s", bgms / bgreps, still ? " (outlasted the rounds)" : " (ENDED EARLY)");
      printf("\n");
      fflush(stdout);
    }
  }
  return 0;
}
