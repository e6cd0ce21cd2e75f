// ma_common.hpp — shared host-side plumbing for libmathaudio_hip.so (error text, HIP checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <string>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>
#include "../../include/mathaudio_hip.h"

namespace ma {

void set_error(const char* fmt, ...);

// Evaluate a HIP call; on failure record the text and return MA_ERR_HIP from the caller.
#define MA_HIP(call)                                                                     \
  do {                                                                                   \
    hipError_t ma_e_ = (call);                                                           \
    if (ma_e_ != hipSuccess) {                                                           \
      ::ma::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(ma_e_), __FILE__, __LINE__); \
      return (ma_e_ == hipErrorOutOfMemory) ? MA_ERR_NOMEM : MA_ERR_HIP;                 \
    }                                                                                    \
  } while (0)

#define MA_REQUIRE(cond, code, ...)                 \
  do {                                              \
    if (!(cond)) {                                  \
      ::ma::set_error(__VA_ARGS__);                 \
      return (code);                                \
    }                                               \
  } while (0)

// Select `device` after checking that one exists and is a gfx950 part. No CPU fallback.
int use_device(int device);

struct c64 { double re, im; };
static_assert(sizeof(c64) == sizeof(ma_c64), "layout");

// Host threads for setup work that is independent per item (MA_HOST_THREADS, default min(16, cores)). fn(begin, end) gets contiguous
// blocks; callers keep per-item results in per-item slots, so what is computed does not depend on the count.
inline int host_threads() {
  const char* e = getenv("MA_HOST_THREADS");
  int t = e ? atoi(e) : 0;
  if (t <= 0) { t = (int)std::thread::hardware_concurrency(); if (t > 16) t = 16; }
  return t < 1 ? 1 : t;
}
inline void host_parallel_for(long long n, long long min_per_thread, const std::function<void(long long, long long)>& fn) {
  long long T = host_threads();
  if (min_per_thread > 0 && n / min_per_thread < T) T = n / min_per_thread;
  if (T <= 1) { fn(0, n); return; }
  std::vector<std::thread> th;
  for (long long t = 0; t < T; ++t) th.emplace_back([&, t]() { fn(n * t / T, n * (t + 1) / T); });
  for (auto& x : th) x.join();
}

// HIP event pair helper for the optional per-phase timing.
struct PhaseTimer {
  hipEvent_t a = nullptr, b = nullptr;
  bool armed = false;
  int init() {
    MA_HIP(hipEventCreate(&a));
    MA_HIP(hipEventCreate(&b));
    return MA_OK;
  }
  void destroy() {
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    a = b = nullptr;
  }
};

}  // namespace ma
