// ma_common.hpp — shared host-side plumbing for libmathaudio_hip.so (error text, HIP checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <string>
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
