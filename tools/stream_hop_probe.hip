// Diagnostic (round 3): what a dependency between two streams costs (record on one, wait on the other, a tiny kernel each), for
// ordinary non-blocking streams, CU-masked streams (hipExtStreamCreateWithCUMask: a hardware queue of their own) and mixed pairs.
// build: hipcc -O2 --offload-arch=gfx950 tools/stream_hop_probe.hip -o tools/stream_hop_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void tiny(unsigned* p) { if (threadIdx.x == 0) atomicAdd(p, 1u); }

static double pingpong(hipStream_t a, hipStream_t b, unsigned* d, int hops, bool timing_events) {
  hipEvent_t ea, eb;
  CK(hipEventCreateWithFlags(&ea, timing_events ? hipEventDefault : hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&eb, timing_events ? hipEventDefault : hipEventDisableTiming));
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < hops; ++i) {
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
    CK(hipEventRecord(ea, a)); CK(hipStreamWaitEvent(b, ea, 0));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d);
    CK(hipEventRecord(eb, b)); CK(hipStreamWaitEvent(a, eb, 0));
  }
  CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
  auto t1 = std::chrono::steady_clock::now();
  CK(hipEventDestroy(ea)); CK(hipEventDestroy(eb));
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / (2.0 * hops);
}

int main() {
  CK(hipSetDevice(0));
  unsigned* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
  std::vector<uint32_t> mA(8, 0u), mB(8, 0u);
  for (int i = 0; i < 256; ++i) (i < 32 ? mA : mB)[i / 32] |= 1u << (i % 32);
  hipStream_t n1, n2, a1, a2, b1;
  CK(hipStreamCreateWithFlags(&n1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&n2, hipStreamNonBlocking));
  CK(hipExtStreamCreateWithCUMask(&a1, 8, mA.data())); CK(hipExtStreamCreateWithCUMask(&a2, 8, mA.data())); CK(hipExtStreamCreateWithCUMask(&b1, 8, mB.data()));
  const int hops = 2000;
  // same-stream baseline: 2 kernels per iteration on one stream
  {
    CK(hipDeviceSynchronize());
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2 * hops; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, n1, d);
    CK(hipStreamSynchronize(n1));
    auto t1 = std::chrono::steady_clock::now();
    printf("same stream, back to back                  : %7.2f us per kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (2.0 * hops));
    CK(hipDeviceSynchronize());
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 2 * hops; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a1, d);
    CK(hipStreamSynchronize(a1));
    t1 = std::chrono::steady_clock::now();
    printf("same MASKED stream, back to back           : %7.2f us per kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (2.0 * hops));
  }
  for (int rep = 0; rep < 2; ++rep) {
    const bool te = rep == 1;
    printf("-- events %s\n", te ? "with timing" : "hipEventDisableTiming");
    printf("normal <-> normal                          : %7.2f us per hop\n", pingpong(n1, n2, d, hops, te));
    printf("masked A <-> normal                        : %7.2f us per hop\n", pingpong(a1, n1, d, hops, te));
    printf("masked A <-> masked A (two streams)        : %7.2f us per hop\n", pingpong(a1, a2, d, hops, te));
    printf("masked A <-> masked B                      : %7.2f us per hop\n", pingpong(a1, b1, d, hops, te));
  }
  return 0;
}
