// Standalone probe: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950, with the in-kernel clock.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, unsigned long long* clk, int iters) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ __launch_bounds__(256) void k_fma(double* out, unsigned long long* clk, int iters) {
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3 + i;
  double a = 1.0000001, b = 1e-9;
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], a, b);
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename F>
void run(const char* name, F launch, int blocks, int threads, double flop_per_thread_iter_wave, int iters, double* d, unsigned long long* dc) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(blocks, threads, iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(a); launch(blocks, threads, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), dc, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double ghz = (double)h[0] / (double)h[1] * 0.1;
  double waves = (double)blocks * threads / 64.0;
  double tf = waves * iters * flop_per_thread_iter_wave / (ms * 1e-3) / 1e12;
  printf("%-34s blocks=%4d thr=%4d  %8.3f ms  %7.2f TFLOP/s  in-kernel clock %.2f GHz  cycles/iter/wave %.1f\n", name, blocks, threads, ms, tf, ghz,
         (double)h[0] / iters);
}

int main() {
  double* d; unsigned long long* dc;
  hipMalloc(&d, sizeof(double) * 1024 * 2048); hipMalloc(&dc, sizeof(unsigned long long) * 2 * 4096);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  int cu = p.multiProcessorCount;
  const double mf = 2.0 * 16 * 16 * 4;
  for (int wps = 1; wps <= 4; wps *= 2) {
    int threads = 256, blocks = cu * wps;   // wps waves per SIMD
    run("mfma f64 16x16x4, 4 acc", [&](int b, int t, int it) { hipLaunchKernelGGL(k_mfma<4>, dim3(b), dim3(t), 0, 0, d, dc, it); }, blocks, threads, 4 * mf, 20000, d, dc);
    run("mfma f64 16x16x4, 8 acc", [&](int b, int t, int it) { hipLaunchKernelGGL(k_mfma<8>, dim3(b), dim3(t), 0, 0, d, dc, it); }, blocks, threads, 8 * mf, 10000, d, dc);
    run("mfma f64 16x16x4, 16 acc", [&](int b, int t, int it) { hipLaunchKernelGGL(k_mfma<16>, dim3(b), dim3(t), 0, 0, d, dc, it); }, blocks, threads, 16 * mf, 5000, d, dc);
    run("v_fma_f64, 16 chains", [&](int b, int t, int it) { hipLaunchKernelGGL(k_fma, dim3(b), dim3(t), 0, 0, d, dc, it); }, blocks, threads, 16 * 64 * 2.0, 20000, d, dc);
  }
  return 0;
}
