// Round 5 probe: the tournament panel kernels alone on the chip -- launch time by events and the root chain's phases by stamps.
// hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMA_CALU_STAMPS -Imath_audio_amd/csrc tools/calu_probe.hip math_audio_amd/csrc/<set_error stub> ...
#include "../math_audio_amd/csrc/lu_calu.hip"
#include <vector>
#include <random>
namespace ma { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
using namespace ma;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 10000;
  const int reps = 20;
  std::vector<c64> h((size_t)n * 64);
  std::mt19937_64 g(1); std::normal_distribution<double> N(0.0, 1.0);
  c64* dA; CK(hipMalloc(&dA, sizeof(c64) * (size_t)n * n));
  CK(hipMemset(dA, 0, sizeof(c64) * (size_t)n * n));
  for (auto& v : h) { v.re = N(g); v.im = N(g); }
  for (int r = 0; r < n; ++r) CK(hipMemcpy(dA + (size_t)r * n, h.data() + (size_t)r * 64, sizeof(c64) * 64, hipMemcpyHostToDevice));
  LuCaluWs ws; const int nodes = lu_calu_tree_nodes((n + 255) / 256);
  CK(hipMalloc(&ws.cand, sizeof(int) * nodes * LU_REG_NB)); CK(hipMalloc(&ws.counters, sizeof(unsigned) * nodes)); CK(hipMemset(ws.counters, 0, sizeof(unsigned) * nodes)); ws.max_nodes = nodes;
  int *info, *ipiv, *lists; CK(hipMalloc(&info, 64)); CK(hipMemset(info, 0, 64)); CK(hipMalloc(&ipiv, sizeof(int) * n)); CK(hipMalloc(&lists, sizeof(int) * (1 + 4 * LU_NB_MAX)));
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  for (int k0 : {0, 32, n / 2, n - 2048, n - 256}) {
    if (k0 < 0 || k0 + 32 > n) continue;
    const int leaves = (n - k0 + 255) / 256;
    unsigned long long z[16] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_calu_stamps), z, sizeof(z)));
    float tp = 0, tf = 0;
    for (int r = 0; r < reps + 2; ++r) {
      if (r == 2) CK(hipMemcpyToSymbol(HIP_SYMBOL(g_calu_stamps), z, sizeof(z)));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(lu_calu_panel_kernel<LU_REG_NB>, dim3(leaves), dim3(256), 0, 0, reinterpret_cast<dc*>(dA), n, k0, 32, ws.cand, ws.counters, info, ipiv, lists, (dc*)nullptr, 0, (const int*)nullptr);
      CK(hipEventRecord(e1));
      hipLaunchKernelGGL(lu_calu_finish_kernel<LU_REG_NB>, dim3((n - k0 - 32 + 255) / 256), dim3(256), 0, 0, reinterpret_cast<dc*>(dA), n, k0, 32, (const int*)nullptr);
      CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
      float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
      if (r >= 2) { tp += a; tf += b; }
    }
    unsigned long long s[16]; CK(hipMemcpyFromSymbol(s, HIP_SYMBOL(g_calu_stamps), sizeof(s)));
#ifdef MA_CALU_SUBSTAMPS
    unsigned long long u[8]; CK(hipMemcpyFromSymbol(u, HIP_SYMBOL(g_calu_sub), sizeof(u)));
    { const double C = (double)(u[5] ? u[5] : 1); printf("   per column, shader clocks (workgroup 0, thread 0): reduce+recip %.0f  row->LDS %.0f  barrier %.0f  select %.0f  eliminate %.0f\n", u[0] / C, u[1] / C, u[2] / C, u[3] / C, u[4] / C); }
    unsigned long long z8[8] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_calu_sub), z8, sizeof(z8)));
#endif
    const double R = (double)(s[7] ? s[7] : 1) * 100.0;   // ticks of 10 ns -> us per root
    printf("n %d k0 %5d leaves %3d: panel %.1f us  finish %.1f us | root chain us: leaf load %.1f  leaf elim %.1f  publish %.1f  node load %.1f  node elim %.1f  root seq+reads %.1f  root writes %.1f\n",
           n, k0, leaves, tp / reps * 1e3, tf / reps * 1e3, s[0] / R, s[1] / R, s[2] / R, s[3] / R, s[4] / R, s[5] / R, s[6] / R);
  }
  return 0;
}
