// Diagnostic: throughput of the LU trailing-update kernel (C -= A B, complex128) as a function of K.
// build: hipcc -O2 --offload-arch=gfx950 -I math_audio_amd/csrc tools/zgemm_bench.hip -L math_audio_amd/lib -lmathaudio_hip -Wl,-rpath,$PWD/math_audio_amd/lib -o /tmp/zgemm_bench
#include "lu_kernels.hpp"
#include <cstdio>
#include <vector>
using namespace ma;
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 10000;
  c64 *A, *B, *C;
  const size_t ld = (size_t)n + 512;
  hipMalloc(&A, sizeof(c64) * ld * ld);
  {   // random entries of magnitude ~1e-3 (so that repeated C -= A B stays finite): zeros would run at atypical clocks
    std::vector<double> h(2 * ld * 64);
    unsigned long long sd = 88172645463325252ull;
    for (auto& v : h) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; v = ((double)(sd >> 11) / 9007199254740992.0 - 0.5) * 2e-3; }
    for (size_t r = 0; r < ld; r += 64) hipMemcpy(A + r * ld, h.data(), sizeof(c64) * ld * (r + 64 <= ld ? 64 : ld - r), hipMemcpyHostToDevice);
  }
  B = A; C = A;
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int use3m = 1; use3m >= 0; --use3m)
    for (int K : {64, 128, 192, 256, 384, 512}) {
      // C at (K.., K..), A at (K.., 0..K), B at (0..K, K..): the shapes of a trailing update
      const c64* pa = A + (size_t)K * ld; const c64* pb = A + K; c64* pc = A + (size_t)K * ld + K;
      for (int it = 0; it < 2; ++it) lu_launch_zgemm_sub(n, n, K, pa, ld, pb, ld, pc, ld, st, use3m);
      hipEventRecord(e0, st);
      const int reps = 5;
      for (int it = 0; it < reps; ++it) lu_launch_zgemm_sub(n, n, K, pa, ld, pb, ld, pc, ld, st, use3m);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      printf("%s n=%d K=%3d  %.3f ms  %.1f TFLOP/s algorithmic  C traffic %.2f TB/s\n", use3m ? "3M" : "4M", n, K, ms, 8.0 * n * (double)n * K / ms * 1e-9,
             32.0 * n * (double)n / ms * 1e-9);
    }
  return 0;
}
